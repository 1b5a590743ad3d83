// tip_stream.hip — on-device streaming front/back-end: the model-facing half of
// RTRunnerMin.step (/root/reference/real_time_runner_minimal.py) for B concurrent streams, so that a closed-loop
// frame costs ZERO host<->device round trips for the model inputs (the reference crosses PCIe twice per frame per
// stream at :146-150 and does ~7 ms of numpy per stream on the host).
//
//   stream_ingest_kernel   :59-76   raw-IMU ring (5-frame priming), 11-tap acceleration mean, rotations delayed 5
//                          :131-147 root-frame rotation of the 5 other IMUs (data_utils.py:190-219), acc-sum feature
//                                   over the <=40-frame window (/15, constants.py:17-18), window gather -> x_imu, x_s
//   stream_consume_kernel  :87-112  6-tap 0.6^k output filter (incl. the reference's in-place quirk on the first 5
//                                   rows), SBP flag threshold, offsets / 5
//                          :154-167 6D -> axis-angle (data_utils.py:164-179; scipy Rotation.from_matrix/as_rotvec
//                                   restated), root rotation from the IMU, averaging with the previous pose
//                          :78-85   axis-angle -> 6D history row (data_utils.py:182-187; from_rotvec/as_matrix)
// PyBullet FK and the SBP root-translation correction (:169-194) stay on the host, unchanged: they only touch the
// root translation, which is not a model input.
//
// State per stream (floats, caller-owned device buffer): raw[11][72] | loc[40][72] | accs[40][18] | hist[40][131] |
// outs[6][131] | last[54] | frame counter (int).  Frame / call counters live on the host (all streams advance in lock step); the
// ingest kernel also leaves the frame index in the state, so that a call with TIP_STREAM_FRAME_AUTO continues from it — kernel
// arguments are frozen in a captured HIP graph, a counter in HBM is not.
#include "tip_internal.h"

namespace tip {

namespace sz {
constexpr int NIMU = 72, NS = 131, NX = 90, WIN = 40, RAWN = 11, OUTN = 6;
constexpr int RAW = 0;
constexpr int LOC = RAW + RAWN * NIMU;        // 792
constexpr int ACCS = LOC + WIN * NIMU;        // 3672
constexpr int HIST = ACCS + WIN * 18;         // 4392
constexpr int OUTS = HIST + WIN * NS;         // 9632
constexpr int LAST = OUTS + OUTN * NS;        // 10418
constexpr int CTR = LAST + 54;                // 10472: frame index of the last ingest (int)
constexpr int STRIDE = 10496;                 // CTR + 1 = 10473, padded to a multiple of 64
}  // namespace sz

// ---- scipy.spatial.transform.Rotation restated (fp32) -------------------------------------------------------
// from_matrix() first replaces a non-orthogonal input by the NEAREST rotation (U V^T of its SVD), then extracts the
// quaternion.  The two inputs that occur here have closed forms for that polar factor:
//   * M = [a1 a2 a1 x a2] (data_utils.py:171-176, columns normalised but not orthogonalised): M^T M is block
//     diagonal, so R = [ [a1 a2] A^{-1/2} , a3/|a3| ] with the 2x2 A = [[a1.a1, a1.a2], [a1.a2, a2.a2]] and
//     A^{1/2} = (A + sqrt(det A) I) / sqrt(tr A + 2 sqrt(det A)).
//   * the root IMU rotation (orthonormal up to sensor round-off): two Newton steps X <- (X + X^{-T}) / 2.
__device__ __forceinline__ void polar_two_axis(const float a1[3], const float a2[3], float m[3][3]) {
    const float a3[3] = {a1[1] * a2[2] - a1[2] * a2[1], a1[2] * a2[0] - a1[0] * a2[2], a1[0] * a2[1] - a1[1] * a2[0]};
    const float p = a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2];
    const float q = a2[0] * a2[0] + a2[1] * a2[1] + a2[2] * a2[2];
    const float c = a1[0] * a2[0] + a1[1] * a2[1] + a1[2] * a2[2];
    const float d = sqrtf(fmaxf(p * q - c * c, 0.f));
    const float tau = sqrtf(p + q + 2.f * d);
    const float den = (p + d) * (q + d) - c * c;
    const float n3sq = a3[0] * a3[0] + a3[1] * a3[1] + a3[2] * a3[2];
    if (!(den > 1e-10f) || !(n3sq > 1e-12f)) {
        // (anti)parallel or vanishing predicted axes: M is singular, its polar factor is not unique (scipy's SVD returns SOME rotation
        // there) and the closed form above divides by zero — seen once in 300 k stream-frames of a random-weight closed loop, after
        // which the NaN lived on in the pose average.  Any orthonormal frame is as right as another: first axis along a1 (or x), the
        // second from the coordinate axis least aligned with it.
        float e1[3] = {a1[0], a1[1], a1[2]};
        float n1 = sqrtf(p);
        if (!(n1 > 1e-12f)) { e1[0] = 1.f; e1[1] = 0.f; e1[2] = 0.f; n1 = 1.f; }
        for (int i = 0; i < 3; ++i) e1[i] /= n1;
        const float ax = fabsf(e1[0]), ay = fabsf(e1[1]), az = fabsf(e1[2]);
        float h[3] = {0.f, 0.f, 0.f};
        h[(ax <= ay && ax <= az) ? 0 : (ay <= az ? 1 : 2)] = 1.f;
        const float hd = h[0] * e1[0] + h[1] * e1[1] + h[2] * e1[2];
        float e2[3] = {h[0] - hd * e1[0], h[1] - hd * e1[1], h[2] - hd * e1[2]};
        const float n2 = sqrtf(e2[0] * e2[0] + e2[1] * e2[1] + e2[2] * e2[2]);
        for (int i = 0; i < 3; ++i) e2[i] /= n2;
        const float e3[3] = {e1[1] * e2[2] - e1[2] * e2[1], e1[2] * e2[0] - e1[0] * e2[2], e1[0] * e2[1] - e1[1] * e2[0]};
        for (int i = 0; i < 3; ++i) { m[i][0] = e1[i]; m[i][1] = e2[i]; m[i][2] = e3[i]; }
        return;
    }
    const float f = tau / den;
    const float b00 = (q + d) * f, b01 = -c * f, b11 = (p + d) * f;
    const float n3 = 1.f / sqrtf(n3sq);
    for (int i = 0; i < 3; ++i) {
        m[i][0] = a1[i] * b00 + a2[i] * b01;
        m[i][1] = a1[i] * b01 + a2[i] * b11;
        m[i][2] = a3[i] * n3;
    }
}

__device__ __forceinline__ void polar_newton(float m[3][3]) {
    for (int it = 0; it < 2; ++it) {
        // inverse transpose = cofactor matrix / det
        float cof[3][3];
        cof[0][0] = m[1][1] * m[2][2] - m[1][2] * m[2][1];
        cof[0][1] = m[1][2] * m[2][0] - m[1][0] * m[2][2];
        cof[0][2] = m[1][0] * m[2][1] - m[1][1] * m[2][0];
        cof[1][0] = m[0][2] * m[2][1] - m[0][1] * m[2][2];
        cof[1][1] = m[0][0] * m[2][2] - m[0][2] * m[2][0];
        cof[1][2] = m[0][1] * m[2][0] - m[0][0] * m[2][1];
        cof[2][0] = m[0][1] * m[1][2] - m[0][2] * m[1][1];
        cof[2][1] = m[0][2] * m[1][0] - m[0][0] * m[1][2];
        cof[2][2] = m[0][0] * m[1][1] - m[0][1] * m[1][0];
        const float det = m[0][0] * cof[0][0] + m[0][1] * cof[0][1] + m[0][2] * cof[0][2];
        const float id = 1.f / det;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) m[i][j] = 0.5f * (m[i][j] + cof[i][j] * id);
    }
}

// quaternion of a rotation matrix by the largest-of-{diag, trace} rule, normalised (the extraction scipy uses).
__device__ __forceinline__ void mat_to_quat(const float m[3][3], float q[4]) {
    const float tr = m[0][0] + m[1][1] + m[2][2];
    int choice = 0;
    float best = m[0][0];
    if (m[1][1] > best) { best = m[1][1]; choice = 1; }
    if (m[2][2] > best) { best = m[2][2]; choice = 2; }
    if (tr > best) { choice = 3; }
    if (choice != 3) {
        const int i = choice, j = (i + 1) % 3, k = (j + 1) % 3;
        q[i] = 1.f - tr + 2.f * m[i][i];
        q[j] = m[j][i] + m[i][j];
        q[k] = m[k][i] + m[i][k];
        q[3] = m[k][j] - m[j][k];
    } else {
        q[0] = m[2][1] - m[1][2];
        q[1] = m[0][2] - m[2][0];
        q[2] = m[1][0] - m[0][1];
        q[3] = 1.f + tr;
    }
    const float n = sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

// as_rotvec(): w >= 0 branch, small-angle series below 1e-3.
__device__ __forceinline__ void quat_to_rotvec(const float qi[4], float rv[3]) {
    float q[4] = {qi[0], qi[1], qi[2], qi[3]};
    if (q[3] < 0.f) { q[0] = -q[0]; q[1] = -q[1]; q[2] = -q[2]; q[3] = -q[3]; }
    const float angle = 2.f * atan2f(sqrtf(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]), q[3]);
    float scale;
    if (angle <= 1e-3f) {
        const float a2 = angle * angle;
        scale = 2.f + a2 / 12.f + 7.f * a2 * a2 / 2880.f;
    } else {
        scale = angle / sinf(angle * 0.5f);
    }
    rv[0] = scale * q[0]; rv[1] = scale * q[1]; rv[2] = scale * q[2];
}

// from_rotvec().as_matrix()[:, :2] -> (3x2) row-major
__device__ __forceinline__ void rotvec_to_6d(const float rv[3], float out[6]) {
    const float angle = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    float scale;
    if (angle <= 1e-3f) {
        const float a2 = angle * angle;
        scale = 0.5f - a2 / 48.f + a2 * a2 / 3840.f;
    } else {
        scale = sinf(angle * 0.5f) / angle;
    }
    const float x = scale * rv[0], y = scale * rv[1], z = scale * rv[2], w = cosf(angle * 0.5f);
    const float x2 = x * x, y2 = y * y, z2 = z * z, w2 = w * w;
    const float xy = x * y, zw = z * w, xz = x * z, yw = y * w, yz = y * z, xw = x * w;
    out[0] = x2 - y2 - z2 + w2;   // m00
    out[1] = 2.f * (xy - zw);     // m01
    out[2] = 2.f * (xy + zw);     // m10
    out[3] = -x2 + y2 - z2 + w2;  // m11
    out[4] = 2.f * (xz - yw);     // m20
    out[5] = 2.f * (yz + xw);     // m21
}

// ---- reset: history row 0 from s_init (:45, :78-85) ----------------------------------------------------------
__global__ __launch_bounds__(64) void stream_reset_kernel(float* __restrict__ state, const float* __restrict__ s_init, int B) {
    using namespace sz;
    const int b = blockIdx.x, tid = threadIdx.x;
    float* S = state + (size_t)b * STRIDE;
    for (int i = tid; i < STRIDE; i += 64) S[i] = 0.f;
    __syncthreads();
    const float* si = s_init + (size_t)b * 114;
    if (tid < 18) {
        float rv[3] = {si[3 + tid * 3], si[4 + tid * 3], si[5 + tid * 3]}, o[6];
        rotvec_to_6d(rv, o);
        for (int e = 0; e < 6; ++e) S[HIST + tid * 6 + e] = o[e];
    }
    if (tid < 3) S[HIST + 108 + tid] = si[57 + tid];
}

// ---- ingest one raw frame per stream and emit the model inputs -----------------------------------------------
// newest != 0 (tip_stream_ingest_newest): only row T - 1 of every window is written — what the exact-reuse forward reads
// (tip_forward_reuse takes every older row from its ring).  At >= 1024 streams the window gather IS this kernel: 35 KB read and 35 KB
// written per stream, 18 us at 1024 streams and 53 us at 4096 against 8-9 us for the smoothing chain.
__global__ __launch_bounds__(256) void stream_ingest_kernel(float* __restrict__ state, const float* __restrict__ raw_in, int B,
                                                            int f, float* __restrict__ x_imu, float* __restrict__ x_s, int T, int newest) {
    using namespace sz;
    __shared__ float sm[NIMU], loc[NIMU];
    const int b = blockIdx.x, tid = threadIdx.x;
    float* S = state + (size_t)b * STRIDE;
    int* ctr = reinterpret_cast<int*>(S + CTR);
    if (f < 0) {   // TIP_STREAM_FRAME_AUTO: the frame after the last one ingested (every thread reads before thread 0 writes)
        f = *ctr + 1;
        __syncthreads();
    }
    if (tid == 0) *ctr = f;
    // Everything that does not depend on THIS frame goes first, so that it runs under the latency of the smoothing chain below (five
    // dependent phases with a barrier and a global round trip each) instead of after it: the whole x_s window (history rows written by
    // earlier consume calls), rows 0 .. T-2 of x_imu (local-frame rows and acc-sums of earlier frames) and the acc-sum of the older
    // frames.  As one tail behind the chain these copies were two thirds of the kernel's 13 us at one stream.
    const int k = f - 5;  // index of the smoothed frame produced now == index of this model call (f >= 5)
    float acc_old = 0.f;  // threads < 18: acc-sum over the window's older frames, oldest first (:136)
    if (f >= 5) {
        // (compile-time trip counts, every load of a loop requested before its first store: 36 round trips in flight together)
        constexpr int XS_IT = (WIN * NS + 255) / 256, XI_IT = (WIN * NX + 255) / 256;
        float* xs = x_s + (size_t)b * T * NS;
        float* xi = x_imu + (size_t)b * T * NX;
        if (newest) {
            if (tid < NS) xs[(T - 1) * NS + tid] = S[HIST + (k % WIN) * NS + tid];     // history entry k (:144)
        } else {
        float vs[XS_IT];
#pragma unroll
        for (int u = 0; u < XS_IT; ++u) {
            const int i = tid + 256 * u, ic = i < T * NS ? i : 0;
            const int t = ic / NS, c = ic - t * NS, j = k + 1 - T + t;   // history entries k+1-T .. k (:144)
            vs[u] = S[HIST + (j % WIN) * NS + c];
        }
        float vi[XI_IT];
#pragma unroll
        for (int u = 0; u < XI_IT; ++u) {
            const int i = tid + 256 * u, ic = i < (T - 1) * NX ? i : 0;
            const int t = ic / NX, c = ic - t * NX, j = k - T + 1 + t;
            const float* pv = c < NIMU ? S + LOC + (j % WIN) * NIMU + c : S + ACCS + (j % WIN) * 18 + (c - NIMU);
            const float x = *pv;
            vi[u] = c < NIMU ? x : x / 15.0f;                                                            // :139-141
        }
#pragma unroll
        for (int u = 0; u < XS_IT; ++u)
            if (tid + 256 * u < T * NS) xs[tid + 256 * u] = vs[u];
#pragma unroll
        for (int u = 0; u < XI_IT; ++u)
            if (tid + 256 * u < (T - 1) * NX) xi[tid + 256 * u] = vi[u];
        }
        if (tid < 18) {   // all (up to 39) loads first, then the sum in the reference's order: one round trip, not one per frame
            float av[WIN - 1];
#pragma unroll
            for (int u = 0; u < WIN - 1; ++u) {
                const int j = k - T + 1 + u;
                av[u] = S[LOC + ((j < k ? j : k - 1 < 0 ? 0 : k - 1) % WIN) * NIMU + 54 + tid];
            }
#pragma unroll
            for (int u = 0; u < WIN - 1; ++u)
                if (k - T + 1 + u < k) acc_old += av[u];
        }
    }
    // The chain: two LDS barriers.  The raw frame is used from its register (it also goes to the ring for the next ten calls), the ten
    // older raw rows do not depend on this call, and every thread inverts the 3 x 3 root matrix for itself (40 flops) — with the
    // frame stored, re-read behind a barrier, and the inverse computed by thread 0 between two more, the chain was five barriers
    // with a global round trip each.
    float vnew = 0.f;
    if (tid < NIMU) {
        vnew = raw_in[(size_t)b * NIMU + tid];
        if (f == 0) {
            for (int e = 0; e < 6; ++e) S[RAW + e * NIMU + tid] = vnew;   // 5 priming copies + the frame itself (:61-66)
        } else {
            S[RAW + ((f + 5) % RAWN) * NIMU + tid] = vnew;
        }
    }
    if (f < 5) return;   // fewer than 11 raw entries: the smoother is still priming (:68, :125-128)
    if (tid < 54) {
        sm[tid] = S[RAW + (f % RAWN) * NIMU + tid];            // rotations of the frame 5 steps back (:71)
    } else if (tid < NIMU) {
        float acc = 0.f;
        for (int i = 0; i < RAWN - 1; ++i) acc += S[RAW + ((f - 5 + i) % RAWN) * NIMU + tid];   // oldest -> newest (:72) ...
        acc += vnew;                                                                             // ... the newest is this call's frame
        sm[tid] = acc / (float)RAWN;
    }
    __syncthreads();
    if (tid < NIMU) {
        // general 3x3 inverse (np.linalg.inv at data_utils.py:200,211), adjugate / determinant
        float inv[9];
        {
            const float a = sm[0], bq = sm[1], c = sm[2], d = sm[3], e = sm[4], g = sm[5], h = sm[6], i9 = sm[7], j = sm[8];
            const float A = e * j - g * i9, Bc = -(d * j - g * h), Cc = d * i9 - e * h;
            const float det = a * A + bq * Bc + c * Cc;
            const float id = 1.f / det;
            inv[0] = A * id;  inv[1] = -(bq * j - c * i9) * id; inv[2] = (bq * g - c * e) * id;
            inv[3] = Bc * id; inv[4] = (a * j - c * h) * id;    inv[5] = -(a * g - c * d) * id;
            inv[6] = Cc * id; inv[7] = -(a * i9 - bq * h) * id; inv[8] = (a * e - bq * d) * id;
        }
        float lv;
        if (tid < 9) {
            lv = sm[tid];
        } else if (tid < 54) {
            const int s = (tid - 9) / 9, e = (tid - 9) % 9, i = e / 3, j = e % 3;
            const float* Rm = sm + 9 + s * 9;
            lv = inv[i * 3 + 0] * Rm[0 * 3 + j] + inv[i * 3 + 1] * Rm[1 * 3 + j] + inv[i * 3 + 2] * Rm[2 * 3 + j];
        } else if (tid < 57) {
            lv = sm[tid];
        } else {
            const int s = (tid - 57) / 3, i = (tid - 57) % 3;
            const float* am = sm + 57 + s * 3;
            lv = inv[i * 3 + 0] * am[0] + inv[i * 3 + 1] * am[1] + inv[i * 3 + 2] * am[2];
        }
        loc[tid] = lv;
    }
    __syncthreads();
    // this frame's row: to the state (next frames read it) and straight to the newest row of x_imu, both from LDS
    float* xin = x_imu + (size_t)b * T * NX + (size_t)(T - 1) * NX;
    if (tid < NIMU) {
        S[LOC + (k % WIN) * NIMU + tid] = loc[tid];
        xin[tid] = loc[tid];
    }
    if (tid < 18) {   // acc-sum feature over the (<= 40-frame) window: the older frames' partial + this frame, added last (:136)
        const float acc = acc_old + loc[54 + tid];
        S[ACCS + (k % WIN) * 18 + tid] = acc;
        xin[NIMU + tid] = acc / 15.0f;                        // :139-141
    }
}

// ---- consume the model's last row: filter, decode, pose assembly, history feedback ----------------------------
__global__ __launch_bounds__(192) void stream_consume_kernel(float* __restrict__ state, const float* __restrict__ y_last, int B,
                                                             int k, float* __restrict__ s_rest, float* __restrict__ c_out) {
    using namespace sz;
    __shared__ float s[NS], aa[54], rootv[3];
    const int b = blockIdx.x, tid = threadIdx.x;
    float* S = state + (size_t)b * STRIDE;
    const float coeff[OUTN] = {0.07776f, 0.1296f, 0.216f, 0.36f, 0.6f, 1.0f};   // 0.6^(5..0) (:57)
    const float csum = 0.07776f + 0.1296f + 0.216f + 0.36f + 0.6f + 1.0f;
    if (k < 0) k = *reinterpret_cast<const int*>(S + CTR) - 5;   // TIP_STREAM_FRAME_AUTO: the call that belongs to the last ingested frame
    const int n = k + 1;
    // requested up front, with the prediction row: the previous pose / root velocity (the averaging below) and the root IMU rotation
    // — behind the first barrier each was one more exposed round trip
    float last3[3] = {0.f, 0.f, 0.f}, rr[9];
    if (tid >= 1 && tid < 18 && k > 0)
        for (int e = 0; e < 3; ++e) last3[e] = S[LAST + (tid - 1) * 3 + e];
    if (tid >= 64 && tid < 67 && k > 0) last3[0] = S[LAST + 51 + (tid - 64)];
    if (tid == 0)
        for (int e = 0; e < 9; ++e) rr[e] = S[LOC + (k % WIN) * NIMU + e];
    if (tid < NS) {
        const float y = y_last[(size_t)b * NS + tid];
        S[OUTS + (k % OUTN) * NS + tid] = y;
        float v;
        if (n >= OUTN) {
            // the five older rows do not depend on this call's row: their loads go out with its load; the newest term is y itself
            // (the row just stored: same value, no store -> load round trip)
            float acc = 0.f;
            for (int i = 0; i < OUTN - 1; ++i) acc += S[OUTS + ((k - 5 + i) % OUTN) * NS + tid] * coeff[i];
            acc += y * coeff[OUTN - 1];
            v = acc / csum;
        } else {
            v = y;
        }
        if (tid >= 111) {
            const int ci = tid - 111;
            v = (ci & 3) == 0 ? (v > 0.f ? 1.f : 0.f) : v / 5.0f;      // :107-110
            // reference quirk: with fewer than 6 buffered rows the decode happens IN PLACE in the buffered row (:99)
            if (n < OUTN) S[OUTS + (k % OUTN) * NS + tid] = v;
            c_out[(size_t)b * 20 + ci] = v;
        }
        s[tid] = v;
    }
    __syncthreads();
    if (tid < 18) {
        float rv[3];
        if (tid == 0) {   // root rotation comes from the IMU, not from the prediction (:160-162)
            float m[3][3] = {{rr[0], rr[1], rr[2]}, {rr[3], rr[4], rr[5]}, {rr[6], rr[7], rr[8]}};
            polar_newton(m);
            float q[4];
            mat_to_quat(m, q);
            quat_to_rotvec(q, rv);
        } else {          // 6D -> rotation (columns normalised with +1e-6, third by cross product; data_utils.py:171-176)
            const float* p = s + tid * 6;
            float a1[3] = {p[0], p[2], p[4]}, a2[3] = {p[1], p[3], p[5]};
            const float n1 = sqrtf(a1[0] * a1[0] + a1[1] * a1[1] + a1[2] * a1[2]) + 1e-6f;
            const float n2 = sqrtf(a2[0] * a2[0] + a2[1] * a2[1] + a2[2] * a2[2]) + 1e-6f;
            for (int e = 0; e < 3; ++e) { a1[e] /= n1; a2[e] /= n2; }
            float m[3][3];
            polar_two_axis(a1, a2, m);
            float q[4];
            mat_to_quat(m, q);
            quat_to_rotvec(q, rv);
            if (k > 0) {  // averaged with the previous pose (:165-166)
                for (int e = 0; e < 3; ++e) rv[e] = (rv[e] + last3[e]) * 0.5f;
            }
            for (int e = 0; e < 3; ++e) S[LAST + (tid - 1) * 3 + e] = rv[e];
        }
        for (int e = 0; e < 3; ++e) aa[tid * 3 + e] = rv[e];
        float o6[6];
        rotvec_to_6d(rv, o6);   // :78-85
        for (int e = 0; e < 6; ++e) S[HIST + ((k + 1) % WIN) * NS + tid * 6 + e] = o6[e];
    } else if (tid >= 64 && tid < 67) {
        const int e = tid - 64;
        float v = s[108 + e];
        if (k > 0) v = (v + last3[0]) * 0.5f;
        S[LAST + 51 + e] = v;
        rootv[e] = v;
        S[HIST + ((k + 1) % WIN) * NS + 108 + e] = v;
    } else if (tid >= 128 && tid < 148) {
        S[HIST + ((k + 1) % WIN) * NS + 111 + (tid - 128)] = s[111 + (tid - 128)];
    }
    __syncthreads();
    // s_t[3:114]: 54 axis-angles, root velocity, 54 zeros
    if (tid < 111) s_rest[(size_t)b * 111 + tid] = tid < 54 ? aa[tid] : (tid < 57 ? rootv[tid - 54] : 0.f);
}

}  // namespace tip

using namespace tip;

extern "C" {

int tip_stream_state_bytes(int n_streams, size_t* bytes) {
    if (n_streams < 0 || !bytes) return TIP_ERR_INVALID_ARG;
    *bytes = (size_t)n_streams * sz::STRIDE * sizeof(float);
    return TIP_OK;
}

int tip_stream_reset(void* state, const float* s_init, int n_streams, tip_stream_t stream) {
    if (!state || !s_init || n_streams < 0) return TIP_ERR_INVALID_ARG;
    if (n_streams == 0) return TIP_OK;
    hipLaunchKernelGGL(stream_reset_kernel, dim3(n_streams), dim3(64), 0, static_cast<hipStream_t>(stream),
                       static_cast<float*>(state), s_init, n_streams);
    return hipGetLastError() == hipSuccess ? TIP_OK : TIP_ERR_HIP;
}

int tip_stream_frame_counter_offset(size_t* bytes) {   // where stream 0's block keeps the index of the last ingested frame (an int)
    if (!bytes) return TIP_ERR_INVALID_ARG;
    *bytes = (size_t)sz::CTR * sizeof(float);
    return TIP_OK;
}

int tip_stream_window_len(int frame_idx) {   // T of the model call issued for frame `frame_idx`; 0 while priming
    if (frame_idx < 5) return 0;
    const int t = frame_idx - 4;
    return t < sz::WIN ? t : sz::WIN;
}

static int stream_ingest(void* state, const float* raw_imu, int n_streams, int frame_idx, float* x_imu, float* x_s, tip_stream_t stream,
                         int newest) {
    if (!state || !raw_imu || n_streams < 0 || frame_idx < TIP_STREAM_FRAME_AUTO) return TIP_ERR_INVALID_ARG;
    const int T = frame_idx == TIP_STREAM_FRAME_AUTO ? sz::WIN : tip_stream_window_len(frame_idx);
    if (T > 0 && (!x_imu || !x_s)) return TIP_ERR_INVALID_ARG;
    if (n_streams == 0) return TIP_OK;
    hipLaunchKernelGGL(stream_ingest_kernel, dim3(n_streams), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<float*>(state), raw_imu, n_streams, frame_idx, x_imu, x_s, T, newest);
    return hipGetLastError() == hipSuccess ? TIP_OK : TIP_ERR_HIP;
}

int tip_stream_ingest(void* state, const float* raw_imu, int n_streams, int frame_idx, float* x_imu, float* x_s,
                      tip_stream_t stream) {
    return stream_ingest(state, raw_imu, n_streams, frame_idx, x_imu, x_s, stream, 0);
}

int tip_stream_ingest_newest(void* state, const float* raw_imu, int n_streams, int frame_idx, float* x_imu, float* x_s,
                             tip_stream_t stream) {
    return stream_ingest(state, raw_imu, n_streams, frame_idx, x_imu, x_s, stream, 1);
}

int tip_stream_consume(void* state, const float* y_last, int n_streams, int call_idx, float* s_rest, float* c_t,
                       tip_stream_t stream) {
    if (!state || !y_last || !s_rest || !c_t || n_streams < 0 || call_idx < TIP_STREAM_FRAME_AUTO) return TIP_ERR_INVALID_ARG;
    if (n_streams == 0) return TIP_OK;
    hipLaunchKernelGGL(stream_consume_kernel, dim3(n_streams), dim3(192), 0, static_cast<hipStream_t>(stream),
                       static_cast<float*>(state), y_last, n_streams, call_idx, s_rest, c_t);
    return hipGetLastError() == hipSuccess ? TIP_OK : TIP_ERR_HIP;
}

}  // extern "C"

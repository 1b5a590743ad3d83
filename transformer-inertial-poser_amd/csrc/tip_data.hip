// tip_data.hip — train-set combiner and window gather on the GPU.
//
// Reference: /root/reference/preprocess_and_combine_syn_amass.py:73-101 (per motion file: trim, 11-tap acceleration
// average with edge replication, constant bias, root-local IMU frame, running 40-frame acceleration sum / 15,
// axis-angle -> first two rotation-matrix columns, SBP channels) and /root/reference/training_data_loader.py:53-58,72-86
// (one training sample = three overlapping slices of the combined arrays ending at a sampled frame t).
//
// The reference does this once on the CPU in numpy fp64 and stores float32; here the combined arrays are produced on and
// stay on the GPU (they are a few GB for all of AMASS — nothing next to 288 GB of HBM), and a batch is an index gather
// from them straight into the tensors tip_train_forward reads: no host window materialisation (the reference's
// TrainSubDataset copies every sampled window into host memory first), no H2D per batch.
// All arithmetic is fp64 like the reference's, results are rounded to float32 at the store: HBM-bound byte work.
#include "tip_internal.h"

namespace tip {

constexpr int kAccTaps = 5;      // constants.py:15-16: 2*5+1 = 11-tap moving average
constexpr int kSumWin = 40;      // constants.py:17
constexpr double kSumScale = 15.0;   // constants.py:18
constexpr int kTrim = 4;         // preprocess_and_combine_syn_amass.py:73-75

__device__ __forceinline__ void inv3(const double* m, double* o) {
    const double c00 = m[4] * m[8] - m[5] * m[7], c01 = m[5] * m[6] - m[3] * m[8], c02 = m[3] * m[7] - m[4] * m[6];
    const double det = m[0] * c00 + m[1] * c01 + m[2] * c02;
    const double id = 1.0 / det;
    o[0] = c00 * id; o[1] = (m[2] * m[7] - m[1] * m[8]) * id; o[2] = (m[1] * m[5] - m[2] * m[4]) * id;
    o[3] = c01 * id; o[4] = (m[0] * m[8] - m[2] * m[6]) * id; o[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    o[6] = c02 * id; o[7] = (m[1] * m[6] - m[0] * m[7]) * id; o[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

// One thread per kept frame f (source frame f + 4): filtered + biased accelerations (:81-85), root-local rotations and
// accelerations (:86, data_utils.py:190-219).  Writes the float32 IMU row and the fp64 local accelerations (scratch).
// A frame is a 576-byte fp64 row: read per thread that is a 576-byte lane stride.  So the block's 128 rows (+ the
// accelerations of 5 halo rows each side for the 11-tap filter) are first copied to LDS with unit-stride loads, at odd row
// strides so that the per-thread row reads spread over all banks; results go back through LDS the same way (unit-stride
// stores).
constexpr int kCombFrames = 128;
constexpr int kCombRows = kCombFrames + 2 * kAccTaps;   // 138 rows of accelerations (with the filter halo)
constexpr int kRotLd = 55, kAccLd = 19;                  // staged row strides in doubles (odd: bank-conflict free)
constexpr int kCombLdsDoubles = kCombFrames * kRotLd + kCombRows * kAccLd;   // 9 662 doubles = 77.3 KB: two blocks per CU

constexpr int kCombThreads = 256;   // all of them copy; the first 128 each own a frame

__global__ __launch_bounds__(kCombThreads) void combine_imu_kernel(const double* __restrict__ imu, const double* __restrict__ bias, int Lp,
                                                                  float* __restrict__ imu_out, double* __restrict__ loc_acc) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    double* rot = lds;                              // [128][55]: the six rotation matrices of each frame
    double* accs = lds + kCombFrames * kRotLd;      // [138][19]: accelerations of frames f0-5 .. f0+132 (edge-replicated)
    const int f0 = blockIdx.x * kCombFrames;
    for (int i = threadIdx.x; i < kCombRows * 36; i += kCombThreads) {     // 16-byte loads: 36 double2 per row
        const int r = i / 36, c = (i - r * 36) * 2;
        const int fr = f0 + r - kAccTaps;
        const int g = fr < 0 ? 0 : (fr > Lp - 1 ? Lp - 1 : fr);     // mode="nearest" inside the trimmed sequence
        const bool body = r >= kAccTaps && r < kAccTaps + kCombFrames;
        if (c >= 54 || body) {
            const double2 v = *reinterpret_cast<const double2*>(imu + (size_t)(g + kTrim) * 72 + c);
            if (c >= 54) {
                accs[r * kAccLd + (c - 54)] = v.x;
                accs[r * kAccLd + (c - 53)] = v.y;
            } else {
                rot[(r - kAccTaps) * kRotLd + c] = v.x;
                rot[(r - kAccTaps) * kRotLd + c + 1] = v.y;
            }
        }
    }
    __syncthreads();
    const int t = threadIdx.x & (kCombFrames - 1);       // threads 128..255 shadow a frame but never publish it
    const bool owner = threadIdx.x < kCombFrames;
    const double* row = rot + t * kRotLd;
    double R[9], inv[9], la[18];
    float o[72];
#pragma unroll
    for (int i = 0; i < 9; ++i) R[i] = row[i];
    inv3(R, inv);
#pragma unroll
    for (int i = 0; i < 9; ++i) o[i] = (float)R[i];
#pragma unroll
    for (int sidx = 0; sidx < 5; ++sidx) {
        double m[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) m[i] = row[9 + 9 * sidx + i];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int k = 0; k < 3; ++k)
                o[9 + 9 * sidx + 3 * i + k] = (float)(inv[3 * i] * m[k] + inv[3 * i + 1] * m[3 + k] + inv[3 * i + 2] * m[6 + k]);
    }
    double acc[18];
#pragma unroll
    for (int ch = 0; ch < 18; ++ch) acc[ch] = 0.0;
    for (int k = 0; k <= 2 * kAccTaps; ++k) {
        const double* a = accs + (t + k) * kAccLd;
#pragma unroll
        for (int ch = 0; ch < 18; ++ch) acc[ch] += a[ch];
    }
#pragma unroll
    for (int ch = 0; ch < 18; ++ch) acc[ch] = acc[ch] / (double)(2 * kAccTaps + 1) + bias[ch];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        la[ch] = acc[ch];
        o[54 + ch] = (float)acc[ch];
    }
#pragma unroll
    for (int sidx = 0; sidx < 5; ++sidx) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double v = inv[3 * i] * acc[3 + 3 * sidx] + inv[3 * i + 1] * acc[4 + 3 * sidx] + inv[3 * i + 2] * acc[5 + 3 * sidx];
            la[3 + 3 * sidx + i] = v;
            o[57 + 3 * sidx + i] = (float)v;
        }
    }
    __syncthreads();                                   // everyone is done with the staged inputs: reuse them for the outputs
    float* so = reinterpret_cast<float*>(lds);                         // [128][73] floats (37.4 KB)
    double* sl = lds + (kCombFrames * 73 + 1) / 2 + 1;                  // [128][19] doubles behind it
    if (owner) {
#pragma unroll
        for (int i = 0; i < 72; ++i) so[t * 73 + i] = o[i];
#pragma unroll
        for (int i = 0; i < 18; ++i) sl[t * kAccLd + i] = la[i];
    }
    __syncthreads();
    const int nf = min(kCombFrames, Lp - f0);
    for (int i = threadIdx.x; i < nf * 72; i += kCombThreads) {
        const int r = i / 72, c = i - r * 72;
        imu_out[(size_t)f0 * 72 + i] = so[r * 73 + c];
    }
    for (int i = threadIdx.x; i < nf * 18; i += kCombThreads) {
        const int r = i / 18, c = i - r * 18;
        loc_acc[(size_t)f0 * 18 + i] = sl[r * kAccLd + c];
    }
}

// running sum of the last <= 40 local accelerations / 15 (:90-93).  Block = 256 frames x 18 channels; the 295 rows the
// block needs are staged in LDS once instead of being re-read 40 times from L2.
constexpr int kSumFrames = 256;

__global__ __launch_bounds__(256) void acc_sum_kernel(const double* __restrict__ loc_acc, int Lp, float* __restrict__ sum_out) {
    __shared__ double win[(kSumFrames + kSumWin - 1) * 18];
    const int f0 = blockIdx.x * kSumFrames;
    const int g0 = f0 - (kSumWin - 1);
    const int nrows = min(kSumFrames, Lp - f0) + kSumWin - 1;
    for (int i = threadIdx.x; i < nrows * 18; i += 256) {
        const int g = g0 + i / 18;
        win[i] = g >= 0 ? loc_acc[(long long)g0 * 18 + i] : 0.0;       // frames before the sequence start contribute nothing
    }
    __syncthreads();
    const int nf = min(kSumFrames, Lp - f0);
    for (int i = threadIdx.x; i < nf * 18; i += 256) {
        const int r = i / 18, ch = i - r * 18;
        double a = 0.0;
        // ascending frame order, as np.cumsum accumulates
        for (int k = 0; k < kSumWin; ++k) a += win[(r + k) * 18 + ch];
        sum_out[(size_t)f0 * 18 + i] = (float)(a / kSumScale);
    }
}

// S row (:96,:99-100,:127-128): 18 joints x first two columns of R(axis-angle) | root velocity | 20 SBP channels.
// One thread per (frame, slot): slots 0..17 = joints, slot 18 = root velocity + SBP copy.
__global__ __launch_bounds__(256) void s_2axis_kernel(const double* __restrict__ s, const double* __restrict__ c, int Lp,
                                                      int nan_root_vel, float* __restrict__ s_out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= Lp * 19) return;
    const int f = i / 19, slot = i - f * 19;
    const double* row = s + (size_t)(f + kTrim) * 114;
    float* out = s_out + (size_t)f * 131;
    if (slot == 18) {
#pragma unroll
        for (int k = 0; k < 3; ++k) out[108 + k] = nan_root_vel ? __builtin_nanf("") : (float)row[57 + k];
        const double* cr = c + (size_t)(f + kTrim) * 20;
#pragma unroll
        for (int k = 0; k < 20; ++k) out[111 + k] = (float)cr[k];
        return;
    }
    // scipy Rotation.from_rotvec -> unit quaternion -> matrix (fairmotion A2R)
    const double x = row[3 + 3 * slot], y = row[4 + 3 * slot], z = row[5 + 3 * slot];
    const double ang = sqrt(x * x + y * y + z * z);
    double sc;
    if (ang <= 1e-3) {
        const double a2 = ang * ang;
        sc = 0.5 - a2 / 48.0 + a2 * a2 / 3840.0;
    } else {
        sc = sin(ang * 0.5) / ang;
    }
    const double qx = x * sc, qy = y * sc, qz = z * sc, qw = cos(ang * 0.5);
    const double xx = qx * qx, yy = qy * qy, zz = qz * qz, ww = qw * qw;
    const double xy = qx * qy, zw = qz * qw, xz = qx * qz, yw = qy * qw, yz = qy * qz, xw = qx * qw;
    float* o = out + 6 * slot;   // (3 x 2) row-major: R00 R01 R10 R11 R20 R21
    o[0] = (float)(xx - yy - zz + ww);
    o[1] = (float)(2.0 * (xy - zw));
    o[2] = (float)(2.0 * (xy + zw));
    o[3] = (float)(-xx + yy - zz + ww);
    o[4] = (float)(2.0 * (xz - yw));
    o[5] = (float)(2.0 * (yz + xw));
}

// one workgroup per sampled window: x_imu[n] = [IMU[t-T:t] | SUM[t-T:t]], x_s[n] = S[t-T:t], y[n] = S[t-T+1:t+1]
__global__ __launch_bounds__(256) void gather_windows_kernel(const float* __restrict__ imu_c, const float* __restrict__ sum_c,
                                                             const float* __restrict__ s_c, const long long* __restrict__ t_idx,
                                                             int T, float* __restrict__ x_imu, float* __restrict__ x_s,
                                                             float* __restrict__ y) {
    const int n = blockIdx.x;
    const long long t0 = t_idx[n] - T;
    const int wi = sum_c ? 90 : 72;
    float* xi = x_imu + (size_t)n * T * wi;
    for (int i = threadIdx.x; i < T * wi; i += blockDim.x) {
        const int r = i / wi, cc = i - r * wi;
        xi[i] = cc < 72 ? imu_c[(size_t)(t0 + r) * 72 + cc] : sum_c[(size_t)(t0 + r) * 18 + (cc - 72)];
    }
    // S rows t0 .. t0+T are contiguous in memory: x_s = the first T rows, y = the last T rows
    const float* sp = s_c + (size_t)t0 * 131;
    float* xs = x_s + (size_t)n * T * 131;
    float* yo = y + (size_t)n * T * 131;
    for (int i = threadIdx.x; i < (T + 1) * 131; i += blockDim.x) {
        const float v = sp[i];
        if (i < T * 131) xs[i] = v;
        if (i >= 131) yo[i - 131] = v;
    }
}

}  // namespace tip

using namespace tip;

extern "C" {

int tip_combine_frames(int L_imu, int L_s) {
    const int m = L_imu < L_s ? L_imu : L_s;
    return m <= kSumWin ? 0 : m - 2 * kTrim;     // "too short" files are skipped (:68-70)
}

int tip_combine_scratch_bytes(int L_imu, int L_s, size_t* bytes) {
    if (!bytes || L_imu < 0 || L_s < 0) return TIP_ERR_INVALID_ARG;
    *bytes = (size_t)(tip_combine_frames(L_imu, L_s) > 0 ? tip_combine_frames(L_imu, L_s) : 1) * 18 * sizeof(double);
    return TIP_OK;
}

int tip_combine_sequence(const double* imu, const double* s, const double* c, int L_imu, int L_s, const double* bias,
                         int nan_root_vel, float* imu_out, float* sum_out, float* s_out, void* scratch, size_t scratch_bytes,
                         void* stream) {
    if (!imu || !s || !c || !bias || !imu_out || !sum_out || !s_out || !scratch) return TIP_ERR_INVALID_ARG;
    if (abs(L_imu - L_s) > 1) return TIP_ERR_INVALID_ARG;             // the reference asserts this (:66)
    const int Lp = tip_combine_frames(L_imu, L_s);
    if (Lp <= 0) return 0;
    if (scratch_bytes < (size_t)Lp * 18 * sizeof(double) || reinterpret_cast<uintptr_t>(scratch) % 8) return TIP_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* la = static_cast<double*>(scratch);
    constexpr size_t comb_lds = (size_t)kCombLdsDoubles * sizeof(double);
    static PerDeviceFlag attr_flag; bool& attr_set = attr_flag.cur();
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(combine_imu_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)comb_lds) != hipSuccess)
            return TIP_ERR_HIP;
        attr_set = true;
    }
    hipLaunchKernelGGL(combine_imu_kernel, dim3((Lp + kCombFrames - 1) / kCombFrames), dim3(kCombThreads), comb_lds, st, imu, bias, Lp,
                       imu_out, la);
    hipLaunchKernelGGL(acc_sum_kernel, dim3((Lp + kSumFrames - 1) / kSumFrames), dim3(256), 0, st, la, Lp, sum_out);
    hipLaunchKernelGGL(s_2axis_kernel, dim3((Lp * 19 + 255) / 256), dim3(256), 0, st, s, c, Lp, nan_root_vel, s_out);
    if (hipGetLastError() != hipSuccess) return TIP_ERR_HIP;
    return Lp;
}

int tip_gather_windows(const float* imu_c, const float* sum_c, const float* s_c, long long n_frames, const long long* t_idx,
                       int n, int T, float* x_imu, float* x_s, float* y, void* stream) {
    if (!imu_c || !s_c || !t_idx || !x_imu || !x_s || !y || n < 0 || T < 1 || n_frames < T + 1) return TIP_ERR_INVALID_ARG;
    if (n == 0) return TIP_OK;
    hipLaunchKernelGGL(gather_windows_kernel, dim3(n), dim3(256), 0, static_cast<hipStream_t>(stream), imu_c, sum_c, s_c, t_idx, T,
                       x_imu, x_s, y);
    return hipGetLastError() == hipSuccess ? TIP_OK : TIP_ERR_HIP;
}

}  // extern "C"

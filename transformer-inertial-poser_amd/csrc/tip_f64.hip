// tip_f64.hip — the forward pass in fp64 (the reference's `--double` switch, /root/reference/train_model.py:62-63,84-85,161-164:
// torch.set_default_dtype(torch.float64) builds the module with fp64 parameters and feeds it fp64 windows).
//
// Same function as tip_forward (/root/reference/simple_transformer_with_state.py:60-102), every operation in IEEE double:
// the parameters are read RAW (fp64 device tensors in state-dict order, as tip_train_forward reads its fp32 ones — nothing is
// packed), the big products run on the fp64 matrix cores (v_mfma_f64_16x16x4_f64, 64x64 tiles staged through LDS), softmax /
// LayerNorm / tanh use the double-precision device math library.  `--double` is a debugging switch of the training script, so
// this path is built for exactness and any configuration, layer by layer.  Round 4 took the obvious costs out (the recurrence as
// one fp64-MFMA launch per time step over all windows instead of one workgroup per window, attention per (window, head) in LDS,
// parallel column sums, GEMM operand loads under the MFMAs): forward 5.55 -> 2.69 ms, forward + backward 17.4 -> 8.9 ms at B = 256,
// T = 40 — 1.7x / 3.5x faster than the same module on stock PyTorch-ROCm fp64 ops (tools/f64_bench.py); the GEMMs sit at ~58 % of
// the fp64 matrix peak and are what is left.  It is also the on-device high-precision check of the fp32 plans at sizes the CPU
// oracle does not finish in seconds (tests/test_f64_gpu.py).
#include <hip/hip_runtime.h>
#include <string>
#include "../../include/tip_hip.h"
#include "tip_internal.h"

namespace tip {
namespace f64 {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int TM = 64, TN = 64, KC = 16, LDS_LD = KC + 1;

// :63-78 — clone, NaN scrub, root-velocity columns zeroed, past-state keep mask, concat: U [M][In]
__global__ void prologue_kernel(const double* __restrict__ x_imu, const double* __restrict__ x_s, const double* __restrict__ keep_mask,
                                double keep_scale, double* __restrict__ U, long long M, int NI, int S, int z0, int z1) {
    const int In = NI + S;
    const long long n = M * In;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / In;
        const int c = (int)(i - r * In);
        double v;
        if (c < NI) v = x_imu[r * NI + c];
        else {
            const int cs = c - NI;
            v = x_s[r * S + cs];
            if (v != v) v = 0.0;                                     // :65
            if (cs >= z0 && cs < z1) v *= 0.0;                       // :75
            if (keep_mask) v = v * keep_mask[r * S + cs] * keep_scale;   // :77
        }
        U[i] = v;
    }
}

// C[M][ldc] = epi(A[M][K] (row stride lda) * W[N][K]^T + bias (+ res)).  FLAGS: 1 ReLU, 2 residual, 4 the head-interleave
// shuffle of :88-89 on the store (output channel n = b*dh + a lands in column a*H + b).
template <int FLAGS>
__global__ __launch_bounds__(256) void gemm_kernel(const double* __restrict__ A, long long lda, const double* __restrict__ W,
                                                   const double* __restrict__ bias, const double* res, int ldres, double* C, int ldc,
                                                   int M, int N, int K, int H, int dh) {
    __shared__ double As[TM * LDS_LD], Ws[TN * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const long long row0 = (long long)blockIdx.y * TM;
    const int col0 = blockIdx.x * TN;
    d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
    const int lr = tid >> 2, lk = (tid & 3) * 4;   // this thread stages 4 consecutive k of row lr of both operands
    double av[4], wv[4];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = k0 + lk + q;
            av[q] = (row0 + lr < M && k < K) ? A[(row0 + lr) * lda + k] : 0.0;
            wv[q] = (col0 + lr < N && k < K) ? W[(long long)(col0 + lr) * K + k] : 0.0;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += KC) {
        __syncthreads();   // the previous chunk's fragments have been read
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            As[lr * LDS_LD + lk + q] = av[q];
            Ws[lr * LDS_LD + lk + q] = wv[q];
        }
        __syncthreads();
        if (k0 + KC < K) fetch(k0 + KC);   // the next chunk's operands travel during this chunk's MFMAs (round 4)
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
            double a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = As[(wr * 32 + i * 16 + l15) * LDS_LD + 4 * s + lg];
                b[i] = Ws[(wc * 32 + i * 16 + l15) * LDS_LD + 4 * s + lg];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    // accumulator layout of the f64 16x16x4 (tools/probes/mfma_f64_probe.hip): lane (l15, lg) holds column l15, rows 4 e + lg of its
    // 16x16 tile (NOT 4 lg + e as the f32 16x16x4 does); operands as the f32 instruction: lane = (row / column l15, k = lg)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = col0 + wc * 32 + j * 16 + l15;
            if (n >= N) continue;
            const double bv = bias ? bias[n] : 0.0;
            const int cn = (FLAGS & 4) ? (n % dh) * H + n / dh : n;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const long long r = row0 + wr * 32 + i * 16 + e * 4 + lg;
                if (r >= M) continue;
                double v = acc[i][j][e] + bv;
                if (FLAGS & 1) v = v > 0.0 ? v : 0.0;
                if (FLAGS & 2) v = res[r * ldres + cn] + v;
                C[r * ldc + cn] = v;
            }
        }
}

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v, o, 64); v = t > v ? t : v; }
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// 16-head causal scaled-dot-product attention (:85,91; torch multi_head_attention_forward): one wave per (window, head, query)
__global__ __launch_bounds__(64) void attention_kernel(const double* __restrict__ qkv, double* __restrict__ att, int T, int D, int H,
                                                      int dh, double scale) {
    extern __shared__ double p[];   // [T]
    const int lane = threadIdx.x;
    const long long u = blockIdx.x;
    const int i = (int)(u % T);
    const int hd = (int)((u / T) % H);
    const long long b = u / ((long long)T * H);
    const double* base = qkv + b * T * 3 * D;
    const double* q = base + (long long)i * 3 * D + hd * dh;
    double mx = -INFINITY;
    for (int j = lane; j <= i; j += 64) {
        const double* k = base + (long long)j * 3 * D + D + hd * dh;
        double s = 0.0;
        for (int e = 0; e < dh; ++e) s += (q[e] * scale) * k[e];
        p[j] = s;
        mx = s > mx ? s : mx;
    }
    mx = wave_max(mx);
    double den = 0.0;
    for (int j = lane; j <= i; j += 64) {
        const double e_ = exp(p[j] - mx);
        p[j] = e_;
        den += e_;
    }
    den = wave_sum(den);
    __syncthreads();
    for (int e = lane; e < dh; e += 64) {
        double o = 0.0;
        for (int j = 0; j <= i; ++j) o += (p[j] / den) * base[(long long)j * 3 * D + 2 * D + hd * dh + e];
        att[(b * T + i) * D + hd * dh + e] = o;
    }
}

// nn.LayerNorm(D, eps = 1e-5), biased variance, in place; one wave per row
__global__ __launch_bounds__(64) void layernorm_kernel(double* __restrict__ Z, const double* __restrict__ g, const double* __restrict__ be, int D) {
    double* x = Z + (long long)blockIdx.x * D;
    const int lane = threadIdx.x;
    double s = 0.0;
    for (int c = lane; c < D; c += 64) s += x[c];
    const double mean = wave_sum(s) / D;
    double v = 0.0;
    for (int c = lane; c < D; c += 64) { const double d = x[c] - mean; v += d * d; }
    const double rstd = 1.0 / sqrt(wave_sum(v) / D + 1e-5);
    for (int c = lane; c < D; c += 64) x[c] = (x[c] - mean) * rstd * g[c] + be[c];
}

__global__ void transpose_kernel(const double* __restrict__ W, double* __restrict__ Wt, int R) {
    const long long n = (long long)R * R;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i / R), c = (int)(i - (long long)k * R);
        Wt[i] = W[(long long)c * R + k];   // Wt[k][c] = W[c][k]
    }
}

// :98-99 nn.RNN tanh, h0 = 0: h_t = tanh((W_ih x_t + b_ih) + (W_hh h_{t-1} + b_hh)); IH = the first bracket for every row.
struct DropD {
    unsigned key, thresh;
    double scale;
};
static DropD make_drop_d(float p, unsigned long long seed, unsigned site) {
    DropD d;
    d.key = tip_drop_key(seed, site);
    if (p <= 0.f) { d.thresh = 0; d.scale = 1.0; return d; }
    double t = (double)p * 4294967296.0;
    if (t > 4294967295.0) t = 4294967295.0;
    d.thresh = (unsigned)t;
    if (d.thresh == 0) d.thresh = 1;
    d.scale = (double)(1.0f / (1.0f - p));
    return d;
}
__device__ __forceinline__ double drop_factor(const DropD& d, unsigned long long idx) {
    if (!d.thresh) return 1.0;
    return tip_drop_hash_k(d.key, (unsigned)idx) >= d.thresh ? d.scale : 0.0;
}

// Round 4: forward attention with one workgroup per (window, head): q | k | v of the head staged in LDS (odd row strides: a thread per
// query row walks its row without bank conflicts, every thread of an iteration reads the same key row: broadcast), thread i the
// softmax of query i over keys 0 .. i, then thread (row, channel) the P V sum.  Same expressions, same order per element as the
// wave-per-query kernels it replaces where it fits (the row's denominator is summed in key order instead of by a wave butterfly);
// 330 us -> ~25 us per layer at B = 256, T = 40.  DROP: dropout on the probabilities (site 4 l), the training forward.
template <bool DROP>
__global__ __launch_bounds__(256) void attention_wg_kernel(const double* __restrict__ qkv, double* __restrict__ att, int T, int D, int H, int dh,
                                                           double scale, DropD dr) {
    extern __shared__ double sm[];
    const int ldq = dh | 1, ldp = T | 1;
    double* qs = sm;
    double* ks = qs + T * ldq;
    double* vs = ks + T * ldq;
    double* P = vs + T * ldq;
    double* rmx = P + T * ldp;     // [T] row maximum, then 1 / row sum
    const int hd = blockIdx.x % H;
    const long long b = blockIdx.x / H;
    const double* base = qkv + b * T * 3 * D;
    for (int i = threadIdx.x; i < T * dh; i += blockDim.x) {
        const int r = i / dh, e = i - r * dh;
        qs[r * ldq + e] = base[(long long)r * 3 * D + hd * dh + e] * scale;
        ks[r * ldq + e] = base[(long long)r * 3 * D + D + hd * dh + e];
        vs[r * ldq + e] = base[(long long)r * 3 * D + 2 * D + hd * dh + e];
    }
    __syncthreads();
    // scores of the lower triangle, one element per thread and pass (the expensive parts — dot product, exp, division, dropout hash —
    // are spread over the whole workgroup; only the order-sensitive row maximum / row sum are walked by one thread per row)
    for (int idx = threadIdx.x; idx < T * T; idx += blockDim.x) {
        const int i = idx / T, j = idx - i * T;
        if (j > i) continue;
        double sc = 0.0;
        for (int e = 0; e < dh; ++e) sc += qs[i * ldq + e] * ks[j * ldq + e];
        P[i * ldp + j] = sc;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        double mx = -INFINITY;
        for (int j = 0; j <= i; ++j) { const double sc = P[i * ldp + j]; mx = sc > mx ? sc : mx; }
        rmx[i] = mx;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < T * T; idx += blockDim.x) {
        const int i = idx / T, j = idx - i * T;
        if (j > i) continue;
        P[i * ldp + j] = exp(P[i * ldp + j] - rmx[i]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        double den = 0.0;
        for (int j = 0; j <= i; ++j) den += P[i * ldp + j];
        rmx[i] = den;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < T * T; idx += blockDim.x) {
        const int i = idx / T, j = idx - i * T;
        if (j > i) continue;
        double pv = P[i * ldp + j] / rmx[i];
        if (DROP) pv *= drop_factor(dr, (unsigned long long)((b * H + hd) * T + i) * T + j);
        P[i * ldp + j] = pv;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T * dh; i += blockDim.x) {
        const int r = i / dh, e = i - r * dh;
        double o = 0.0;
        for (int j = 0; j <= r; ++j) o += P[r * ldp + j] * vs[j * ldq + e];
        att[(b * T + r) * D + hd * dh + e] = o;
    }
}
static size_t attention_wg_smem(int T, int dh) { return (size_t)(3 * T * (dh | 1) + T * (T | 1) + T) * sizeof(double); }
static bool attention_wg_ok(int T, int dh) { return T <= 128 && attention_wg_smem(T, dh) <= 64 * 1024; }

// Round 4: the recurrence as ONE LAUNCH PER TIME STEP over all windows — H_t = tanh(IH_t + (H_{t-1} W_hh^T + b_hh)) is a
// [B x R] x [R x R] product, 16-window x 32-channel tiles on the fp64 matrix cores, operands straight from L2 (32-byte row segments
// per lane, a whole k-quarter's loads in flight at once), the four k-quarters of a tile on four waves and summed through LDS in a fixed order.  (Until round 4: one
// workgroup per window, W_hh^T — 2 MB at R = 512 — streamed through every CU at every step, R dependent FMAs per thread: 2.28 ms at
// B = 256, T = 40 against 0.58 ms for 40 of these launches.)  BWD: delta_t = (dH_t + delta_{t+1} W_hh) (1 - h_t^2), Wm = W_hh^T, and the
// h_{t-1} rows are copied next to it (dW_hh's operand).  R % 64 == 0: tip_create admits nothing else.  `add` / `side` may alias (the backward writes h_{t-1} where it just read dH_t): no __restrict__ there.
template <bool BWD>
__global__ __launch_bounds__(512) void rnn_step_kernel(const double* prev, const double* __restrict__ Wm, const double* __restrict__ bias,
                                                       const double* add, const double* __restrict__ hcur, const double* __restrict__ hprev_src,
                                                       double* out, double* side, long long ld, int B, int R, int first) {
    __shared__ double red[3 * 2 * 64 * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int ct = wave & 1, kq = wave >> 1;                  // channel tile, k-quarter
    const int w0 = blockIdx.y * 16, c0 = blockIdx.x * 32 + ct * 16;
    d4 acc = {0.0, 0.0, 0.0, 0.0};
    if (!first) {
        const int nkb = R / 64;                              // 16-wide k-blocks of this wave's quarter
        const int wrow = w0 + l15 < B ? w0 + l15 : B - 1;    // (padded rows of a ragged last tile re-read the last window: never stored)
        const double* pa = prev + (long long)wrow * ld + kq * (R / 4) + lg * 4;
        const double* pb = Wm + (long long)(c0 + l15) * R + kq * (R / 4) + lg * 4;
        constexpr int G = 8;                                 // k-blocks whose operands are in flight together (R = 512: the whole quarter)
        for (int kb0 = 0; kb0 < nkb; kb0 += G) {
            d4 a[G], b[G];
#pragma unroll
            for (int j = 0; j < G; ++j) {
                const int kb = kb0 + j < nkb ? kb0 + j : nkb - 1;
                a[j] = *reinterpret_cast<const d4*>(pa + kb * 16);
                b[j] = *reinterpret_cast<const d4*>(pb + kb * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < G; ++j)
                if (kb0 + j < nkb) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j][q], b[j][q], acc, 0, 0, 0);
                }
        }
    }
    if (kq > 0) *reinterpret_cast<d4*>(red + (((kq - 1) * 2 + ct) * 64 + lane) * 4) = acc;
    __syncthreads();
    if (kq > 0) return;
    {   // (q0 + q1) + (q2 + q3)
        const d4 p1 = *reinterpret_cast<const d4*>(red + ((0 * 2 + ct) * 64 + lane) * 4);
        const d4 p2 = *reinterpret_cast<const d4*>(red + ((1 * 2 + ct) * 64 + lane) * 4);
        const d4 p3 = *reinterpret_cast<const d4*>(red + ((2 * 2 + ct) * 64 + lane) * 4);
        acc = (acc + p1) + (p2 + p3);
    }
    const int c = c0 + l15;
    const double bv = BWD ? 0.0 : bias[c];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int w = w0 + 4 * e + lg;                       // fp64 16x16x4 accumulator: register e = row 4 e + lg, column l15
        if (w >= B) continue;
        const long long i = (long long)w * ld + c;
        if (BWD) {
            const double h = hcur[i];
            const double v = (add[i] + acc[e]) * (1.0 - h * h);
            out[i] = v;
            side[i] = hprev_src ? hprev_src[i] : 0.0;
        } else {
            out[i] = tanh(add[i] + (acc[e] + bv));
        }
    }
}
// forward: HALL[:, t] for t = 0 .. T-1 (IH, HALL: [B][T][R])
static hipError_t rnn_steps_fwd(const double* IH, const double* Whh, const double* bhh, double* HALL, int B, int T, int R, hipStream_t s) {
    const dim3 grid(R / 32, (B + 15) / 16);
    const long long ld = (long long)T * R;
    for (int t = 0; t < T; ++t)
        hipLaunchKernelGGL(rnn_step_kernel<false>, grid, dim3(512), 0, s, t ? HALL + (size_t)(t - 1) * R : HALL, Whh, bhh, IH + (size_t)t * R,
                           (const double*)nullptr, (const double*)nullptr, HALL + (size_t)t * R, (double*)nullptr, ld, B, R, t == 0 ? 1 : 0);
    return hipGetLastError();
}
// backward: delta[:, t] for t = T-1 .. 0; hprev lands in dH's place (dH[:, t] is read before it is overwritten, same thread)
static hipError_t rnn_steps_bwd(double* dH, const double* WhhT, const double* HALL, double* delta, int B, int T, int R, hipStream_t s) {
    const dim3 grid(R / 32, (B + 15) / 16);
    const long long ld = (long long)T * R;
    for (int t = T - 1; t >= 0; --t)
        hipLaunchKernelGGL(rnn_step_kernel<true>, grid, dim3(512), 0, s, t + 1 < T ? delta + (size_t)(t + 1) * R : delta, WhhT, (const double*)nullptr,
                           dH + (size_t)t * R, HALL + (size_t)t * R, t ? HALL + (size_t)(t - 1) * R : (const double*)nullptr, delta + (size_t)t * R,
                           dH + (size_t)t * R, ld, B, R, t == T - 1 ? 1 : 0);
    return hipGetLastError();
}

struct Layout {
    size_t U, Z, QKV, ATT, HID, IH, HALL, WT, total;   // double offsets
};

static Layout layout(const Dims& d, int B, int T) {
    Layout L;
    const size_t M = (size_t)B * T;
    auto al = [](size_t v) { return (v + 31) & ~(size_t)31; };
    size_t o = 0;
    L.U = o;    o += al(M * d.In);
    L.Z = o;    o += al(M * d.D);
    L.QKV = o;  o += al(M * 3 * d.D);
    L.ATT = o;  o += al(M * d.D);
    L.HID = o;  o += al(M * d.F);
    L.IH = o;   o += al(d.with_rnn ? M * d.R : 0);
    L.HALL = o; o += al(d.with_rnn ? M * d.R : 0);
    L.WT = o;   // (unused since the per-step recurrence reads W_hh as it is; kept so that the workspace size is unchanged)
    o += al(d.with_rnn ? (size_t)d.R * d.R : 0);
    L.total = o;
    return L;
}

static hipError_t gemm(int flags, const double* A, long long lda, const double* W, const double* bias, const double* res, int ldres,
                       double* C, int ldc, long long M, int N, int K, int H, int dh, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    const dim3 grid((N + TN - 1) / TN, (unsigned)((M + TM - 1) / TM)), block(256);
    switch (flags) {
        case 0: hipLaunchKernelGGL(gemm_kernel<0>, grid, block, 0, s, A, lda, W, bias, res, ldres, C, ldc, (int)M, N, K, H, dh); break;
        case 1: hipLaunchKernelGGL(gemm_kernel<1>, grid, block, 0, s, A, lda, W, bias, res, ldres, C, ldc, (int)M, N, K, H, dh); break;
        case 2: hipLaunchKernelGGL(gemm_kernel<2>, grid, block, 0, s, A, lda, W, bias, res, ldres, C, ldc, (int)M, N, K, H, dh); break;
        case 4: hipLaunchKernelGGL(gemm_kernel<4>, grid, block, 0, s, A, lda, W, bias, res, ldres, C, ldc, (int)M, N, K, H, dh); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace f64
}  // namespace tip

using namespace tip;

extern "C" {

int tip_forward_f64_bytes(const tip_handle* h, int B, int T, size_t* bytes) {
    if (!h || !bytes || B < 0 || T < 0) return TIP_ERR_INVALID_ARG;
    *bytes = f64::layout(h->d, B, T).total * sizeof(double) + 256;
    return TIP_OK;
}

int tip_forward_f64(tip_handle* h, const double* const* params, int n_params, const double* x_imu, const double* x_s, double* y,
                    int B, int T, int flags, const double* keep_mask, double keep_scale, void* workspace, size_t workspace_bytes,
                    void* stream) {
    if (!h || !params || B < 0 || T < 0) return TIP_ERR_INVALID_ARG;
    if (n_params != (int)h->tensor_names.size()) return TIP_ERR_INVALID_ARG;
    if (B == 0 || T == 0) return TIP_OK;
    if (!x_imu || !x_s || !y || !workspace) return TIP_ERR_INVALID_ARG;
    const Dims& d = h->d;
    const long long M = (long long)B * T;
    if (M * (long long)(3 * d.D > d.F ? 3 * d.D : d.F) > 0x7fffffffLL || (long long)B * T * d.H > 0x7fffffffLL || M > 65535LL * f64::TM)
        return TIP_ERR_UNSUPPORTED_CONFIG;   // tip_max_batch(h, T, 1, &b) gives the largest B served
    const f64::Layout L = f64::layout(d, B, T);
    if (reinterpret_cast<uintptr_t>(workspace) % 256 || workspace_bytes < L.total * sizeof(double)) return TIP_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* W = static_cast<double*>(workspace);
    auto fail = [&](hipError_t e, const char* what) {
        h->last_hip_error = std::string(what) + ": " + hipGetErrorString(e);
        return (int)TIP_ERR_HIP;
    };
    // no cooperating kernel here, but these launches can hold CUs while a cooperating fp32 forward of another stream needs them
    // all: same cross-stream guard as tip_forward
    CoopSerial serial(h->device, s);
    if (serial.status != hipSuccess) return fail(serial.status, "stream serialisation");
#define TF(expr, what) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(e_, what); } while (0)
    // state-dict order (simple_transformer_with_state.py:20-46): in_linear W, b; per layer in_proj W, b, out_proj W, b, linear1 W, b,
    // linear2 W, b, norm1 g, b, norm2 g, b; rnn W_ih, W_hh, b_ih, b_hh; linear W, b
    const double* const* tw = params + 2 + 12 * d.L;
    {
        const long long n = M * d.In;
        const int grid = (int)((n + 255) / 256 < 65536 ? (n + 255) / 256 : 65536);
        hipLaunchKernelGGL(f64::prologue_kernel, dim3(grid), dim3(256), 0, s, x_imu, x_s, keep_mask, keep_scale, W + L.U, M,
                           d.n_imu_total, d.S, d.rootv0, d.rootv1);
        TF(hipGetLastError(), "f64 prologue");
    }
    TF(f64::gemm(4, W + L.U, d.In, params[0], params[1], nullptr, 0, W + L.Z, d.D, M, d.D, d.In, d.H, d.dh, s), "f64 in_linear");   // :79,88-89
    const double scale = 1.0 / sqrt((double)d.dh);
    for (int l = 0; l < d.L; ++l) {
        const double* const* lp = params + 2 + 12 * l;
        TF(f64::gemm(0, W + L.Z, d.D, lp[0], lp[1], nullptr, 0, W + L.QKV, 3 * d.D, M, 3 * d.D, d.D, 0, 1, s), "f64 in_proj");
        if (f64::attention_wg_ok(T, d.dh))
            hipLaunchKernelGGL(f64::attention_wg_kernel<false>, dim3((unsigned)(B * d.H)), dim3(256), f64::attention_wg_smem(T, d.dh), s, W + L.QKV,
                               W + L.ATT, T, d.D, d.H, d.dh, scale, f64::DropD{0u, 0u, 1.0});
        else
            hipLaunchKernelGGL(f64::attention_kernel, dim3((unsigned)(M * d.H)), dim3(64), (size_t)T * sizeof(double), s, W + L.QKV,
                               W + L.ATT, T, d.D, d.H, d.dh, scale);
        TF(hipGetLastError(), "f64 attention");
        TF(f64::gemm(2, W + L.ATT, d.D, lp[2], lp[3], W + L.Z, d.D, W + L.Z, d.D, M, d.D, d.D, 0, 1, s), "f64 out_proj");
        hipLaunchKernelGGL(f64::layernorm_kernel, dim3((unsigned)M), dim3(64), 0, s, W + L.Z, lp[8], lp[9], d.D);
        TF(hipGetLastError(), "f64 norm1");
        TF(f64::gemm(1, W + L.Z, d.D, lp[4], lp[5], nullptr, 0, W + L.HID, d.F, M, d.F, d.D, 0, 1, s), "f64 linear1");
        TF(f64::gemm(2, W + L.HID, d.F, lp[6], lp[7], W + L.Z, d.D, W + L.Z, d.D, M, d.D, d.F, 0, 1, s), "f64 linear2");
        hipLaunchKernelGGL(f64::layernorm_kernel, dim3((unsigned)M), dim3(64), 0, s, W + L.Z, lp[10], lp[11], d.D);
        TF(hipGetLastError(), "f64 norm2");
    }
    const bool last = (flags & TIP_FWD_LAST_ROW_ONLY) != 0;
    const double* feat = W + L.Z;
    int fw = d.D;
    if (d.with_rnn) {
        TF(f64::gemm(0, W + L.Z, d.D, tw[0], tw[2], nullptr, 0, W + L.IH, d.R, M, d.R, d.D, 0, 1, s), "f64 rnn W_ih");
        if (d.R % 64) return TIP_ERR_UNSUPPORTED_CONFIG;   // (tip_create admits multiples of 64 only)
        TF(f64::rnn_steps_fwd(W + L.IH, tw[1], tw[3], W + L.HALL, B, T, d.R, s), "f64 rnn steps");
        feat = W + L.HALL;
        fw = d.R;
        tw += 4;
    }
    // :102 self.linear on every row (or on row T-1 of every window)
    if (last) TF(f64::gemm(0, feat + (size_t)(T - 1) * fw, (long long)T * fw, tw[0], tw[1], nullptr, 0, y, d.S, B, d.S, fw, 0, 1, s), "f64 linear (last rows)");
    else TF(f64::gemm(0, feat, fw, tw[0], tw[1], nullptr, 0, y, d.S, M, d.S, fw, 0, 1, s), "f64 linear");
#undef TF
    ++h->forward_count;
    return TIP_OK;
}

}  // extern "C"

// =====================================================================================================================
// Training step in fp64: the model call of train_model.py:175 and its backward (:192) for a module built under `--double`
// (:84-85).  Same contract as tip_train_forward / tip_train_backward (include/tip_hip.h) with double tensors: raw fp64 parameters
// in state-dict order, the four dropout sites of every nn.TransformerEncoderLayer from the same counter-based hash (same keep
// decisions as the fp32 step for the same seed; kept values scaled by the fp32 value of 1 / (1 - p), widened), activations
// stashed in `saved`, gradients as one flat fp64 buffer in state-dict order.  Like the fp64 forward this is the debugging /
// verification path of the training script: built for exactness and any configuration (with or without the RNN, any widths),
// layer by layer, deterministic (fixed-order reductions), not tuned.
// =====================================================================================================================
namespace tip {
namespace f64 {

// y[i] = (res ? res[i] : 0) + x[i] * keep(i)   (dropout sites 1 / 3 with their residual; site 2 in place with res = null)
__global__ void drop_res_kernel(const double* __restrict__ x, const double* __restrict__ res, double* __restrict__ y, long long n, DropD d) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = (res ? res[i] : 0.0) + x[i] * drop_factor(d, (unsigned long long)i);
}

// attention with dropout on the probabilities (site 4 l): one wave per (window, head, query); P is recomputed by the backward
__global__ __launch_bounds__(64) void attention_train_kernel(const double* __restrict__ qkv, double* __restrict__ att, int T, int D, int H,
                                                            int dh, double scale, DropD dr) {
    extern __shared__ double p[];   // [T]
    const int lane = threadIdx.x;
    const long long u = blockIdx.x;
    const int i = (int)(u % T);
    const int hd = (int)((u / T) % H);
    const long long b = u / ((long long)T * H);
    const double* base = qkv + b * T * 3 * D;
    const double* q = base + (long long)i * 3 * D + hd * dh;
    double mx = -INFINITY;
    for (int j = lane; j <= i; j += 64) {
        const double* k = base + (long long)j * 3 * D + D + hd * dh;
        double s = 0.0;
        for (int e = 0; e < dh; ++e) s += (q[e] * scale) * k[e];
        p[j] = s;
        mx = s > mx ? s : mx;
    }
    mx = wave_max(mx);
    double den = 0.0;
    for (int j = lane; j <= i; j += 64) {
        const double e_ = exp(p[j] - mx);
        p[j] = e_;
        den += e_;
    }
    den = wave_sum(den);
    for (int j = lane; j <= i; j += 64)
        p[j] = (p[j] / den) * drop_factor(dr, (unsigned long long)((b * H + hd) * T + i) * T + j);
    __syncthreads();
    for (int e = lane; e < dh; e += 64) {
        double o = 0.0;
        for (int j = 0; j <= i; ++j) o += p[j] * base[(long long)j * 3 * D + 2 * D + hd * dh + e];
        att[(b * T + i) * D + hd * dh + e] = o;
    }
}

// attention backward: one workgroup per (window, head).  dqkv rows of this head <- d(att) of this head.
//   P_ij = softmax_j(q_i k_j * scale), Pd = P * keep; O = Pd V;  dPd_ij = dO_i . v_j;  dV_j = sum_i Pd_ij dO_i;
//   dP = dPd * keep;  dS_ij = P_ij (dP_ij - sum_k dP_ik P_ik);  dq_i = scale sum_j dS_ij k_j;  dk_j = scale sum_i dS_ij q_i
__global__ __launch_bounds__(256) void attention_bwd_kernel(const double* __restrict__ qkv, const double* __restrict__ datt, double* __restrict__ dqkv,
                                                            int T, int D, int H, int dh, double scale, DropD dr) {
    // q, k, v, do: [T][dh | 1] each; P, dS: [T][T | 1].  Odd row strides (round 4): thread i walks row i of q / do / P / dS, and with
    // the even strides dh and T all threads of a wave sat on two LDS banks.
    extern __shared__ double sm[];
    const int ldq = dh | 1, ldp = T | 1;
    double* qs = sm;
    double* ks = qs + T * ldq;
    double* vs = ks + T * ldq;
    double* dos = vs + T * ldq;
    double* P = dos + T * ldq;
    double* dS = P + T * ldp;
    const int hd = blockIdx.x % H;
    const long long b = blockIdx.x / H;
    const double* base = qkv + b * T * 3 * D;
    for (int i = threadIdx.x; i < T * dh; i += blockDim.x) {
        const int r = i / dh, e = i - r * dh;
        qs[r * ldq + e] = base[(long long)r * 3 * D + hd * dh + e];
        ks[r * ldq + e] = base[(long long)r * 3 * D + D + hd * dh + e];
        vs[r * ldq + e] = base[(long long)r * 3 * D + 2 * D + hd * dh + e];
        dos[r * ldq + e] = datt[(b * T + r) * D + hd * dh + e];
    }
    __syncthreads();
    // Element-wise passes over the lower triangle with the whole workgroup (dot products, exp, division, dropout hash), one thread per
    // row only for the order-sensitive row maximum / row sum / sum_k dP_ik P_ik.  P is left as P * keep (dV's operand).
    double* rmx = dS + T * ldp;    // [T] row maximum, then row sum, then the row's dot
    for (int idx = threadIdx.x; idx < T * T; idx += blockDim.x) {
        const int i = idx / T, j = idx - i * T;
        if (j > i) { P[i * ldp + j] = 0.0; dS[i * ldp + j] = 0.0; continue; }
        double sc = 0.0, dpd = 0.0;
        for (int e = 0; e < dh; ++e) {
            sc += (qs[i * ldq + e] * scale) * ks[j * ldq + e];
            dpd += dos[i * ldq + e] * vs[j * ldq + e];
        }
        P[i * ldp + j] = sc;
        dS[i * ldp + j] = dpd * drop_factor(dr, (unsigned long long)((b * H + hd) * T + i) * T + j);   // dP
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        double mx = -INFINITY;
        for (int j = 0; j <= i; ++j) { const double sc = P[i * ldp + j]; mx = sc > mx ? sc : mx; }
        rmx[i] = mx;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < T * T; idx += blockDim.x) {
        const int i = idx / T, j = idx - i * T;
        if (j <= i) P[i * ldp + j] = exp(P[i * ldp + j] - rmx[i]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        double den = 0.0;
        for (int j = 0; j <= i; ++j) den += P[i * ldp + j];
        rmx[i] = den;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < T * T; idx += blockDim.x) {
        const int i = idx / T, j = idx - i * T;
        if (j <= i) P[i * ldp + j] = P[i * ldp + j] / rmx[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < T; i += blockDim.x) {
        double dot = 0.0;
        for (int j = 0; j <= i; ++j) dot += dS[i * ldp + j] * P[i * ldp + j];
        rmx[i] = dot;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < T * T; idx += blockDim.x) {
        const int i = idx / T, j = idx - i * T;
        if (j > i) continue;
        const double pij = P[i * ldp + j];
        dS[i * ldp + j] = pij * (dS[i * ldp + j] - rmx[i]);
        P[i * ldp + j] = pij * drop_factor(dr, (unsigned long long)((b * H + hd) * T + i) * T + j);
    }
    __syncthreads();
    double* out = dqkv + b * T * 3 * D;
    for (int i = threadIdx.x; i < T * dh; i += blockDim.x) {
        const int r = i / dh, e = i - r * dh;
        double dq = 0.0, dk = 0.0, dv = 0.0;
        for (int j = 0; j <= r; ++j) dq += dS[r * ldp + j] * ks[j * ldq + e];
        for (int ii = r; ii < T; ++ii) {
            dk += dS[ii * ldp + r] * qs[ii * ldq + e];
            dv += P[ii * ldp + r] * dos[ii * ldq + e];
        }
        out[(long long)r * 3 * D + hd * dh + e] = dq * scale;
        out[(long long)r * 3 * D + D + hd * dh + e] = dk * scale;
        out[(long long)r * 3 * D + 2 * D + hd * dh + e] = dv;
    }
}
static size_t attention_bwd_smem(int T, int dh) { return (size_t)(4 * T * (dh | 1) + 2 * T * (T | 1) + T) * sizeof(double); }

// LayerNorm forward into a separate output (the pre-norm sum is kept for the backward)
__global__ __launch_bounds__(64) void ln_fwd_kernel(const double* __restrict__ Z, const double* __restrict__ g, const double* __restrict__ be,
                                                   double* __restrict__ X, int D) {
    const double* z = Z + (long long)blockIdx.x * D;
    double* x = X + (long long)blockIdx.x * D;
    const int lane = threadIdx.x;
    double s = 0.0;
    for (int c = lane; c < D; c += 64) s += z[c];
    const double mean = wave_sum(s) / D;
    double v = 0.0;
    for (int c = lane; c < D; c += 64) { const double d = z[c] - mean; v += d * d; }
    const double rstd = 1.0 / sqrt(wave_sum(v) / D + 1e-5);
    for (int c = lane; c < D; c += 64) x[c] = (z[c] - mean) * rstd * g[c] + be[c];
}
// LayerNorm backward: dz = rstd (dxh - mean(dxh) - xh mean(dxh xh)), dxh = dy g; dgx = dy xh (column sums of dgx / dy = dgamma / dbeta)
__global__ __launch_bounds__(64) void ln_bwd_kernel(const double* __restrict__ Z, const double* __restrict__ g, const double* __restrict__ dY,
                                                   double* __restrict__ dZ, double* __restrict__ dGX, int D) {
    const long long r = blockIdx.x;
    const double* z = Z + r * D;
    const double* dy = dY + r * D;
    const int lane = threadIdx.x;
    double s = 0.0;
    for (int c = lane; c < D; c += 64) s += z[c];
    const double mean = wave_sum(s) / D;
    double v = 0.0;
    for (int c = lane; c < D; c += 64) { const double d = z[c] - mean; v += d * d; }
    const double rstd = 1.0 / sqrt(wave_sum(v) / D + 1e-5);
    double a = 0.0, bsum = 0.0;
    for (int c = lane; c < D; c += 64) {
        const double xh = (z[c] - mean) * rstd, dxh = dy[c] * g[c];
        a += dxh;
        bsum += dxh * xh;
    }
    a = wave_sum(a) / D;
    bsum = wave_sum(bsum) / D;
    for (int c = lane; c < D; c += 64) {
        const double xh = (z[c] - mean) * rstd, dxh = dy[c] * g[c];
        dZ[r * D + c] = rstd * (dxh - a - xh * bsum);
        dGX[r * D + c] = dy[c] * xh;
    }
}

// out[c] (+= when acc) = sum over rows of x[r][c]: kColRows rows per block in a fixed order, then a fixed-order pass over the blocks
// (32 rows, round 4: with 256 a [10240 x 256] sum was 40 workgroups of 256 dependent loads each — 65 us, 35 times per step)
constexpr int kColRows = 32;
__global__ __launch_bounds__(256) void colsum_part_kernel(const double* __restrict__ x, int ld, long long M, int N, double* __restrict__ part) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    const long long r0 = (long long)blockIdx.y * kColRows, r1 = r0 + kColRows < M ? r0 + kColRows : M;
    double s = 0.0;
    for (long long r = r0; r < r1; ++r) s += x[r * ld + c];
    part[(long long)blockIdx.y * N + c] = s;
}
// 16 columns per block; thread (pl, c) sums parts pl, pl + 16, ... in order, then the 16 lanes of a column meet in LDS in a fixed order
__global__ __launch_bounds__(256) void colsum_final_kernel(const double* __restrict__ part, int nparts, int N, double* __restrict__ out, double* __restrict__ out2) {
    __shared__ double red[16][17];
    const int cl = threadIdx.x & 15, pl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    double s = 0.0;
    if (c < N)
        for (int p_ = pl; p_ < nparts; p_ += 16) s += part[(long long)p_ * N + c];
    red[pl][cl] = s;
    __syncthreads();
    if (pl == 0 && c < N) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) t += red[q][cl];
        out[c] = t;
        if (out2) out2[c] = t;
    }
}

// dW[N][K] partials: part[z][n][k] = sum over the rows of slice z of dY[m][n] X[m][k]  (64 x 64 tiles, fp64 matrix cores)
__global__ __launch_bounds__(256) void gemm_tn_kernel(const double* __restrict__ dY, int ldy, const double* __restrict__ X, int ldx, long long M,
                                                      int N, int K, long long rows_per_split, double* __restrict__ part) {
    __shared__ double Ys[KC * (TM + 1)], Xs[KC * (TN + 1)];   // [m][n], [m][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int wr = wave >> 1, wc = wave & 1;
    const int n0 = blockIdx.y * TM, k0 = blockIdx.x * TN;
    const long long m0 = (long long)blockIdx.z * rows_per_split, m1 = m0 + rows_per_split < M ? m0 + rows_per_split : M;
    d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = d4{0.0, 0.0, 0.0, 0.0};
    const int lm = tid >> 4, lc = (tid & 15) * 4;   // this thread stages 4 consecutive columns of row lm of both operands
    double yv[4], xv[4];
    auto fetch = [&](long long mm) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long long m = mm + lm;
            yv[q] = (m < m1 && n0 + lc + q < N) ? dY[m * ldy + n0 + lc + q] : 0.0;
            xv[q] = (m < m1 && k0 + lc + q < K) ? X[m * ldx + k0 + lc + q] : 0.0;
        }
    };
    fetch(m0);
    for (long long mm = m0; mm < m1; mm += KC) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            Ys[lm * (TM + 1) + lc + q] = yv[q];
            Xs[lm * (TN + 1) + lc + q] = xv[q];
        }
        __syncthreads();
        if (mm + KC < m1) fetch(mm + KC);   // the next rows travel during this chunk's MFMAs (round 4)
#pragma unroll
        for (int s = 0; s < KC / 4; ++s) {
            double a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = Ys[(4 * s + lg) * (TM + 1) + wr * 32 + i * 16 + l15];
                b[i] = Xs[(4 * s + lg) * (TN + 1) + wc * 32 + i * 16 + l15];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    double* P = part + (long long)blockIdx.z * N * K;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = k0 + wc * 32 + j * 16 + l15;
            if (k >= K) continue;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = n0 + wr * 32 + i * 16 + e * 4 + lg;
                if (n < N) P[(long long)n * K + k] = acc[i][j][e];
            }
        }
}
// out[map(n)][k] = sum_z part[z][n][k]; shuffle (H > 0): row n = a*H + b of the product is row b*dh + a of the tensor (:88-89 undone);
// columns [z0, z1) are written as 0 (the root-velocity columns :75 never reach the model)
__global__ __launch_bounds__(256) void splitk_sum_kernel(const double* __restrict__ part, int nsplit, int N, int K, double* __restrict__ out,
                                                         int H, int dh, int z0, int z1) {
    const long long total = (long long)N * K;
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int n = (int)(i / K), k = (int)(i - (long long)n * K);
        double s = 0.0;
        for (int z = 0; z < nsplit; ++z) s += part[(long long)z * total + i];
        const int on = H ? (n % H) * dh + n / H : n;
        out[(long long)on * K + k] = (k >= z0 && k < z1) ? 0.0 : s;
    }
}
__global__ void unshuffle_vec_kernel(const double* __restrict__ in, double* __restrict__ out, int D, int H, int dh) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < D) out[(n % H) * dh + n / H] = in[n];
}

// general transpose: out[c][r] = in[r][c]
__global__ void transpose2_kernel(const double* __restrict__ in, double* __restrict__ out, int rows, int cols) {
    const long long n = (long long)rows * cols;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i / rows), r = (int)(i - (long long)c * rows);
        out[i] = in[(long long)r * cols + c];
    }
}

// dpre = dhid * [hid > 0] * scale2   (hid is stored after ReLU and dropout: > 0 exactly where the unit's gate is open)
__global__ void relu_gate_kernel(const double* __restrict__ dhid, const double* __restrict__ hid, double* __restrict__ dpre, long long n, double scale) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        dpre[i] = hid[i] > 0.0 ? dhid[i] * scale : 0.0;
}
__global__ void add_kernel(const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ y, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = a[i] + b[i];
}

struct TrainLay {
    // saved (doubles)
    size_t U, X0, IH, HALL, WT, total_saved;
    size_t qkv[16], att[16], z1[16], x1[16], hid[16], z2[16], xo[16];
    // scratch (doubles)
    size_t ga, gb, gc, gbig, gbig2, wt, part, colpart, dwin, dbin, total_scratch;
    int nsplit;
    long long rows_per_split;
};
constexpr int kMaxLayersF64 = 16;

static TrainLay train_layout(const Dims& d, int B, int T) {
    TrainLay L{};
    const size_t M = (size_t)B * T;
    auto al = [](size_t v) { return (v + 31) & ~(size_t)31; };
    size_t o = 0;
    L.U = o; o += al(M * d.In);
    L.X0 = o; o += al(M * d.D);
    for (int l = 0; l < d.L; ++l) {
        L.qkv[l] = o; o += al(M * 3 * d.D);
        L.att[l] = o; o += al(M * d.D);
        L.z1[l] = o; o += al(M * d.D);
        L.x1[l] = o; o += al(M * d.D);
        L.hid[l] = o; o += al(M * d.F);
        L.z2[l] = o; o += al(M * d.D);
        L.xo[l] = o; o += al(M * d.D);
    }
    L.IH = o; o += al(d.with_rnn ? M * d.R : 0);
    L.HALL = o; o += al(d.with_rnn ? M * d.R : 0);
    L.WT = o; o += al(d.with_rnn ? (size_t)d.R * d.R : 0);   // W_hh^T: the backward recurrence's operand
    L.total_saved = o;
    o = 0;
    size_t wide = 3 * (size_t)d.D;
    if ((size_t)d.F > wide) wide = d.F;
    if ((size_t)d.R > wide) wide = d.R;
    if ((size_t)round_up(d.S, 4) > wide) wide = round_up(d.S, 4);
    L.ga = o; o += al(M * d.D);
    L.gb = o; o += al(M * d.D);
    L.gc = o; o += al(M * d.D);
    L.gbig = o; o += al(M * wide);
    L.gbig2 = o; o += al(M * wide);
    size_t wmax = (size_t)d.F * d.D;
    for (size_t v : {(size_t)3 * d.D * d.D, (size_t)d.R * d.R, (size_t)d.R * d.D, (size_t)d.S * (d.with_rnn ? d.R : d.D), (size_t)d.D * d.In})
        if (v > wmax) wmax = v;
    L.wt = o; o += al(wmax);
    L.rows_per_split = 512;
    L.nsplit = (int)((M + L.rows_per_split - 1) / L.rows_per_split);
    if (L.nsplit > 64) { L.nsplit = 64; L.rows_per_split = (long long)((M + 63) / 64); L.rows_per_split = (L.rows_per_split + KC - 1) / KC * KC; L.nsplit = (int)((M + L.rows_per_split - 1) / L.rows_per_split); }
    L.part = o; o += al(wmax * L.nsplit);
    L.colpart = o; o += al(((M + kColRows - 1) / kColRows) * wide);
    L.dwin = o; o += al((size_t)d.D * d.In);
    L.dbin = o; o += al(d.D);
    L.total_scratch = o;
    return L;
}

static int grid_n(long long n) { const long long g = (n + 255) / 256; return (int)(g < 65536 ? (g > 0 ? g : 1) : 65536); }

}  // namespace f64
}  // namespace tip

extern "C" {

int tip_train_bytes_f64(const tip_handle* h, int B, int T, size_t* saved_bytes, size_t* scratch_bytes) {
    if (!h || B < 0 || T < 0) return TIP_ERR_INVALID_ARG;
    const Dims& d = h->d;
    if (d.L > f64::kMaxLayersF64 || T > 128) return TIP_ERR_UNSUPPORTED_CONFIG;   // attention backward keeps two T x T tiles in LDS
    const long long M = (long long)B * T;
    if (M * (long long)(3 * d.D > d.F ? 3 * d.D : d.F) > 0x7fffffffLL || M * d.H > 0x7fffffffLL || M > 65535LL * f64::kColRows) return TIP_ERR_UNSUPPORTED_CONFIG;
    if (f64::attention_bwd_smem(T, d.dh) > 160 * 1024) return TIP_ERR_UNSUPPORTED_CONFIG;
    const f64::TrainLay L = f64::train_layout(d, B > 0 ? B : 1, T > 0 ? T : 1);
    if (saved_bytes) *saved_bytes = L.total_saved * sizeof(double) + 256;
    if (scratch_bytes) *scratch_bytes = L.total_scratch * sizeof(double) + 256;
    return TIP_OK;
}

int tip_train_forward_f64(tip_handle* h, const double* const* params, int n_params, const double* x_imu, const double* x_s,
                          const double* keep_mask, double keep_scale, float p_drop, unsigned long long seed, double* y, void* saved,
                          size_t saved_bytes, int B, int T, void* stream) {
    if (!h || !params || !x_imu || !x_s || !y || !saved || B < 1 || T < 1) return TIP_ERR_INVALID_ARG;
    if (n_params != (int)h->tensor_names.size()) return TIP_ERR_INVALID_ARG;
    if (p_drop < 0.f || p_drop >= 1.f) return TIP_ERR_INVALID_ARG;
    size_t need = 0;
    const int st = tip_train_bytes_f64(h, B, T, &need, nullptr);
    if (st != TIP_OK) return st;
    if (reinterpret_cast<uintptr_t>(saved) % 256 || saved_bytes < need - 256) return TIP_ERR_WORKSPACE;
    const Dims& d = h->d;
    const f64::TrainLay L = f64::train_layout(d, B, T);
    const long long M = (long long)B * T;
    hipStream_t s = static_cast<hipStream_t>(stream);
    double* W = static_cast<double*>(saved);
    auto fail = [&](hipError_t e, const char* what) {
        h->last_hip_error = std::string(what) + ": " + hipGetErrorString(e);
        return (int)TIP_ERR_HIP;
    };
    CoopSerial serial(h->device, s);
    if (serial.status != hipSuccess) return fail(serial.status, "stream serialisation");
#define TF(expr, what) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(e_, what); } while (0)
    const double* const* tw = params + 2 + 12 * d.L;
    hipLaunchKernelGGL(f64::prologue_kernel, dim3(f64::grid_n(M * d.In)), dim3(256), 0, s, x_imu, x_s, keep_mask, keep_mask ? keep_scale : 1.0, W + L.U,
                       M, d.n_imu_total, d.S, d.rootv0, d.rootv1);
    TF(hipGetLastError(), "f64 train prologue");
    TF(f64::gemm(4, W + L.U, d.In, params[0], params[1], nullptr, 0, W + L.X0, d.D, M, d.D, d.In, d.H, d.dh, s), "f64 train in_linear");
    const double scale = 1.0 / sqrt((double)d.dh);
    const double* x = W + L.X0;
    for (int l = 0; l < d.L; ++l) {
        const double* const* lp = params + 2 + 12 * l;
        TF(f64::gemm(0, x, d.D, lp[0], lp[1], nullptr, 0, W + L.qkv[l], 3 * d.D, M, 3 * d.D, d.D, 0, 1, s), "f64 train in_proj");
        if (f64::attention_wg_ok(T, d.dh))
            hipLaunchKernelGGL(f64::attention_wg_kernel<true>, dim3((unsigned)(B * d.H)), dim3(256), f64::attention_wg_smem(T, d.dh), s, W + L.qkv[l],
                               W + L.att[l], T, d.D, d.H, d.dh, scale, f64::make_drop_d(p_drop, seed, 4 * l + 0));
        else
            hipLaunchKernelGGL(f64::attention_train_kernel, dim3((unsigned)(M * d.H)), dim3(64), (size_t)T * sizeof(double), s, W + L.qkv[l],
                               W + L.att[l], T, d.D, d.H, d.dh, scale, f64::make_drop_d(p_drop, seed, 4 * l + 0));
        TF(hipGetLastError(), "f64 train attention");
        // z1 = x + drop1(att Wo^T + bo): the product lands in z1, then the dropout / residual pass rewrites it in place
        TF(f64::gemm(0, W + L.att[l], d.D, lp[2], lp[3], nullptr, 0, W + L.z1[l], d.D, M, d.D, d.D, 0, 1, s), "f64 train out_proj");
        hipLaunchKernelGGL(f64::drop_res_kernel, dim3(f64::grid_n(M * d.D)), dim3(256), 0, s, W + L.z1[l], x, W + L.z1[l], M * d.D,
                           f64::make_drop_d(p_drop, seed, 4 * l + 1));
        hipLaunchKernelGGL(f64::ln_fwd_kernel, dim3((unsigned)M), dim3(64), 0, s, W + L.z1[l], lp[8], lp[9], W + L.x1[l], d.D);
        TF(hipGetLastError(), "f64 train norm1");
        TF(f64::gemm(1, W + L.x1[l], d.D, lp[4], lp[5], nullptr, 0, W + L.hid[l], d.F, M, d.F, d.D, 0, 1, s), "f64 train linear1");
        hipLaunchKernelGGL(f64::drop_res_kernel, dim3(f64::grid_n(M * d.F)), dim3(256), 0, s, W + L.hid[l], (const double*)nullptr, W + L.hid[l],
                           M * d.F, f64::make_drop_d(p_drop, seed, 4 * l + 2));
        TF(f64::gemm(0, W + L.hid[l], d.F, lp[6], lp[7], nullptr, 0, W + L.z2[l], d.D, M, d.D, d.F, 0, 1, s), "f64 train linear2");
        hipLaunchKernelGGL(f64::drop_res_kernel, dim3(f64::grid_n(M * d.D)), dim3(256), 0, s, W + L.z2[l], W + L.x1[l], W + L.z2[l], M * d.D,
                           f64::make_drop_d(p_drop, seed, 4 * l + 3));
        hipLaunchKernelGGL(f64::ln_fwd_kernel, dim3((unsigned)M), dim3(64), 0, s, W + L.z2[l], lp[10], lp[11], W + L.xo[l], d.D);
        TF(hipGetLastError(), "f64 train norm2");
        x = W + L.xo[l];
    }
    const double* feat = x;
    int fw = d.D;
    if (d.with_rnn) {
        TF(f64::gemm(0, x, d.D, tw[0], tw[2], nullptr, 0, W + L.IH, d.R, M, d.R, d.D, 0, 1, s), "f64 train rnn W_ih");
        hipLaunchKernelGGL(f64::transpose_kernel, dim3(f64::grid_n((long long)d.R * d.R)), dim3(256), 0, s, tw[1], W + L.WT, d.R);
        if (d.R % 64) return TIP_ERR_UNSUPPORTED_CONFIG;
        TF(f64::rnn_steps_fwd(W + L.IH, tw[1], tw[3], W + L.HALL, B, T, d.R, s), "f64 train rnn steps");   // (W_hh^T above is the backward's operand)
        feat = W + L.HALL;
        fw = d.R;
        tw += 4;
    }
    TF(f64::gemm(0, feat, fw, tw[0], tw[1], nullptr, 0, y, d.S, M, d.S, fw, 0, 1, s), "f64 train linear");
#undef TF
    ++h->forward_count;
    return TIP_OK;
}

int tip_train_backward_f64(tip_handle* h, const double* const* params, int n_params, const double* dy, const void* saved, size_t saved_bytes,
                           void* scratch, size_t scratch_bytes, double* grads, size_t grads_doubles, float p_drop, unsigned long long seed,
                           int B, int T, void* stream) {
    if (!h || !params || !dy || !saved || !scratch || !grads || B < 1 || T < 1) return TIP_ERR_INVALID_ARG;
    if (n_params != (int)h->tensor_names.size()) return TIP_ERR_INVALID_ARG;
    size_t need_s = 0, need_x = 0;
    const int st = tip_train_bytes_f64(h, B, T, &need_s, &need_x);
    if (st != TIP_OK) return st;
    if (reinterpret_cast<uintptr_t>(saved) % 256 || saved_bytes < need_s - 256) return TIP_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(scratch) % 256 || scratch_bytes < need_x - 256) return TIP_ERR_WORKSPACE;
    const Dims& d = h->d;
    const f64::TrainLay L = f64::train_layout(d, B, T);
    const long long M = (long long)B * T;
    std::vector<size_t> goff(n_params);
    size_t gtot = 0;
    for (int i = 0; i < n_params; ++i) {
        goff[i] = gtot;
        const auto& sh = h->tensor_shapes[i];
        gtot += (size_t)sh.first * (sh.second ? sh.second : 1);
    }
    if (grads_doubles < gtot) return TIP_ERR_INVALID_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const double* W = static_cast<const double*>(saved);
    double* X = static_cast<double*>(scratch);
    auto fail = [&](hipError_t e, const char* what) {
        h->last_hip_error = std::string(what) + ": " + hipGetErrorString(e);
        return (int)TIP_ERR_HIP;
    };
    CoopSerial serial(h->device, s);
    if (serial.status != hipSuccess) return fail(serial.status, "stream serialisation");
#define TF(expr, what) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return fail(e_, what); } while (0)
    // out[N][K] = dY^T X (fixed-order split over the rows); shuffle / zeroed columns for in_linear only
    auto grad_w = [&](const double* dYp, int ldy, const double* Xp, int ldx, int N, int K, double* out, int H = 0, int dh = 0, int z0 = 0, int z1 = 0) -> hipError_t {
        const dim3 grid((K + f64::TN - 1) / f64::TN, (N + f64::TM - 1) / f64::TM, L.nsplit);
        hipLaunchKernelGGL(f64::gemm_tn_kernel, grid, dim3(256), 0, s, dYp, ldy, Xp, ldx, M, N, K, L.rows_per_split, X + L.part);
        hipLaunchKernelGGL(f64::splitk_sum_kernel, dim3(f64::grid_n((long long)N * K)), dim3(256), 0, s, X + L.part, L.nsplit, N, K, out, H, dh, z0, z1);
        return hipGetLastError();
    };
    auto col_sum = [&](const double* x, int ld, int N, double* out, double* out2 = nullptr) -> hipError_t {
        const int nparts = (int)((M + f64::kColRows - 1) / f64::kColRows);
        hipLaunchKernelGGL(f64::colsum_part_kernel, dim3((N + 255) / 256, nparts), dim3(256), 0, s, x, ld, M, N, X + L.colpart);
        hipLaunchKernelGGL(f64::colsum_final_kernel, dim3((N + 15) / 16), dim3(256), 0, s, X + L.colpart, nparts, N, out, out2);
        return hipGetLastError();
    };
    // dX[M][K] = dY[M][N] W[N][K]: the NT GEMM on a transposed copy of the weight
    auto grad_x = [&](const double* dYp, int ldy, const double* Wp, int N, int K, double* out, const double* res) -> hipError_t {
        hipLaunchKernelGGL(f64::transpose2_kernel, dim3(f64::grid_n((long long)N * K)), dim3(256), 0, s, Wp, X + L.wt, N, K);   // wt [K][N]
        return f64::gemm(res ? 2 : 0, dYp, ldy, X + L.wt, nullptr, res, K, out, K, M, K, N, 0, 1, s);
    };
    const int rbase = 2 + 12 * d.L;
    const int g_lin_w = rbase + (d.with_rnn ? 4 : 0), g_lin_b = g_lin_w + 1;
    const int Kout = d.with_rnn ? d.R : d.D;
    const double* enc = d.L ? W + L.xo[d.L - 1] : W + L.X0;
    const double* head_in = d.with_rnn ? W + L.HALL : enc;
    double* gx = X + L.ga;      // gradient w.r.t. the current layer's output
    double* galt = X + L.gb;
    // ---- output projection (:102) ------------------------------------------------------------------------------------
    TF(col_sum(dy, d.S, d.S, grads + goff[g_lin_b]), "f64 bwd db_out");
    TF(grad_w(dy, d.S, head_in, Kout, d.S, Kout, grads + goff[g_lin_w]), "f64 bwd dW_out");
    if (d.with_rnn) {
        double* dH = X + L.gbig;
        double* delta = X + L.gbig2;
        TF(grad_x(dy, d.S, params[g_lin_w], d.S, d.R, dH, nullptr), "f64 bwd dH");
        // ---- recurrence (:98-99), time reversed; hprev lands in dH's place afterwards (dH is dead once delta exists) — not in
        //      place: the kernel reads dH[t] and writes hprev[t] at the same index in the same thread, after the read
        TF(f64::rnn_steps_bwd(dH, W + L.WT, W + L.HALL, delta, B, T, d.R, s), "f64 bwd rnn steps");
        TF(col_sum(delta, d.R, d.R, grads + goff[rbase + 2], grads + goff[rbase + 3]), "f64 bwd db_rnn");
        TF(grad_w(delta, d.R, dH /* = hprev */, d.R, d.R, d.R, grads + goff[rbase + 1]), "f64 bwd dW_hh");
        TF(grad_w(delta, d.R, enc, d.D, d.R, d.D, grads + goff[rbase + 0]), "f64 bwd dW_ih");
        TF(grad_x(delta, d.R, params[rbase + 0], d.R, d.D, gx, nullptr), "f64 bwd d_enc");
    } else {
        TF(grad_x(dy, d.S, params[g_lin_w], d.S, d.D, gx, nullptr), "f64 bwd d_enc");
    }
    // ---- encoder layers, last to first -----------------------------------------------------------------------------------
    const double qs = 1.0 / sqrt((double)d.dh);
    const size_t att_smem = f64::attention_bwd_smem(T, d.dh);
    if (att_smem > 64 * 1024)   // (per call: cheap, and correct for every device the process drives)
        TF(hipFuncSetAttribute(reinterpret_cast<const void*>(f64::attention_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)att_smem),
           "f64 attention_bwd LDS");
    for (int l = d.L - 1; l >= 0; --l) {
        const int pb = 2 + 12 * l;
        const double* const* lp = params + pb;
        const double* x_in = l ? W + L.xo[l - 1] : W + L.X0;
        double* dz = galt;            // LayerNorm input gradient (= the residual path's gradient)
        double* dgx = X + L.gc;       // dy * xhat (dgamma's summand), then reused
        double* big = X + L.gbig;
        double* big2 = X + L.gbig2;
        // LayerNorm2
        hipLaunchKernelGGL(f64::ln_bwd_kernel, dim3((unsigned)M), dim3(64), 0, s, W + L.z2[l], lp[10], gx, dz, dgx, d.D);
        TF(col_sum(dgx, d.D, d.D, grads + goff[pb + 10]), "f64 bwd dgamma2");
        TF(col_sum(gx, d.D, d.D, grads + goff[pb + 11]), "f64 bwd dbeta2");
        // linear2: its output gradient is dz under dropout site 3
        double* dff2 = X + L.gc;
        hipLaunchKernelGGL(f64::drop_res_kernel, dim3(f64::grid_n(M * d.D)), dim3(256), 0, s, dz, (const double*)nullptr, dff2, M * d.D,
                           f64::make_drop_d(p_drop, seed, 4 * l + 3));
        TF(col_sum(dff2, d.D, d.D, grads + goff[pb + 7]), "f64 bwd db2");
        TF(grad_w(dff2, d.D, W + L.hid[l], d.F, d.D, d.F, grads + goff[pb + 6]), "f64 bwd dW2");
        TF(grad_x(dff2, d.D, lp[6], d.D, d.F, big, nullptr), "f64 bwd dhid");
        hipLaunchKernelGGL(f64::relu_gate_kernel, dim3(f64::grid_n(M * d.F)), dim3(256), 0, s, big, W + L.hid[l], big, M * d.F,
                           f64::make_drop_d(p_drop, seed, 4 * l + 2).scale);
        TF(col_sum(big, d.F, d.F, grads + goff[pb + 5]), "f64 bwd db1");
        TF(grad_w(big, d.F, W + L.x1[l], d.D, d.F, d.D, grads + goff[pb + 4]), "f64 bwd dW1");
        TF(grad_x(big, d.F, lp[4], d.F, d.D, gx, dz), "f64 bwd dx1");          // gx = dz2 + dpre W1
        // LayerNorm1
        hipLaunchKernelGGL(f64::ln_bwd_kernel, dim3((unsigned)M), dim3(64), 0, s, W + L.z1[l], lp[8], gx, dz, dgx, d.D);
        TF(col_sum(dgx, d.D, d.D, grads + goff[pb + 8]), "f64 bwd dgamma1");
        TF(col_sum(gx, d.D, d.D, grads + goff[pb + 9]), "f64 bwd dbeta1");
        double* dout = X + L.gc;
        hipLaunchKernelGGL(f64::drop_res_kernel, dim3(f64::grid_n(M * d.D)), dim3(256), 0, s, dz, (const double*)nullptr, dout, M * d.D,
                           f64::make_drop_d(p_drop, seed, 4 * l + 1));
        TF(col_sum(dout, d.D, d.D, grads + goff[pb + 3]), "f64 bwd dbo");
        TF(grad_w(dout, d.D, W + L.att[l], d.D, d.D, d.D, grads + goff[pb + 2]), "f64 bwd dWo");
        double* datt = gx;            // gx is free until dx_in is formed
        TF(grad_x(dout, d.D, lp[2], d.D, d.D, datt, nullptr), "f64 bwd datt");
        hipLaunchKernelGGL(f64::attention_bwd_kernel, dim3((unsigned)(B * d.H)), dim3(256), att_smem, s,
                           W + L.qkv[l], datt, big2, T, d.D, d.H, d.dh, qs, f64::make_drop_d(p_drop, seed, 4 * l + 0));
        TF(hipGetLastError(), "f64 bwd attention");
        TF(col_sum(big2, 3 * d.D, 3 * d.D, grads + goff[pb + 1]), "f64 bwd dbqkv");
        TF(grad_w(big2, 3 * d.D, x_in, d.D, 3 * d.D, d.D, grads + goff[pb + 0]), "f64 bwd dWqkv");
        TF(grad_x(big2, 3 * d.D, lp[0], 3 * d.D, d.D, gx, dz), "f64 bwd dx_in");   // gx = dz1 + dqkv Wqkv
    }
    // ---- in_linear (:79) with the channel shuffle (:88-89) undone and the root-velocity columns (:75) at zero ----------------
    TF(col_sum(gx, d.D, d.D, X + L.dbin), "f64 bwd db_in");
    hipLaunchKernelGGL(f64::unshuffle_vec_kernel, dim3((d.D + 255) / 256), dim3(256), 0, s, X + L.dbin, grads + goff[1], d.D, d.H, d.dh);
    TF(grad_w(gx, d.D, W + L.U, d.In, d.D, d.In, grads + goff[0], d.H, d.dh, d.n_imu_total + d.rootv0, d.n_imu_total + d.rootv1), "f64 bwd dW_in");
#undef TF
    return TIP_OK;
}

}  // extern "C"

// tip_f64.hip — the forward pass in fp64 (the reference's `--double` switch, /root/reference/train_model.py:62-63,84-85,161-164:
// torch.set_default_dtype(torch.float64) builds the module with fp64 parameters and feeds it fp64 windows).
//
// Same function as tip_forward (/root/reference/simple_transformer_with_state.py:60-102), every operation in IEEE double:
// the parameters are read RAW (fp64 device tensors in state-dict order, as tip_train_forward reads its fp32 ones — nothing is
// packed), the big products run on the fp64 matrix cores (v_mfma_f64_16x16x4_f64, 64x64 tiles staged through LDS), softmax /
// LayerNorm / tanh use the double-precision device math library.  `--double` is a debugging switch of the training script, so
// this path is built for exactness and any configuration, not for the roofline: layer-by-layer kernels, one workgroup per window
// for the recurrence (W_hh^T streamed from L2 every step).  It is also the on-device high-precision check of the fp32 plans at sizes the CPU
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
    for (int k0 = 0; k0 < K; k0 += KC) {
        double av[4], wv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = k0 + lk + q;
            av[q] = (row0 + lr < M && k < K) ? A[(row0 + lr) * lda + k] : 0.0;
            wv[q] = (col0 + lr < N && k < K) ? W[(long long)(col0 + lr) * K + k] : 0.0;
        }
        __syncthreads();   // the previous chunk's fragments have been read
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            As[lr * LDS_LD + lk + q] = av[q];
            Ws[lr * LDS_LD + lk + q] = wv[q];
        }
        __syncthreads();
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
// One workgroup per window; thread c owns output channels c, c + blockDim.x, ...; W_hh^T rows are read coalesced.
__global__ void rnn_kernel(const double* __restrict__ IH, const double* __restrict__ WhhT, const double* __restrict__ bhh,
                           double* __restrict__ HALL, int T, int R) {
    extern __shared__ double hbuf[];   // [2][R]
    const long long b = blockIdx.x;
    for (int c = threadIdx.x; c < R; c += blockDim.x) hbuf[c] = 0.0;
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        const double* h = hbuf + (t & 1) * R;
        double* hn = hbuf + ((t + 1) & 1) * R;
        for (int c = threadIdx.x; c < R; c += blockDim.x) {
            double a = bhh[c];
            for (int k = 0; k < R; ++k) a += h[k] * WhhT[(long long)k * R + c];
            const double v = tanh(IH[(b * T + t) * R + c] + a);
            hn[c] = v;
            HALL[(b * T + t) * R + c] = v;
        }
        __syncthreads();
    }
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
    L.WT = o;   o += al(d.with_rnn ? (size_t)d.R * d.R : 0);
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
        const long long n = (long long)d.R * d.R;
        hipLaunchKernelGGL(f64::transpose_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tw[1], W + L.WT, d.R);
        TF(hipGetLastError(), "f64 W_hh transpose");
        const int threads = d.R < 1024 ? ((d.R + 63) / 64) * 64 : 1024;
        hipLaunchKernelGGL(f64::rnn_kernel, dim3(B), dim3(threads), 2 * (size_t)d.R * sizeof(double), s, W + L.IH, W + L.WT, tw[3],
                           W + L.HALL, T, d.R);
        TF(hipGetLastError(), "f64 rnn");
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

// tip_general.hip — "general" execution plan: the forward pass as layer-by-layer gfx950 kernels.
// Works for any configuration (paper, scaled, tiny); the fused plan (tip_fused.hip) specialises the
// paper configuration.  All arithmetic is fp32; GEMMs run on v_mfma_f32_32x32x2_f32 /
// v_mfma_f32_16x16x4_f32 (exact fp32, fmaf-chain numerics).
//
// Stage map (reference: /root/reference/simple_transformer_with_state.py)
//   prologue_kernel     :63-78   clone, NaN scrub, past-state keep-mask, concat (root-vel zero :75 is folded
//                                into the packed in_linear columns; the scrub must stay in-kernel: 0*NaN = NaN)
//   gemm_kernel         :79 in_linear (+ channel shuffle :88-89 folded into its rows), QKV / out_proj /
//                                linear1 / linear2 of every encoder layer (:91), RNN input projection (:99),
//                                output projection (:102)
//   attention_kernel    :85,:91  causal 16-head scaled-dot-product attention (torch functional.py SDPA)
//   layernorm_kernel    :91      post-norm LayerNorm (eps 1e-5, biased variance)
//   rnn_kernel          :98-99   h_t = tanh(ih_t + W_hh h_{t-1}), h_0 = 0
#include <stdlib.h>

#include <atomic>
#include <type_traits>
#include <random>

#include "tip_internal.h"
#include "tip_head.h"
#include "tip_layernorm.h"

namespace tip {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// every bounded hand-off spin that gives up bumps this counter (read with tip_spin_timeouts()): a non-zero value
// means a cluster member never arrived and the outputs of that launch are invalid.
__device__ unsigned g_spin_timeouts_general;
__device__ __forceinline__ void note_spin_timeout(unsigned* err) {
    atomicAdd(&g_spin_timeouts_general, 1u);
    guard_report(err);
}

// ------------------------------------------------------------------------------------------------
// prologue: U[row][0:InPad] = [x_imu | scrub(x_s) * mask * scale | 0-pad]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prologue_kernel(const float* __restrict__ x_imu,
                                                       const float* __restrict__ x_s,
                                                       const float* __restrict__ keep_mask, float keep_scale,
                                                       float* __restrict__ U, int M, int NI, int S, int InPad) {
    const size_t total = (size_t)M * InPad;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int row = (int)(i / InPad);
        const int c = (int)(i - (size_t)row * InPad);
        float v = 0.f;
        if (c < NI) {
            v = x_imu[(size_t)row * NI + c];
        } else if (c < NI + S) {
            const size_t j = (size_t)row * S + (c - NI);
            v = x_s[j];
            if (v != v) v = 0.f;  // :65 x_s[x_s.isnan()] = 0
            if (keep_mask) v = v * keep_mask[j] * keep_scale;  // :77 with an explicit Bernoulli keep-mask
        }
        U[i] = v;
    }
}

hipError_t launch_prologue(const Dims& d, const float* x_imu, const float* x_s, const float* keep_mask,
                           float keep_scale, float* U, int M, hipStream_t s) {
    const size_t total = (size_t)M * d.InPad;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(prologue_kernel, dim3(blocks), dim3(256), 0, s, x_imu, x_s, keep_mask, keep_scale, U, M,
                       d.n_imu_total, d.S, d.InPad);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// GEMM: C[M,N] = epi(A[M,K] * W[Npad,K]^T + bias (+res)).  128x128x16 block tile, 4 waves as 2x2, each wave
// 64x64 = 2x2 tiles of v_mfma_f32_32x32x2_f32.  A and W tiles are staged through LDS k-major
// ([k][row], row stride 130 floats) so both the transposing ds_write_b32 and the fragment ds_read_b32 are
// bank-conflict free; global->register prefetch of tile k+1 overlaps the MFMAs of tile k.
// ------------------------------------------------------------------------------------------------
constexpr int LDT = kGemmBM + 2;

template <int FLAGS>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W,
                                                   int K, const float* __restrict__ bias,
                                                   const float* __restrict__ res, int ldres, float* __restrict__ C,
                                                   int ldc, int M, int N) {
    __shared__ float As[2][kGemmBK][LDT];
    __shared__ float Bs[2][kGemmBK][LDT];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int bm = blockIdx.y, bn = blockIdx.x;
    const int lr = tid >> 2;         // 0..63
    const int lk = (tid & 3) * 4;    // 0,4,8,12

    const int arow0 = bm * kGemmBM + lr, arow1 = arow0 + 64;
    const bool av0 = arow0 < M, av1 = arow1 < M;
    const float* ap0 = A + (size_t)(av0 ? arow0 : 0) * lda + lk;
    const float* ap1 = A + (size_t)(av1 ? arow1 : 0) * lda + lk;
    const float* wp0 = W + (size_t)(bn * kGemmBN + lr) * K + lk;   // packed W is padded: always in range
    const float* wp1 = wp0 + (size_t)64 * K;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ra0, ra1, rb0, rb1;
    ra0 = av0 ? *reinterpret_cast<const float4*>(ap0) : zero4;
    ra1 = av1 ? *reinterpret_cast<const float4*>(ap1) : zero4;
    rb0 = *reinterpret_cast<const float4*>(wp0);
    rb1 = *reinterpret_cast<const float4*>(wp1);

    auto stage = [&](int buf) {
        As[buf][lk + 0][lr] = ra0.x; As[buf][lk + 1][lr] = ra0.y; As[buf][lk + 2][lr] = ra0.z; As[buf][lk + 3][lr] = ra0.w;
        As[buf][lk + 0][lr + 64] = ra1.x; As[buf][lk + 1][lr + 64] = ra1.y; As[buf][lk + 2][lr + 64] = ra1.z; As[buf][lk + 3][lr + 64] = ra1.w;
        Bs[buf][lk + 0][lr] = rb0.x; Bs[buf][lk + 1][lr] = rb0.y; Bs[buf][lk + 2][lr] = rb0.z; Bs[buf][lk + 3][lr] = rb0.w;
        Bs[buf][lk + 0][lr + 64] = rb1.x; Bs[buf][lk + 1][lr + 64] = rb1.y; Bs[buf][lk + 2][lr + 64] = rb1.z; Bs[buf][lk + 3][lr + 64] = rb1.w;
    };
    stage(0);
    __syncthreads();

    const int nk = K / kGemmBK;
    const int l31 = lane & 31, lhi = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            const int ko = (kt + 1) * kGemmBK;
            ra0 = av0 ? *reinterpret_cast<const float4*>(ap0 + ko) : zero4;
            ra1 = av1 ? *reinterpret_cast<const float4*>(ap1 + ko) : zero4;
            rb0 = *reinterpret_cast<const float4*>(wp0 + ko);
            rb1 = *reinterpret_cast<const float4*>(wp1 + ko);
        }
#pragma unroll
        for (int kk = 0; kk < kGemmBK / 2; ++kk) {
            const int k = kk * 2 + lhi;
            float a[2], b[2];
            a[0] = As[cur][k][wm * 64 + l31];
            a[1] = As[cur][k][wm * 64 + 32 + l31];
            b[0] = Bs[cur][k][wn * 64 + l31];
            b[1] = Bs[cur][k][wn * 64 + 32 + l31];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) stage(cur ^ 1);
        __syncthreads();
    }

    // epilogue.  C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = bn * kGemmBN + wn * 64 + j * 32 + l31;
            if (col >= N) continue;
            const float bv = bias[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = bm * kGemmBM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (row < M) {
                    float v = acc[i][j][r] + bv;
                    if (FLAGS & 2) v += res[(size_t)row * ldres + col];
                    if (FLAGS & 1) v = v > 0.f ? v : 0.f;
                    C[(size_t)row * ldc + col] = v;
                }
            }
        }
    }
}

hipError_t launch_gemm(const float* A, int lda, const float* W, int Kpad, const float* bias, const float* res,
                       int ldres, float* C, int ldc, int M, int N, int Npad, int flags, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    // default: the 16x16x4 / 64x64-tile kernel of tip_train.hip (measured: paper B=256 general plan 1.27 vs 1.59 ms, scaled
    // B=128 32.3 vs 35.5 ms, scaled B=512 equal); TIP_GENERAL_GEMM=32 selects the 128x128 32x32x2 kernel below
    static int use16 = -1;
    if (use16 < 0) use16 = (tip_env("TIP_GENERAL_GEMM") && atoi(tip_env("TIP_GENERAL_GEMM")) == 32) ? 0 : 1;
    if (use16) return launch_gemm16(A, lda, W, Kpad, bias, res, ldres, C, ldc, M, N, Kpad, flags, s);
    dim3 grid(Npad / kGemmBN, (M + kGemmBM - 1) / kGemmBM);
    dim3 block(256);
    switch (flags & 3) {
        case 0: hipLaunchKernelGGL(gemm_kernel<0>, grid, block, 0, s, A, lda, W, Kpad, bias, res, ldres, C, ldc, M, N); break;
        case 1: hipLaunchKernelGGL(gemm_kernel<1>, grid, block, 0, s, A, lda, W, Kpad, bias, res, ldres, C, ldc, M, N); break;
        case 2: hipLaunchKernelGGL(gemm_kernel<2>, grid, block, 0, s, A, lda, W, Kpad, bias, res, ldres, C, ldc, M, N); break;
        default: hipLaunchKernelGGL(gemm_kernel<3>, grid, block, 0, s, A, lda, W, Kpad, bias, res, ldres, C, ldc, M, N); break;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Skinny "head" GEMM (:102 output projection, N = 131): Y[M,N] = A[M,K] * W^T + b.  N is far below the 128-wide
// block tile, so every wave owns 16 rows x ALL column blocks (v_mfma_f32_16x16x4_f32): its A fragments come
// straight from global memory (each row is read exactly once, 64 B per lane group) and the weight fragments
// (fragment-ordered, L2-resident, <= 288 KB) through a buffer descriptor, register double-buffered.
// ------------------------------------------------------------------------------------------------
template <int NBO>   // 16-column blocks per workgroup (9 = all of N = 131; 1 when M is small and grid.y walks the blocks)
__global__ __launch_bounds__(256) void head_gemm_kernel(const float* __restrict__ A, long long lda,
                                                        const float* __restrict__ wfrag, const float* __restrict__ bias,
                                                        float* __restrict__ Y, int ldy, int M, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int row0 = (blockIdx.x * 4 + wave) * 16;
    if (row0 >= M) return;
    const int KB = K / 16;
    const int nbase = blockIdx.y * NBO;   // first column block of this workgroup
    const __amdgpu_buffer_rsrc_t wrs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wfrag), 0, ((N + 15) / 16) * K * 64, 0x00020000);
    auto ldw = [&](int n, int kb) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, ((nbase + n) * KB + kb) * 1024, 0));
    };
    const int arow = row0 + l15 < M ? row0 + l15 : M - 1;   // clamp: padded rows are computed but never stored
    const float* ap = A + (size_t)arow * lda + lg * 4;
    f32x4 acc[NBO];
#pragma unroll
    for (int n = 0; n < NBO; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (KB % 8 == 0) {
        // Steady state without a branch: a ring of 4 k-blocks of weight fragments (L2) and 8 of A rows (HBM: HALL has just
        // been written by the RNN); stage j is consumed, THEN refilled (index clamped at the end: a harmless reload), and a
        // scheduling barrier pins each refill where it stands.  The first version guarded every stage with `if (kb < KB)` and
        // rotated wc <- wn by copies: the compiler answered with `s_waitcnt vmcnt(0)` right after each load, i.e. no
        // prefetch at all — a full memory round trip per k-block (23 us for a 10-us product).
        f32x4 wr[4][NBO];
        float4 ar[8];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < NBO; ++n) wr[j][n] = ldw(n, j);
#pragma unroll
        for (int j = 0; j < 8; ++j) ar[j] = *reinterpret_cast<const float4*>(ap + j * 16);
        auto body = [&](int kb0, int KBc) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float4 a = ar[j];
#pragma unroll
                for (int n = 0; n < NBO; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wr[j & 3][n].x, acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NBO; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wr[j & 3][n].y, acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NBO; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wr[j & 3][n].z, acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NBO; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wr[j & 3][n].w, acc[n], 0, 0, 0);
                const int kw = kb0 + j + 4 < KBc ? kb0 + j + 4 : KBc - 1;
                const int ka = kb0 + j + 8 < KBc ? kb0 + j + 8 : KBc - 1;
#pragma unroll
                for (int n = 0; n < NBO; ++n) wr[j & 3][n] = ldw(n, kw);
                ar[j] = *reinterpret_cast<const float4*>(ap + ka * 16);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#pragma unroll 1
        for (int kb0 = 0; kb0 < KB; kb0 += 8) body(kb0, KB);   // (fully unrolling K = 512 measured no better)
    } else {
    f32x4 wc[NBO], wn[NBO];
#pragma unroll
    for (int n = 0; n < NBO; ++n) wc[n] = ldw(n, 0);
    // A fragments come from HBM (HALL was just written by the RNN): keep 4 k-blocks of them in flight
    constexpr int AD = 4;
    float4 aq[AD];
#pragma unroll
    for (int j = 0; j < AD; ++j) aq[j] = *reinterpret_cast<const float4*>(ap + (j < KB ? j : KB - 1) * 16);
#pragma unroll 1
    for (int kb0 = 0; kb0 < KB; kb0 += AD) {
#pragma unroll
        for (int j = 0; j < AD; ++j) {
            const int kb = kb0 + j;
            if (kb < KB) {
                const int kn = kb + 1 < KB ? kb + 1 : kb;
#pragma unroll
                for (int n = 0; n < NBO; ++n) wn[n] = ldw(n, kn);
                const float4 a = aq[j];
                const int ka = kb + AD < KB ? kb + AD : KB - 1;
                aq[j] = *reinterpret_cast<const float4*>(ap + ka * 16);
#pragma unroll
                for (int n = 0; n < NBO; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, wc[n].x, acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NBO; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, wc[n].y, acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NBO; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, wc[n].z, acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NBO; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, wc[n].w, acc[n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < NBO; ++n) wc[n] = wn[n];
            }
        }
    }
    }
#pragma unroll
    for (int n = 0; n < NBO; ++n) {
        const int col = (nbase + n) * 16 + l15;
        if (col < N) {
            const float bv = bias[col];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = row0 + lg * 4 + e;
                if (row < M) Y[(size_t)row * ldy + col] = acc[n][e] + bv;
            }
        }
    }
}

hipError_t launch_head_gemm(const float* A, long long lda, const float* wfrag, const float* bias, float* Y, int ldy,
                            int M, int N, int K, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    const int nbo = (N + 15) / 16;
    const dim3 block(256);
    if ((M + 63) / 64 * 4 < 256) {
        // few rows: one column block per workgroup so the (rows x column blocks) grid still covers many CUs
        hipLaunchKernelGGL(head_gemm_kernel<1>, dim3((M + 63) / 64, nbo), block, 0, s, A, lda, wfrag, bias, Y, ldy, M, N, K);
        return hipGetLastError();
    }
    if (nbo % 3 == 0) {
        // 3 column blocks per wave: three times the waves (every SIMD gets work, 2+ waves each hide the fragment latency)
        hipLaunchKernelGGL(head_gemm_kernel<3>, dim3((M + 63) / 64, nbo / 3), block, 0, s, A, lda, wfrag, bias, Y, ldy, M, N, K);
        return hipGetLastError();
    }
    const dim3 grid((M + 63) / 64);
#define TIP_HEAD_CASE(NB) \
    case NB: hipLaunchKernelGGL(head_gemm_kernel<NB>, grid, block, 0, s, A, lda, wfrag, bias, Y, ldy, M, N, K); break;
    switch (nbo) {
        TIP_HEAD_CASE(1) TIP_HEAD_CASE(2) TIP_HEAD_CASE(3) TIP_HEAD_CASE(4) TIP_HEAD_CASE(5) TIP_HEAD_CASE(6)
        TIP_HEAD_CASE(7) TIP_HEAD_CASE(8) TIP_HEAD_CASE(9) TIP_HEAD_CASE(10) TIP_HEAD_CASE(11) TIP_HEAD_CASE(12)
        default: return hipErrorInvalidValue;
    }
#undef TIP_HEAD_CASE
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// attention for ANY window length (the matrix-core kernels of tip_attn.hip serve T <= 128; the reference builds its mask for
// any T, simple_transformer_with_state.py:56-58,85): one wave per (window, head), lane i owns query row r0 + i of a 64-query
// chunk — q and the output accumulator in registers, online softmax, so the row reductions need no cross-lane traffic — and
// the keys/values stream through a wave-private LDS tile of TK keys at a time (key-tiled: LDS use does not grow with T).
// A wave's LDS operations complete in program order: refilling its own tile needs no workgroup barrier.
// ------------------------------------------------------------------------------------------------
template <int DH>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int B,
                                                        int T, int H, float q_scale, int TK) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int waves = blockDim.x >> 6;
    const int D = H * DH;
    const int ld = 3 * D;
    const int bh = blockIdx.x * waves + wave;
    float* Ks = smem + (size_t)wave * 2 * TK * DH;
    float* Vs = Ks + (size_t)TK * DH;
    if (bh >= B * H) return;
    const int b = bh / H, h = bh % H;
    const float* base = qkv + (size_t)b * T * ld + h * DH;
    constexpr int V4 = DH / 4;
    for (int r0 = 0; r0 < T; r0 += 64) {
        const int i = r0 + lane;
        const bool rv = i < T;
        float q[DH], o[DH];
        const float* qp = base + (size_t)(rv ? i : 0) * ld;
#pragma unroll
        for (int e = 0; e < DH; e += 4) {
            const float4 t = *reinterpret_cast<const float4*>(qp + e);
            q[e] = t.x * q_scale; q[e + 1] = t.y * q_scale; q[e + 2] = t.z * q_scale; q[e + 3] = t.w * q_scale;
        }
#pragma unroll
        for (int e = 0; e < DH; ++e) o[e] = 0.f;
        float m = -INFINITY, l = 0.f;
        const int jmax = min(T, r0 + 64);  // keys needed by the last row of this chunk
        for (int k0 = 0; k0 < jmax; k0 += TK) {
            const int kn = min(TK, jmax - k0);
            __builtin_amdgcn_wave_barrier();   // (the previous tile's reads are issued before it is overwritten)
            for (int f = lane; f < kn * V4; f += 64) {
                const int j = f / V4, e = (f % V4) * 4;
                *reinterpret_cast<float4*>(Ks + j * DH + e) = *reinterpret_cast<const float4*>(base + (size_t)(k0 + j) * ld + D + e);
                *reinterpret_cast<float4*>(Vs + j * DH + e) = *reinterpret_cast<const float4*>(base + (size_t)(k0 + j) * ld + 2 * D + e);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int jj = 0; jj < kn; ++jj) {
                const int j = k0 + jj;
                const float* kj = Ks + jj * DH;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
                for (int e = 0; e < DH; e += 4) {
                    const float4 kv = *reinterpret_cast<const float4*>(kj + e);
                    s0 = fmaf(q[e], kv.x, s0); s1 = fmaf(q[e + 1], kv.y, s1);
                    s2 = fmaf(q[e + 2], kv.z, s2); s3 = fmaf(q[e + 3], kv.w, s3);
                }
                const float sc = (s0 + s1) + (s2 + s3);
                if (j <= i) {  // causal mask (:56-58): key j visible to query i iff j <= i
                    const float mn = fmaxf(m, sc);
                    const float corr = expf(m - mn);   // exp(-inf) = 0 on the first key
                    const float p = expf(sc - mn);
                    l = l * corr + p;
                    const float* vj = Vs + jj * DH;
#pragma unroll
                    for (int e = 0; e < DH; e += 4) {
                        const float4 vv = *reinterpret_cast<const float4*>(vj + e);
                        o[e] = fmaf(o[e], corr, p * vv.x); o[e + 1] = fmaf(o[e + 1], corr, p * vv.y);
                        o[e + 2] = fmaf(o[e + 2], corr, p * vv.z); o[e + 3] = fmaf(o[e + 3], corr, p * vv.w);
                    }
                    m = mn;
                }
            }
        }
        if (rv) {
            const float inv = 1.f / l;
            float* op = out + (size_t)(b * T + i) * D + h * DH;
#pragma unroll
            for (int e = 0; e < DH; e += 4)
                *reinterpret_cast<float4*>(op + e) = make_float4(o[e] * inv, o[e + 1] * inv, o[e + 2] * inv, o[e + 3] * inv);
        }
    }
}

hipError_t launch_attention(const Dims& d, const float* qkv, float* out, int B, int T, hipStream_t s) {
    static int valu = -1;   // TIP_GENERAL_ATTN=valu keeps the one-thread-per-query kernel below (measurement / fallback)
    if (valu < 0) valu = (tip_env("TIP_GENERAL_ATTN") && tip_env("TIP_GENERAL_ATTN")[0] == 'v') ? 1 : 0;
    if (!valu && mattn_supported(d.dh, T)) {
        Drop off;
        off.seed = 0; off.site = 0; off.thresh = 0; off.scale = 1.f;
        return launch_mattn_fwd(qkv, out, nullptr, B, T, d.H, d.dh, d.q_scale, off, s);
    }
    // key tile per wave: the whole window when it fits 16 KB per wave (4 waves = 64 KB, no opt-in needed), else 16 KB worth of keys
    const int waves = 4;
    int TK = (16 * 1024) / (2 * d.dh * (int)sizeof(float));
    if (TK > T) TK = T;
    const int nbh = B * d.H;
    dim3 grid((nbh + waves - 1) / waves), block(64 * waves);
    const size_t smem = (size_t)waves * 2 * TK * d.dh * sizeof(float);
    switch (d.dh) {
        case 8: hipLaunchKernelGGL(attention_kernel<8>, grid, block, smem, s, qkv, out, B, T, d.H, d.q_scale, TK); break;
        case 16: hipLaunchKernelGGL(attention_kernel<16>, grid, block, smem, s, qkv, out, B, T, d.H, d.q_scale, TK); break;
        case 32: hipLaunchKernelGGL(attention_kernel<32>, grid, block, smem, s, qkv, out, B, T, d.H, d.q_scale, TK); break;
        case 64: hipLaunchKernelGGL(attention_kernel<64>, grid, block, smem, s, qkv, out, B, T, d.H, d.q_scale, TK); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// LayerNorm (in place): one wave per row, two-pass in registers, wave-shuffle reductions.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) { return wave64_sum(v); }   // DPP + permlane swaps (tip_layernorm.h)

template <int NV>  // float4 per lane: D <= NV*256
__global__ __launch_bounds__(256) void layernorm_kernel(float* __restrict__ x, const float* __restrict__ g,
                                                        const float* __restrict__ be, int M, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= M) return;
    float* xr = x + (size_t)row * D;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        v[i] = c < D ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, dd = v[i].w - mean;
            q += (a * a + b * b) + (cc * cc + dd * dd);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        if (c < D) {
            const float4 gg = *reinterpret_cast<const float4*>(g + c);
            const float4 bb = *reinterpret_cast<const float4*>(be + c);
            float4 o;
            o.x = (v[i].x - mean) * rstd * gg.x + bb.x;
            o.y = (v[i].y - mean) * rstd * gg.y + bb.y;
            o.z = (v[i].z - mean) * rstd * gg.z + bb.z;
            o.w = (v[i].w - mean) * rstd * gg.w + bb.w;
            *reinterpret_cast<float4*>(xr + c) = o;
        }
    }
}

hipError_t launch_layernorm(float* x, const float* g, const float* b, int M, int D, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    dim3 grid((M + 3) / 4), block(256);
    if (D <= 256) hipLaunchKernelGGL(layernorm_kernel<1>, grid, block, 0, s, x, g, b, M, D);
    else if (D <= 512) hipLaunchKernelGGL(layernorm_kernel<2>, grid, block, 0, s, x, g, b, M, D);
    else if (D <= 1024) hipLaunchKernelGGL(layernorm_kernel<4>, grid, block, 0, s, x, g, b, M, D);
    else if (D <= 2048) hipLaunchKernelGGL(layernorm_kernel<8>, grid, block, 0, s, x, g, b, M, D);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// RNN recurrence.  A tile of 16 windows is the M dimension of v_mfma_f32_16x16x4_f32; the 40 (T) steps are
// serial.  `cluster` workgroups share one tile, each owning R/cluster output columns; after every step the
// column slices are exchanged through HALL itself (the [B,T,R] output the out-linear GEMM reads next) with
// an agent-scope release / counter / acquire hand-off (cdna_hip_programming.md section 6 Guideline 16).
// W_hh is pre-packed in B-fragment order: one coalesced 1-KiB dwordx4 load per 16x16 block per wave.
//   whh_frag[((nb*KB + kb)*64 + lane)*4 + s] = W_hh[nb*16 + (lane&15)][kb*16 + 4*(lane>>4) + s]
// and the A fragment of MFMA step s is h[lane&15][kb*16 + 4*(lane>>4) + s]  (same k permutation on both
// operands, so each group of 4 MFMAs covers 16 consecutive k).
// ------------------------------------------------------------------------------------------------
// BWD = true: the backward recurrence of the training step for any rnn_hidden (tip_train.hip; the register-resident kernels below
// serve 512 only): delta_t = (dH_t + delta_{t+1} W_hh) * (1 - h_t^2), t = T-1 .. 0, with `ih` = dH, `whh_frag` = fragments of
// W_hh^T, `gate` = the forward states h, `hall` = delta (output).  Same tiling, hand-off and accumulation order, time reversed.
template <int NBW, bool BWD = false>  // NBW: 16-column blocks per wave: R / (16 * 4 * cluster)
__global__ __launch_bounds__(256) void rnn_kernel(const float* __restrict__ ih, const float* __restrict__ whh_frag,
                                                  float* __restrict__ hall, unsigned* __restrict__ flags, int B, int T,
                                                  int R, int cluster, int ntiles, Guard gd, const float* __restrict__ gate) {
    extern __shared__ __attribute__((aligned(16))) float smem[];  // h tile [16][R+4]
    __shared__ int s_poison;   // a hand-off wait of this workgroup gave up: everything it produces from here on is NaN
    if (threadIdx.x == 0) s_poison = 0;
    if ((gd.fault & 2) && cluster > 1 && blockIdx.x == 1) return;   // TIP_OPT_FAULT_INJECT: this member never arrives
    const int LDH = R + 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int KB = R / 16;                 // k blocks
    const int cid = blockIdx.x % cluster;  // member index inside the cluster
    const int group = blockIdx.x / cluster;
    const int ngroups = gridDim.x / cluster;
    const int l15 = lane & 15, lg = lane >> 4;
    const int nb0 = (cid * 4 + wave) * NBW;  // first global column block of this wave
    // W_hh fragments through a buffer descriptor: opaque to the optimiser, so the one-block-ahead prefetch survives
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(whh_frag), 0, R * R * 4, 0x00020000);
    auto ldw = [&](int n, int kb) -> f32x4 {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, lane * 16, ((nb0 + n) * KB + kb) * 1024, 0));
    };

    for (int tile = group; tile < ntiles; tile += ngroups) {
        const int b0 = tile * kRnnTile;
        for (int i = tid; i < kRnnTile * LDH; i += 256) smem[i] = 0.f;  // h_{-1} = 0
        __syncthreads();
        for (int st = 0; st < T; ++st) {
            const int t = BWD ? T - 1 - st : st;        // time index of this step
            const int tp = BWD ? t + 1 : t - 1;         // ... and of the step before it (whose result is the A operand)
            // input projection of this step (bias b_ih + b_hh already folded in): issued early, used after the MFMAs
            float ihv[NBW][4], gv[NBW][4];
#pragma unroll
            for (int n = 0; n < NBW; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int bb = b0 + lg * 4 + r;
                    ihv[n][r] = bb < B ? ih[((size_t)bb * T + t) * R + (nb0 + n) * 16 + l15] : 0.f;
                    gv[n][r] = (BWD && bb < B) ? gate[((size_t)bb * T + t) * R + (nb0 + n) * 16 + l15] : 0.f;
                }
            if (st > 0 && cluster > 1) {
                // wait for every member's slice of h_{t-1}, then pull the full [16][R] tile from HALL
                if (tid == 0 && !s_poison) {
                    unsigned* f = flags + (size_t)tile * T + tp;
                    const unsigned lim = guard_spin_limit(gd.fault, 1u << 22);
                    unsigned spins = 0;
                    for (; spins < lim; ++spins) {  // bounded: never hang the GPU
                        if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)cluster) break;
                        __builtin_amdgcn_s_sleep(1);
                    }
                    if (spins == lim) {
                        note_spin_timeout(gd.err);
                        s_poison = st;                     // (st >= 1 here: the step whose wait gave up)
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                if (s_poison == st) {
                    // h_{t-1} is incomplete (the missing member's slice of HALL holds stale memory): this member's slice of that
                    // row turns NaN too, so the output row t-1 cannot come out finite
#pragma unroll
                    for (int n = 0; n < NBW; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int bb = b0 + lg * 4 + r;
                            if (bb < B) hall[((size_t)bb * T + tp) * R + (nb0 + n) * 16 + l15] = __uint_as_float(kPoisonBits);
                        }
                }
                for (int i = tid; i < kRnnTile * (R / 4); i += 256) {
                    const int m = i / (R / 4), c = (i % (R / 4)) * 4;
                    const int bb = b0 + m;
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (bb < B) v = *reinterpret_cast<const float4*>(hall + ((size_t)bb * T + tp) * R + c);
                    *reinterpret_cast<float4*>(smem + m * LDH + c) = v;
                }
                __syncthreads();
            }
            // Canonical accumulation order (shared by every RNN kernel variant so that a stream's result does not
            // depend on how many streams share the launch): four fmaf chains per output — k-blocks of the lower /
            // upper half of K, even / odd — combined as (c00 + c01) + (c10 + c11).
            f32x4 acc[NBW];
            if (st > 0) {
                f32x4 ch[4][NBW];
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int n = 0; n < NBW; ++n) ch[c][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                f32x4 wc[NBW], wn[NBW];
#pragma unroll
                for (int n = 0; n < NBW; ++n) wc[n] = ldw(n, 0);
#pragma unroll
                for (int half = 0; half < 2; ++half) {
#pragma unroll 1
                    for (int kb = half * (KB / 2); kb < (half + 1) * (KB / 2); kb += 2) {
#pragma unroll
                        for (int n = 0; n < NBW; ++n) wn[n] = ldw(n, kb + 1);
                        const float4 a0 = *reinterpret_cast<const float4*>(smem + l15 * LDH + kb * 16 + lg * 4);
#pragma unroll
                        for (int n = 0; n < NBW; ++n) ch[half * 2][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, wc[n].x, ch[half * 2][n], 0, 0, 0);
#pragma unroll
                        for (int n = 0; n < NBW; ++n) ch[half * 2][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, wc[n].y, ch[half * 2][n], 0, 0, 0);
#pragma unroll
                        for (int n = 0; n < NBW; ++n) ch[half * 2][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, wc[n].z, ch[half * 2][n], 0, 0, 0);
#pragma unroll
                        for (int n = 0; n < NBW; ++n) ch[half * 2][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, wc[n].w, ch[half * 2][n], 0, 0, 0);
                        const int k2 = kb + 2 < KB ? kb + 2 : kb;
#pragma unroll
                        for (int n = 0; n < NBW; ++n) wc[n] = ldw(n, k2);
                        const float4 a1 = *reinterpret_cast<const float4*>(smem + l15 * LDH + (kb + 1) * 16 + lg * 4);
#pragma unroll
                        for (int n = 0; n < NBW; ++n) ch[half * 2 + 1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, wn[n].x, ch[half * 2 + 1][n], 0, 0, 0);
#pragma unroll
                        for (int n = 0; n < NBW; ++n) ch[half * 2 + 1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, wn[n].y, ch[half * 2 + 1][n], 0, 0, 0);
#pragma unroll
                        for (int n = 0; n < NBW; ++n) ch[half * 2 + 1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, wn[n].z, ch[half * 2 + 1][n], 0, 0, 0);
#pragma unroll
                        for (int n = 0; n < NBW; ++n) ch[half * 2 + 1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, wn[n].w, ch[half * 2 + 1][n], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int n = 0; n < NBW; ++n) acc[n] = (ch[0][n] + ch[1][n]) + (ch[2][n] + ch[3][n]);
            } else {
#pragma unroll
                for (int n = 0; n < NBW; ++n) acc[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            __syncthreads();  // everyone is done reading h_{t-1} from LDS
            // epilogue: D layout of 16x16: col = lane&15, row = 4*(lane>>4) + r.  Values first, stores after (vmcnt counts
            // stores too: interleaving them with uses of the prefetched ihv serialises on every store).
            float hvv[NBW][4];
#pragma unroll
            for (int n = 0; n < NBW; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    hvv[n][r] = (b0 + lg * 4 + r < B) ? (s_poison ? __uint_as_float(kPoisonBits)
                                                             : (BWD ? (acc[n][r] + ihv[n][r]) * (1.0f - gv[n][r] * gv[n][r]) : tip_tanh(acc[n][r] + ihv[n][r])))
                                                  : 0.f;
#pragma unroll
            for (int n = 0; n < NBW; ++n) {
                const int col = (nb0 + n) * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = lg * 4 + r;
                    const int bb = b0 + m;
                    if (bb < B) hall[((size_t)bb * T + t) * R + col] = hvv[n][r];
                    smem[m * LDH + col] = hvv[n][r];
                }
            }
            if (cluster > 1) {
                // publish this slice of h_t: drain stores, workgroup barrier, one agent-scope release, counter++
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __hip_atomic_fetch_add(flags + (size_t)tile * T + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            } else {
                __syncthreads();
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// RNN recurrence, register-resident variant (R = 512).  The serial 40-step chain is latency-bound, so the
// 16-window tile is spread over CLUSTER = 512 / (16*WAVES/KSPLIT) workgroups; every wave keeps its
// [16 cols x 512/KSPLIT] slice of W_hh in VGPRs for the whole launch (64 or 128 registers per lane) — nothing is
// re-read per step except the 32-KB hidden tile.  Per step: sc1 (write-through) stores of the new h slice into
// HALL -> per-(tile,step) arrival counter -> the other members poll it and pull the tile back with sc1 loads
// (agent-scope coherent across the 8 XCD L2s; no fences on the critical path).
// ------------------------------------------------------------------------------------------------
// HANDOFF = 0: arrival counter (stores -> drain -> barrier -> counter; consumers poll the counter, then pull).
// HANDOFF = 1: the data is the flag — HALL is pre-filled with an all-ones sentinel (a NaN bit pattern no arithmetic
//              produces), producers just store, consumers re-pull until no sentinel word is left: two memory round
//              trips (drain + counter) leave the per-step critical path.
constexpr unsigned kRnnSentinel = 0xFFFFFFFFu;

// measurement only (TIP_RNN_TRACE=1): per-step s_memtime stamps of workgroup 0 — after the pull, after the MFMAs,
// after the reduce+tanh+stores — read back with tip_debug_read_rnn_trace().
__device__ unsigned long long g_rnn_trace[2048];   // [0, 256): step stamps of workgroup 0; [1024, 1536): HW_ID | XCC_ID << 32 per workgroup (tools/rnn_hwid.py)

// BWD = true runs the backward recurrence of the training step on the same machinery (tip_train.hip):
//   delta_t = (dH_t + delta_{t+1} W_hh) * (1 - h_t^2),  t = T-1 .. 0
// with `ih` = dH, `whh_frag` = fragments of W_hh^T, `gate` = the forward states h, `hall` = delta (output).
template <int WAVES, int KSPLIT, int HANDOFF, bool TRACE = false, bool BWD = false>
__global__ __launch_bounds__(WAVES * 64) void rnn_resident_kernel(const float* __restrict__ ih,
                                                                   const float* __restrict__ whh_frag,
                                                                   float* __restrict__ hall, unsigned* __restrict__ flags,
                                                                   int B, int T, int ntiles, int hall_bytes,
                                                                   const float* __restrict__ gate, Guard gd, int prepoll) {
    constexpr int R = 512, KB = R / 16, KBW = KB / KSPLIT;      // k-blocks per wave
    constexpr int CBW = WAVES / KSPLIT;                          // 16-column blocks per workgroup
    constexpr int CLUSTER = KB / CBW;
    constexpr int LDH = R + 4;
    constexpr int THREADS = WAVES * 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];  // h tile [16][LDH] | k-split partials
    float* red = smem + kRnnTile * LDH;                           // [KSPLIT-1][CBW][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave % CBW, ks = wave / CBW;
    // Cluster membership.  Workgroups are observed to land on XCD (blockIdx % 8); when the grid allows it the members of
    // a cluster are chosen 8 blocks apart so that they share one XCD and its L2 (a SPEED choice only, see below).
    const int ngroups = gridDim.x / CLUSTER;
    int cid, group;
    if (gridDim.x % (8 * CLUSTER) == 0) {
        const int x = blockIdx.x & 7, j = blockIdx.x >> 3;
        cid = j % CLUSTER;
        group = x + 8 * (j / CLUSTER);
    } else {
        cid = blockIdx.x % CLUSTER;
        group = blockIdx.x / CLUSTER;
    }
    const int l15 = lane & 15, lg = lane >> 4;
    const int nb = cid * CBW + cb;                                // global 16-column block of this wave
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(hall, 0, hall_bytes, 0x00020000);
    if ((gd.fault & 2) && group == 0 && cid == 1) return;         // TIP_OPT_FAULT_INJECT: this member never arrives
    // A wait that gives up poisons: the waiter's words become canonical NaNs (never the sentinel: peers must not wait on
    // them), every h it produces from then on is NaN, and it never spins again (one load pass per step).
    bool poisoned = false;
    __shared__ int s_poison0;                                     // HANDOFF == 0: the counter wait is lane 0's
    if (HANDOFF == 0 && tid == 0) s_poison0 = 0;
    const unsigned spin_big = guard_spin_limit(gd.fault, 1u << 22), spin_pull = guard_spin_limit(gd.fault, 1u << 20);
    // Every member pulls the same 32-KB tile at the same moment: started at the same row they would queue behind each other at
    // every L2 line.  Member cid starts its sweep at row cid (TIP_RNN_ROTATE=0 switches it off for measurement).
    const int rot = prepoll & 2 ? cid : 0;

    // Same-XCD fast path, VERIFIED at run time (placement is never assumed): every member publishes the XCC id it
    // really runs on (agent-scope), reads the others', and only if all 16 agree do producers use plain stores — which
    // stay in the XCD's L2, where the members' L1-bypassing (sc1) loads find them after ~0.4 us instead of a
    // write-through + fabric round trip.  Any disagreement (or HANDOFF == 0) keeps the write-through sc1 stores that
    // are correct for every placement.
    __shared__ int s_same_xcd;
    if (HANDOFF == 1) {
        if (tid == 0) {
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            xcc &= 0xf;
            // the exchange words carry this LAUNCH's tag (23 bits, a process-wide counter from a random start) above the XCC id:
            // what an earlier launch left there never matches, so the words need no memset (a 4.7-us fill kernel plus its
            // boundary in front of every recurrence)
            const unsigned etag = (unsigned)prepoll >> 8, mine = (etag << 5) | (xcc + 1u);   // etag < 2^23
            __hip_atomic_store(flags + group * CLUSTER + cid, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            bool same = true;
            for (int m = 0; m < CLUSTER; ++m) {
                unsigned v = 0;
                bool here = false;
                for (unsigned spins = 0; spins < spin_big; ++spins) {
                    v = __hip_atomic_load(flags + group * CLUSTER + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    here = (v >> 5) == etag && (v & 31u) != 0u;
                    if (here) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!here) note_spin_timeout(gd.err);   // (the step loop below will time out on this member too, and poison)
                same &= (v == mine);
            }
            s_same_xcd = (same && !(gd.fault & 8)) ? 1 : 0;
        }
        __syncthreads();
    }
    const bool same_xcd = HANDOFF == 1 && s_same_xcd != 0;

    // W_hh slice -> registers, once.  KSPLIT = 1 / 2: wave ks owns the contiguous k-blocks [ks*KBW, (ks+1)*KBW).
    // KSPLIT = 4: wave ks owns ONE canonical chain — half ks>>1, parity ks&1: k-blocks (ks>>1)*16 + (ks&1) + 2i.
    float4 wreg[KBW];
    {
        const float4* wf = reinterpret_cast<const float4*>(whh_frag) + (size_t)nb * KB * 64 + lane;
#pragma unroll
        for (int k = 0; k < KBW; ++k) {
            const int kblk = KSPLIT == 4 ? (ks >> 1) * (KB / 2) + (ks & 1) + 2 * k : ks * KBW + k;
            wreg[k] = wf[(size_t)kblk * 64];
        }
    }

    for (int tile = group; tile < ntiles; tile += ngroups) {
        const int b0 = tile * kRnnTile;
        for (int t = 0; t < T; ++t) {
            const int te = BWD ? T - 1 - t : t;        // time index this step produces
            const int tp = BWD ? te + 1 : te - 1;      // time index of the state it consumes
            float ihv[4], gv[4];
            if (ks == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int bb = b0 + lg * 4 + r;
                    ihv[r] = bb < B ? ih[((size_t)bb * T + te) * R + nb * 16 + l15] : 0.f;
                    if (BWD) gv[r] = bb < B ? gate[((size_t)bb * T + te) * R + nb * 16 + l15] : 0.f;
                }
            }
            // canonical accumulation (see rnn_kernel): chains {lower,upper half of K} x {even,odd k-block}
            constexpr int NH = KSPLIT == 1 ? 2 : 1;   // K halves handled by this wave
            static_assert(KSPLIT == 1 || KSPLIT == 2 || KSPLIT == 4, "canonical order: two K halves x two k-block parities");
            f32x4 chn[NH][2];
#pragma unroll
            for (int hh = 0; hh < NH; ++hh) chn[hh][0] = chn[hh][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (t > 0) {
                constexpr int NLD = kRnnTile * (R / 4) / THREADS;   // float4 slots of the h tile per thread
                if (HANDOFF == 0) {
                    if (tid == 0 && !s_poison0) {
                        // bounded spin: a cluster member that never arrives must not hang the GPU; giving up poisons
                        const unsigned* f = flags + (size_t)tile * T + (t - 1);
                        unsigned spins = 0;
                        for (; spins < spin_big; ++spins) {
                            if (__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)CLUSTER) break;
                            __builtin_amdgcn_s_sleep(1);
                        }
                        if (spins == spin_big) {
                            note_spin_timeout(gd.err);
                            s_poison0 = t;                 // t >= 1
                        }
                    }
                    __syncthreads();
                    poisoned = s_poison0 != 0;
                    if (s_poison0 == t && ks == 0) {       // the consumed row is incomplete: this member's slice of it turns NaN too
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int bb = b0 + lg * 4 + r;
                            if (bb < B) hall[((size_t)bb * T + tp) * R + nb * 16 + l15] = __uint_as_float(kPoisonBits);
                        }
                    }
                }
                if (HANDOFF == 1 && (prepoll & 1)) {   // (workgroup-uniform condition: there is a barrier inside)
                    // Cheap arrival probe before the 32-KB pull.  Sixteen workgroups re-pulling whole tiles until the last
                    // sentinel is gone keep the XCD's L2 at its bandwidth limit, so every poll round takes ~700 cycles; here ONE
                    // wave watches 1 KB — a 16-byte piece of the LAST row group each producer wave stores (rows 4 lg + 3 of its 16
                    // columns) — and the full pull starts when none of them holds a sentinel.  A hint only: stores of one wave
                    // may become visible out of order, so the pull below still checks every word and re-asks for stragglers.
                    if (wave == 0) {
                        const int nrows = min(kRnnTile, B - b0);
                        const int prow = min(((lane & 1) ? 15 : 7), nrows - 1);          // rows 7 / 15: last store of lg = 1 / 3
                        const int poff = (int)((((size_t)(b0 + prow) * T + tp) * R + (lane >> 1) * 16 + 12) * 4);
                        const unsigned plim = poisoned ? 1u : spin_pull;   // a poisoned lane never waits again
                        for (unsigned spins = 0; spins < plim; ++spins) {
                            const u32x4 pv = __builtin_amdgcn_raw_buffer_load_b128(hrs, poff, 0, 16);
                            const bool pend = pv.x == kRnnSentinel || pv.y == kRnnSentinel || pv.z == kRnnSentinel || pv.w == kRnnSentinel;
                            if (__builtin_amdgcn_ballot_w64(pend) == 0) break;
                        }
                        if (TRACE && blockIdx.x == 0 && tid == 0 && t < 64) g_rnn_trace[t * 4 + 3] = __builtin_amdgcn_s_memtime();
                    }
                    __syncthreads();
                }
                // pull h_{t-1} [16][512] with sc1 loads (aux = 16): bypass this CU's L1, coherent at agent scope
                u32x4 v[NLD];
                bool need[NLD];
#pragma unroll
                for (int j = 0; j < NLD; ++j) {
                    const int i = tid + j * THREADS;
                    need[j] = b0 + ((i / (R / 4) + rot) & (kRnnTile - 1)) < B;
                    v[j] = (u32x4){0u, 0u, 0u, 0u};
                }
                bool gave_up = true;
                const unsigned pull_lim = poisoned ? 1u : spin_pull;
                for (unsigned spins = 0; spins < pull_lim; ++spins) {
                    bool any = false;
#pragma unroll
                    for (int j = 0; j < NLD; ++j) {
                        if (need[j]) {
                            const int i = tid + j * THREADS;
                            const int m = (i / (R / 4) + rot) & (kRnnTile - 1), c = (i % (R / 4)) * 4;
                            v[j] = __builtin_amdgcn_raw_buffer_load_b128(hrs, (int)((((size_t)(b0 + m) * T + tp) * R + c) * 4), 0, 16);
                        }
                    }
                    if (HANDOFF == 0) { gave_up = false; break; }
#pragma unroll
                    for (int j = 0; j < NLD; ++j) {
                        if (need[j]) {
                            const bool pend = v[j].x == kRnnSentinel || v[j].y == kRnnSentinel || v[j].z == kRnnSentinel ||
                                              v[j].w == kRnnSentinel;
                            need[j] = pend;
                            any |= pend;
                        }
                    }
                    if (TRACE && blockIdx.x == 0 && tid == 0 && t < 24 && spins == 0) g_rnn_trace[160 + (t - 1) * 4 + 0] = __builtin_amdgcn_s_memtime();
                    if (TRACE && blockIdx.x == 0 && tid == 0 && t < 24) g_rnn_trace[160 + (t - 1) * 4 + 3] = spins + 1;
                    if (!any) { gave_up = false; break; }
                    if (!same_xcd) __builtin_amdgcn_s_sleep(2);   // cross-XCD polls travel the fabric: pace them
                }
                if (TRACE && blockIdx.x == 0 && tid == 0 && t < 24) g_rnn_trace[160 + (t - 1) * 4 + 1] = __builtin_amdgcn_s_memtime();
                if (gave_up && HANDOFF == 1) {
                    if (!poisoned) note_spin_timeout(gd.err);
                    poisoned = true;
#pragma unroll
                    for (int j = 0; j < NLD; ++j)   // what never arrived becomes NaN, not the sentinel
                        if (need[j]) {
                            if (v[j].x == kRnnSentinel) v[j].x = kPoisonBits;
                            if (v[j].y == kRnnSentinel) v[j].y = kPoisonBits;
                            if (v[j].z == kRnnSentinel) v[j].z = kPoisonBits;
                            if (v[j].w == kRnnSentinel) v[j].w = kPoisonBits;
                        }
                }
#pragma unroll
                for (int j = 0; j < NLD; ++j) {
                    const int i = tid + j * THREADS;
                    const int m = (i / (R / 4) + rot) & (kRnnTile - 1), c = (i % (R / 4)) * 4;
                    if (HANDOFF == 0 && poisoned) v[j] = (u32x4){kPoisonBits, kPoisonBits, kPoisonBits, kPoisonBits};
                    *reinterpret_cast<u32x4*>(smem + m * LDH + c) = v[j];
                }
                if (TRACE && blockIdx.x == 0 && tid == 0 && t < 24) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    g_rnn_trace[160 + (t - 1) * 4 + 2] = __builtin_amdgcn_s_memtime();
                }
                __syncthreads();
                if (TRACE && blockIdx.x == 0 && tid == 0 && t < 64) g_rnn_trace[t * 4 + 0] = __builtin_amdgcn_s_memtime();
                if (KSPLIT == 4) {
                    // one chain per wave: its 8 k-blocks in ascending order, all into chn[0][0]
                    const float* ap4 = smem + l15 * LDH + ((ks >> 1) * (KB / 2) + (ks & 1)) * 16 + lg * 4;
#pragma unroll
                    for (int k = 0; k < KBW; ++k) {
                        const float4 a0 = *reinterpret_cast<const float4*>(ap4 + k * 32);
                        chn[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, wreg[k].x, chn[0][0], 0, 0, 0);
                        chn[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, wreg[k].y, chn[0][0], 0, 0, 0);
                        chn[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, wreg[k].z, chn[0][0], 0, 0, 0);
                        chn[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, wreg[k].w, chn[0][0], 0, 0, 0);
                    }
                }
                if (KSPLIT != 4) {
                const float* ap = smem + l15 * LDH + ks * KBW * 16 + lg * 4;
#pragma unroll
                for (int hh = 0; hh < NH; ++hh) {
#pragma unroll
                    for (int k = hh * (KBW / NH); k < (hh + 1) * (KBW / NH); k += 2) {
                        const float4 a0 = *reinterpret_cast<const float4*>(ap + k * 16);
                        const float4 a1 = *reinterpret_cast<const float4*>(ap + (k + 1) * 16);
                        chn[hh][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.x, wreg[k].x, chn[hh][0], 0, 0, 0);
                        chn[hh][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.x, wreg[k + 1].x, chn[hh][1], 0, 0, 0);
                        chn[hh][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.y, wreg[k].y, chn[hh][0], 0, 0, 0);
                        chn[hh][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.y, wreg[k + 1].y, chn[hh][1], 0, 0, 0);
                        chn[hh][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.z, wreg[k].z, chn[hh][0], 0, 0, 0);
                        chn[hh][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.z, wreg[k + 1].z, chn[hh][1], 0, 0, 0);
                        chn[hh][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.w, wreg[k].w, chn[hh][0], 0, 0, 0);
                        chn[hh][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.w, wreg[k + 1].w, chn[hh][1], 0, 0, 0);
                    }
                }
                }
            }
            f32x4 acc = KSPLIT == 4 ? chn[0][0] : chn[0][0] + chn[0][1];
            if (NH == 2) acc = acc + (chn[NH - 1][0] + chn[NH - 1][1]);
            if (TRACE && blockIdx.x == 0 && tid == 0 && t < 64) {
                asm volatile("" :: "v"(acc[0]));
                g_rnn_trace[t * 4 + 1] = __builtin_amdgcn_s_memtime();
            }
            if (KSPLIT > 1) {
                if (ks > 0) *reinterpret_cast<f32x4*>(red + (((ks - 1) * CBW + cb) * 64 + lane) * 4) = acc;
                __syncthreads();
                if (ks == 0) {
                    if (KSPLIT == 4) {   // (c00 + c01) + (c10 + c11): the canonical combination of the four chains
                        const f32x4 c01 = *reinterpret_cast<const f32x4*>(red + ((0 * CBW + cb) * 64 + lane) * 4);
                        const f32x4 c10 = *reinterpret_cast<const f32x4*>(red + ((1 * CBW + cb) * 64 + lane) * 4);
                        const f32x4 c11 = *reinterpret_cast<const f32x4*>(red + ((2 * CBW + cb) * 64 + lane) * 4);
                        acc = (acc + c01) + (c10 + c11);
                    } else {
#pragma unroll
                        for (int q = 1; q < KSPLIT; ++q) acc += *reinterpret_cast<const f32x4*>(red + (((q - 1) * CBW + cb) * 64 + lane) * 4);
                    }
                }
            }
            if (ks == 0) {
                // D layout of 16x16: col = lane&15, row = 4*(lane>>4) + r.  sc1 (write-through) stores.
                // All four values are finished BEFORE the first store is issued: gfx950's vmcnt also counts stores, so a
                // store between two uses of the prefetched `ihv` makes the compiler wait for its write-through ack.
                float hv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    hv[r] = BWD ? (acc[r] + ihv[r]) * (1.0f - gv[r] * gv[r]) : tip_tanh(acc[r] + ihv[r]);
                    if (hv[r] != hv[r]) hv[r] = __uint_as_float(kPoisonBits);   // poison travels as the canonical NaN, never as the sentinel
                }
                // pin the four values here (the optimiser otherwise sinks each tanh back into its store's branch)
                asm volatile("" : "+v"(hv[0]), "+v"(hv[1]), "+v"(hv[2]), "+v"(hv[3]));
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int bb = b0 + lg * 4 + r;
                    if (bb < B) {
                        float* dst = hall + ((size_t)bb * T + te) * R + nb * 16 + l15;
                        if (same_xcd) *dst = hv[r];   // lands in the shared L2 (L1 is write-through)
                        else __hip_atomic_store(dst, hv[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1
                    }
                }
            }
            if (TRACE && blockIdx.x == 0 && tid == 0 && t < 64) g_rnn_trace[t * 4 + 2] = __builtin_amdgcn_s_memtime();
            if (HANDOFF == 0) {
                // publish: every storing wave drains its stores, workgroup barrier, one arrival
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0)
                    __hip_atomic_fetch_add(flags + (size_t)tile * T + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else if (KSPLIT == 1) {
                __syncthreads();   // the h tile in LDS is rewritten by the next step's pull
            }
        }
    }
    // Leave the XCC-exchange word cleared: a REPLAY of this launch from a captured HIP graph carries the same tag, and must wait
    // for that launch's words instead of finding this one's.  (T >= 2: a member that has finished has consumed every partner's
    // step-0 state, which a partner stores only after its own exchange — nobody can still be polling this word.  With T = 1
    // there is no such ordering and the word stays.)
    if (HANDOFF == 1 && T >= 2 && tid == 0)
        __hip_atomic_store(flags + group * CLUSTER + cid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ------------------------------------------------------------------------------------------------
// RNN recurrence on FOUR-ROW tiles (R = 512): 4-workgroup clusters, v_mfma_f32_4x4x1_16b_f32.
//
// The 16-row kernel above is bound by its hand-off, not by arithmetic: per step 16 workgroups all-gather a 32-KB tile from 16
// producers through one L2 (1 750 cycles), four waves' k-partials meet in LDS behind a barrier (2 100 with tanh and stores),
// against 2 300 cycles of MFMA.  Here a tile is FOUR windows and a cluster FOUR workgroups: each workgroup owns 128 columns of
// W_hh, each of its 8 waves 16 columns x all 512 k — 128 VGPRs, the same fragment registers as the <8, 1> variant — and
// multiplies with the 4x4x1 MFMA exactly as the hybrid encoder's tail does (tip_fused.hip): lane (l15, lg) of a 16x16x4
// B fragment is column l15 at k = 16 kb + 4 lg + s, block (lg, l15 >> 2) of the 4x4x1; the A operand is h[row lane & 3] at
// the same k (one ds_read_b128 per k-block, conflict-free with a row stride of 528 floats).  So
//   * the tile a member pulls per step is 8 KB from 4 producers — ONE 16-byte load per thread — instead of 32 KB from 16;
//   * a wave owns its 16 columns for every k: the four k-partials of an output sit in the four lane groups of ONE accumulator
//     and meet through three permlane swaps — no LDS reduction, no second barrier;
//   * the matrix-pipe time is unchanged (1 024 4x4x1 MFMAs per workgroup and step at 8+ cycles = 16x16x4's FLOP rate) and
//     nothing is computed on pad rows.
// A cluster serves up to kQ4Tiles tiles at once (their hops overlap; the MFMA phases run back to back) and further tiles in
// sequence.  Hand-off protocol, XCD verification, sentinel, poison and fault injection: exactly rnn_resident_kernel's.
// Numerics: k is summed in four chains per lane group (s = 0..3 of every k-block) — (c0 + c1) + (c2 + c3), then
// (p[lg] + p[lg ^ 2]) + (p[lg ^ 1] + p[lg ^ 3]) — for every batch size; it differs from the 16-row kernels in summation
// order only (they are bit-identical among themselves, this one with itself).
// ------------------------------------------------------------------------------------------------
constexpr int kQ4Rows = 4, kQ4Cluster = 4, kQ4Tiles = 4, kQ4LD = 512 + 16;

// NT = tiles a cluster advances together (1, 2 or 4).  The step loop is straight-line code: a wave issues one VALU instruction
// per 4 cycles and pays ~16 for every taken branch, and with two waves per SIMD a few hundred of either per step cost as much
// as the MFMAs (the first version of this kernel spent 1 500 cycles per step outside MFMAs and memory).  Hence
//   * every memory access of a batch is a buffer instruction: a per-lane byte offset fixed for the batch plus a SCALAR offset
//     time * R * 4 — no per-step address arithmetic on the vector ALU;
//   * no predication: rows past the end of the batch (and whole tiles past it, in a ragged last batch) have a per-lane offset
//     >= hall_bytes, so the descriptor's range check returns zeros for their loads and drops their stores (the scalar offset is
//     not part of that check: the row term alone decides); such tiles multiply zeros;
//   * the poll loop is per WAVE: all its lanes re-ask until none of them sees a sentinel.
// WAVES = 4 (round 4): a member is a 4-wave workgroup owning 64 columns, a cluster EIGHT members, and TWO workgroups — members of
// different clusters, i.e. different tiles — share a CU: the matrix work per CU is what it was, but while one workgroup sits in its
// hand-off (pull, LDS, barrier: 0.7 us of a 1.9-us step) the other one's MFMAs have the pipe.  Same fragments per wave (16 columns x
// all k), same accumulation order: bit-identical to WAVES = 8.
template <int NT, bool TRACE, bool BWD, int WAVES = 8>
__global__ __launch_bounds__(WAVES * 64) void rnn_rows4_kernel(const float* __restrict__ ih, const float* __restrict__ whh_frag,
                                                        float* __restrict__ hall, unsigned* __restrict__ flags, int B, int T,
                                                        int ntiles, int hall_bytes, const float* __restrict__ gate, Guard gd,
                                                        unsigned etag, int abl, int ngroups) {
    // abl: MEASUREMENT-ONLY ablations (wrong results), TIP_RNN_ABLATE: 1 = polls never wait, 2 = no MFMAs
    constexpr int R = 512, KB = R / 16, CLUSTER = 32 / WAVES, LD = kQ4LD;
    constexpr int THREADS = WAVES * 64, PL = 8 / WAVES;            // 16-byte pieces of a pulled tile per thread: 1 (8 waves) or 2 (4 waves)
    static_assert(WAVES == 8 || WAVES == 4, "cluster of 4 or 8 members");
    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2 buffers][NT tiles][4 rows][LD]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    if (TRACE && blockIdx.x == 0 && tid == 0) g_rnn_trace[3] = __builtin_amdgcn_s_memtime();   // kernel entry
    // cluster membership: members are 8 workgroup ids apart — one XCD as workgroups are observed to be dealt (verified below,
    // never assumed); the grid is whole rounds of 8 clusters, the clusters past `ngroups` have nothing to do
    const int cid = (blockIdx.x >> 3) % CLUSTER, group = (blockIdx.x & 7) + 8 * ((blockIdx.x >> 3) / CLUSTER);
    if (group >= ngroups) return;
    const int nb = cid * WAVES + wave;                            // global 16-column block of this wave
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(hall, 0, hall_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ih), 0, hall_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(BWD ? gate : ih), 0, hall_bytes, 0x00020000);
    // TIP_OPT_FAULT_INJECT: this member never arrives
    const bool dead = (gd.fault & 2) && group == 0 && cid == 1;
    if (dead) return;
    const unsigned spin_big = guard_spin_limit(gd.fault, 1u << 22), spin_pull = guard_spin_limit(gd.fault, 1u << 20);

    // this thread's 16-byte pieces of a pulled tile: row prow, columns pcol + 4 i (i < PL)
    const int prow = tid / (THREADS / 4), pcol = (tid % (THREADS / 4)) * (4 * PL);
    const int tpg = (ntiles + ngroups - 1) / ngroups;             // tiles per cluster
    const unsigned rowbytes = (unsigned)T * R * 4;
    int vpull[NT], vout[NT];
    auto set_tiles = [&](int q0) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            int tile = group + ngroups * (q0 + n);
            if (q0 + n >= tpg || tile >= ntiles) tile = ntiles;   // past the batch: rows >= B, out of the descriptor's range
            vpull[n] = (int)((unsigned)(tile * kQ4Rows + prow) * rowbytes + (unsigned)pcol * 4u);
            vout[n] = (int)((unsigned)(tile * kQ4Rows + lg) * rowbytes + (unsigned)(nb * 16 + l15) * 4u);
        }
    };
    // The input term (and the backward's gate) of step t + 1 is requested during step t's MFMA phase, AFTER its pull: vector
    // memory returns in order, so an HBM-latency load issued in front of the poll loads would hold every one of them back.
    float ihn[NT], gn[NT];
    auto request_inputs = [&](int te_) {
        const int so = te_ * (R * 4);
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            ihn[n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(irs, vout[n], so, 0));
            gn[n] = BWD ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(grs, vout[n], so, 0)) : 0.f;
        }
    };
    // The rows the producers are about to write sit in HBM / Infinity Cache (the encoder armed them with the sentinel long
    // ago): a 64-byte store then allocates a PARTIALLY valid line in L2, and the first poll of that line waits for its fill
    // from memory.  Each member therefore touches (one 16-byte load per thread and tile, result unused) the rows of step
    // t + 2 during step t: the producers' stores then hit valid lines and the polls are plain L2 hits.
    u32x4 pfv[NT];
    auto touch_rows = [&](int te_) {
        const int so = te_ * (R * 4);
#pragma unroll
        for (int n = 0; n < NT; ++n) pfv[n] = __builtin_amdgcn_raw_buffer_load_b128(hrs, vpull[n], so, 16);
    };
    auto retire_touch = [&]() {
#pragma unroll
        for (int n = 0; n < NT; ++n) asm volatile("" :: "v"(pfv[n]));
    };
    // Step 0 of the first batch needs no W_hh (h_{-1} = 0): its input term and row touches are requested IN FRONT of the 256-KB
    // weight load (vector memory returns in order), so that step is over when the weights land instead of starting then.
    set_tiles(0);
    request_inputs(BWD ? T - 1 : 0);
    touch_rows(BWD ? T - 1 : 0);
    // W_hh slice -> registers, once: all 32 k-blocks of this wave's 16 columns (in flight during the exchange below)
    float4 wreg[KB];
    {
        const float4* wf = reinterpret_cast<const float4*>(whh_frag) + (size_t)nb * KB * 64 + lane;
#pragma unroll
        for (int k = 0; k < KB; ++k) wreg[k] = wf[(size_t)k * 64];
    }
    // same-XCD fast path, verified at run time through launch-tagged exchange words (see rnn_resident_kernel)
    __shared__ int s_same_xcd;
    if (tid == 0 && dead) s_same_xcd = 0;
    if (tid == 0 && !dead) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xf;
        const unsigned mine = (etag << 5) | (xcc + 1u);
        __hip_atomic_store(flags + group * CLUSTER + cid, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool same = true;
        for (int m = 0; m < CLUSTER; ++m) {
            unsigned v = 0;
            bool here = false;
            for (unsigned spins = 0; spins < spin_big; ++spins) {
                v = __hip_atomic_load(flags + group * CLUSTER + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                here = (v >> 5) == etag && (v & 31u) != 0u;
                if (here) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!here) note_spin_timeout(gd.err);
            same &= (v == mine);
        }
        s_same_xcd = (same && !(gd.fault & 8)) ? 1 : 0;
    }
    __syncthreads();
    const bool same_xcd = s_same_xcd != 0;
    if (TRACE && blockIdx.x == 0 && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g_rnn_trace[7] = __builtin_amdgcn_s_memtime();   // XCC exchange done, W_hh slice in registers
    }
    const int aoff = (lane & 3) * LD + lg * 4;                    // this lane's A operand: row lane & 3, k = 16 kb + 4 lg ..
    const int lds_w = prow * LD + pcol;                            // where this thread's 16 bytes of a pulled tile go
    bool poisoned = false;                                         // (wave-uniform)
    if (TRACE && tid == 0 && blockIdx.x < 512) {                   // which workgroups share a CU (tools/rnn_hwid.py: ids j and j + 32 of an XCD)
        unsigned hwid, xccid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xccid));
        g_rnn_trace[1024 + blockIdx.x] = (unsigned long long)hwid | ((unsigned long long)(xccid & 0xf) << 32);
    }

    u32x4 pv[PL];                                                  // early-requested pull of the next tile (NT > 1)
#pragma unroll
    for (int i = 0; i < PL; ++i) pv[i] = (u32x4){kRnnSentinel, kRnnSentinel, kRnnSentinel, kRnnSentinel};
    // The whole step loop exists twice, for partners on this XCD (plain stores) and elsewhere (sc1 stores, paced polls): as a
    // branch around the store inside the loop, the structurised control flow has a path without a store, and the compiler then
    // cannot count the stores in flight — every wait behind it (the next tile's early pull) became vmcnt(0), i.e. waited for
    // the store's acknowledgement.
    auto run_steps = [&](auto same_tag) {
        constexpr bool SAME_XCD = decltype(same_tag)::value;
        for (int q0 = dead ? tpg : 0; q0 < tpg; q0 += NT) {
            if (q0 > 0) {
                set_tiles(q0);
                request_inputs(BWD ? T - 1 : 0);
                touch_rows(BWD ? T - 1 : 0);
            }
            retire_touch();
            if (T > 1) touch_rows(BWD ? T - 2 : 1);
    #pragma unroll 1
            for (int t = 0; t < T; ++t) {
                const int te = BWD ? T - 1 - t : t;        // time index this step produces
                const int tp = BWD ? te + 1 : te - 1;      // time index of the state it consumes
                float ihv[NT], gv[NT];
    #pragma unroll
                for (int n = 0; n < NT; ++n) ihv[n] = ihn[n], gv[n] = gn[n];
                float* buf = smem + (t & 1) * (NT * kQ4Rows * LD);
                const int so_out = te * (R * 4);
                // Tile by tile: pull -> LDS -> barrier -> MFMAs -> tanh -> store.  With several tiles per cluster (NT > 1) the pull of the
                // NEXT tile (the first tile's of the next step behind the last one) is requested in front of this tile's matrix phase:
                // that tile's state was stored one to NT - 1 matrix phases ago, so the request finds it in L2 and its round trip runs
                // under the MFMAs; only a request that still saw a sentinel falls back to the polling loop.  Pulling every tile at the
                // top of the step left the last tile's hop exposed on every step (first version); pulling tile by tile WITHOUT the early
                // request exposed an L2 round trip + LDS write + barrier per tile (B = 1024: 251 us against 235).  One barrier per
                // tile and step: the tile buffers alternate with t, so nobody overwrites what a slow wave still reads.  Same
                // arithmetic per tile, bit-identical.
    #pragma unroll
                for (int n = 0; n < NT; ++n) {
                    if (t > 0) {
                        // pull h_{t-1}: 16 bytes per thread, sc1 loads (agent-scope coherent); the wave re-asks while any of its lanes
                        // still sees a sentinel word
                        const int so = tp * (R * 4);
                        u32x4 v[PL];
                        bool pend = true;
                        if (NT > 1) {
                            pend = false;
    #pragma unroll
                            for (int i = 0; i < PL; ++i) {
                                v[i] = pv[i];
                                pend |= v[i].x == kRnnSentinel || v[i].y == kRnnSentinel || v[i].z == kRnnSentinel || v[i].w == kRnnSentinel;
                            }
                        }
                        bool gave_up = false;
                        if (NT == 1 || (__builtin_amdgcn_ballot_w64(pend) != 0 && !(abl & 1))) {   // (wave-uniform)
                            gave_up = true;
                            const unsigned pull_lim = poisoned ? 1u : spin_pull;
                            for (unsigned spins = 0; spins < pull_lim; ++spins) {
                                pend = false;
                                asm volatile("" ::: "memory");   // the addresses are loop invariant: without this the optimiser polls a register
    #pragma unroll
                                for (int i = 0; i < PL; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(hrs, vpull[n] + 16 * i, so, 16);
    #pragma unroll
                                for (int i = 0; i < PL; ++i)
                                    pend |= v[i].x == kRnnSentinel || v[i].y == kRnnSentinel || v[i].z == kRnnSentinel || v[i].w == kRnnSentinel;
                                if (TRACE && n == 0 && !(abl & 128) && blockIdx.x == 0 && tid == 0 && t < 24) {
                                    if (spins == 0) g_rnn_trace[160 + (t - 1) * 4 + 0] = __builtin_amdgcn_s_memtime();
                                    g_rnn_trace[160 + (t - 1) * 4 + 3] = spins + 1;
                                }
                                if (__builtin_amdgcn_ballot_w64(pend) == 0 || (abl & 1)) { gave_up = false; break; }
                                if (!SAME_XCD) __builtin_amdgcn_s_sleep(2);   // cross-XCD polls travel the fabric: pace them
                            }
                        }
                        if (TRACE && n == 0 && !(abl & 128) && blockIdx.x == 0 && tid == 0 && t < 24) g_rnn_trace[160 + (t - 1) * 4 + 1] = __builtin_amdgcn_s_memtime();
                        if (TRACE && n == 0 && (abl & 128) && ((abl >> 8) & 7) == 3 && blockIdx.x == 0 && tid == 0 && t < 64) g_rnn_trace[t * 4 + 1] = __builtin_amdgcn_s_memtime();
                        if (gave_up) {   // (wave-uniform)
                            if (!poisoned && lane == 0) note_spin_timeout(gd.err);
                            poisoned = true;
    #pragma unroll
                            for (int i = 0; i < PL; ++i) {   // what never arrived becomes NaN, not the sentinel
                                if (v[i].x == kRnnSentinel) v[i].x = kPoisonBits;
                                if (v[i].y == kRnnSentinel) v[i].y = kPoisonBits;
                                if (v[i].z == kRnnSentinel) v[i].z = kPoisonBits;
                                if (v[i].w == kRnnSentinel) v[i].w = kPoisonBits;
                            }
                        }
    #pragma unroll
                        for (int i = 0; i < PL; ++i) *reinterpret_cast<u32x4*>(buf + n * kQ4Rows * LD + lds_w + 4 * i) = v[i];
                        if (TRACE && n == 0 && !(abl & 128) && blockIdx.x == 0 && tid == 0 && t < 24) {
                            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            g_rnn_trace[160 + (t - 1) * 4 + 2] = __builtin_amdgcn_s_memtime();
                        }
                        __syncthreads();
                    }
                    if (TRACE && n == 0 && !(abl & 128) && blockIdx.x == 0 && tid == 0 && t < 64) g_rnn_trace[t * 4 + 0] = __builtin_amdgcn_s_memtime();
                    if (TRACE && n == 0 && (abl & 128) && ((abl >> 8) & 7) == 1 && blockIdx.x == 0 && tid == 0 && t < 64) g_rnn_trace[t * 4 + 1] = __builtin_amdgcn_s_memtime();
                    if (n == NT - 1) {
                        // the next step's input terms and the row touches of the step after it, behind the LAST pull of this step: vector
                        // memory returns in order, so in front of a pull they would hold it back by their HBM latency
                        // (clamped to the last step instead of skipped: branch-free; the extra loads hit rows this launch owns)
                        retire_touch();
                        const int tn = t + 1 < T ? t + 1 : t, tn2 = t + 2 < T ? t + 2 : t;
                        request_inputs(BWD ? T - 1 - tn : tn);
                        if (!(abl & 4)) touch_rows(BWD ? T - 1 - tn2 : tn2);
                    }
                    if (NT > 1) {
                        // early request of the next pull: tile n + 1 of this step, or tile 0 of the next one
                        const int nn = (n + 1) % NT;
                        const int tq = n == NT - 1 ? t + 1 : t;              // the step that will consume it
                        // (always requested, never behind a branch: a loaded value that meets a constant at a join makes the compiler wait
                        // for it there; steps 0 and T do not pull — their request is clamped to a row of this launch and dropped)
                        const int tqc = tq < 1 ? 1 : (tq > T - 1 ? (T > 1 ? T - 1 : 1) : tq);
                        const int soq = (BWD ? T - tqc : tqc - 1) * (R * 4);
    #pragma unroll
                        for (int i = 0; i < PL; ++i) pv[i] = __builtin_amdgcn_raw_buffer_load_b128(hrs, vpull[nn] + 16 * i, soq, 16);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0;
                    if (t > 0 && !(abl & 2)) {
                        // A fragments in batches of AB k-blocks, NBUF batches in flight (LDS latency is 2-4 k-blocks of 4x4x1 issue time).  A batch
                        // is REQUESTED last k-block first and CONSUMED first k-block first: LDS returns in order, so the wait for the first
                        // fragment used covers the whole batch — one s_waitcnt per batch instead of one per k-block (a wave issues in order,
                        // and with a wait + a read between every four 8-cycle MFMAs it reached 73 % of the pipe's rate).  Measured (B = 256 /
                        // 1024, us): one k-block at a time 76.5 / 234; batches of 2: 75.4 / 244; of 8: 73.8 / 235.5; of 4, two in flight: 72.0-73.1 /
                        // 233 — the first MFMA of a step waits for four fragments, not eight.  The order of the sum over k is unchanged.
                        constexpr int AB = 4, NBUF = 2;
                        const float* ap = buf + n * kQ4Rows * LD + aoff;
                        float4 a[NBUF][AB];
                        auto request = [&](int b) {
    #pragma unroll
                            for (int j = AB - 1; j >= 0; --j) a[b % NBUF][j] = *reinterpret_cast<const float4*>(ap + (b * AB + j) * 16);
                        };
    #pragma unroll
                        for (int b = 0; b < NBUF; ++b) request(b);
                        __builtin_amdgcn_sched_barrier(0);
    #pragma unroll
                        for (int b = 0; b < KB / AB; ++b) {
    #pragma unroll
                            for (int j = 0; j < AB; ++j) {
                                const float4 ak = a[b % NBUF][j];
                                const int k = b * AB + j;
                                c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(ak.x, wreg[k].x, c0, 0, 0, 0);
                                c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(ak.y, wreg[k].y, c1, 0, 0, 0);
                                c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(ak.z, wreg[k].z, c2, 0, 0, 0);
                                c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(ak.w, wreg[k].w, c3, 0, 0, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
#ifdef TIP_MEASURE
                            // MEASUREMENT (TIP_RNN_ABLATE bit 4, wrong results): only the first NBUF batches of A fragments are read from
                            // LDS, the MFMAs of the later ones reuse their registers — 8 instead of 32 ds_read_b128 per wave and tile:
                            // what the matrix phase costs without its LDS traffic
                            if (b + NBUF < KB / AB && !(abl & 16)) request(b + NBUF);
#else
                            if (b + NBUF < KB / AB) request(b + NBUF);
#endif
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    const f32x4 p = (c0 + c1) + (c2 + c3);   // registers = rows 0..3, one k-partial per lane group
                    // reduce-scatter over the lane groups: lane (lg, l15) ends with row lg (as tip_fused.hip's tail_reduce)
                    float a0 = p[0], a2 = p[2];
                    swap32(a0, a2);
                    float k0 = a0 + a2;
                    float a1 = p[1], a3 = p[3];
                    swap32(a1, a3);
                    float k1 = a1 + a3;
                    swap16(k0, k1);
                    const float acc = k0 + k1;
                    if (TRACE && !(abl & 128) && n == 0 && blockIdx.x == 0 && tid == 0 && t < 64) {
                        asm volatile("" :: "v"(acc));
                        g_rnn_trace[t * 4 + 1] = __builtin_amdgcn_s_memtime();
                    }
                    if (TRACE && (abl & 128) && ((abl >> 8) & 7) == 2 && n == 0 && blockIdx.x == 0 && tid == 0 && t < 64) {
                        asm volatile("" :: "v"(c0[0]), "v"(c1[0]), "v"(c2[0]), "v"(c3[0]));   // MFMAs of tile 0 done, before the reduction
                        g_rnn_trace[t * 4 + 1] = __builtin_amdgcn_s_memtime();
                    }
#ifdef TIP_MEASURE
                    // MEASUREMENT (TIP_RNN_ABLATE bit 5, wrong results): a clamp instead of tanh (exp + IEEE division, ~25 dependent
                    // vector instructions between the last MFMA and the state store): what the activation costs on the serial chain
                    float hv = BWD ? (acc + ihv[n]) * (1.0f - gv[n] * gv[n])
                                   : ((abl & 32) ? fminf(fmaxf(acc + ihv[n], -1.0f), 1.0f) : tip_tanh(acc + ihv[n]));
#else
                    float hv = BWD ? (acc + ihv[n]) * (1.0f - gv[n] * gv[n]) : tip_tanh(acc + ihv[n]);
#endif
                    if (hv != hv) hv = __uint_as_float(kPoisonBits);   // poison travels as the canonical NaN, never as the sentinel
                    // same XCD (verified): a plain store lands in the shared L2 (L1 is write-through); otherwise sc1 = write-through
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(hv), hrs, vout[n], so_out, SAME_XCD ? 0 : 16);
                }
                if (TRACE && blockIdx.x == 0 && tid == 0 && t < 64) g_rnn_trace[t * 4 + 2] = __builtin_amdgcn_s_memtime();
            }
            retire_touch();
            __syncthreads();   // the next batch's second step rewrites the buffer the last step of this one may still be reading
        }
    };
    if (same_xcd) run_steps(std::true_type{});
    else run_steps(std::false_type{});
    // leave the XCC-exchange word cleared for a replay of this launch from a HIP graph (same tag): see rnn_resident_kernel
    if (T >= 2 && tid == 0 && !dead) __hip_atomic_store(flags + group * CLUSTER + cid, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

size_t rnn_flag_words(int B, int T) { return (size_t)((B + kRnnTile - 1) / kRnnTile) * (size_t)T + 1024; }

// ONE process-wide launch tag for every kernel family that tags the XCC-exchange words of a workspace (rnn_resident_kernel and
// rnn_rows4_kernel write the same `flags` words): with a counter each, a stale word left by one family could match the other's
// tag and a wait would be satisfied by a stale XCC id.
unsigned next_rnn_launch_tag() {
    static std::atomic<unsigned> launch_tag{[] { std::random_device rd; return (unsigned)rd(); }()};
    return launch_tag.fetch_add(1u) & 0x7FFFFFu;   // 23 bits: fits the int kernel argument above the option bits
}

static int rnn_handoff_mode() {
    static int handoff = -1;
    if (handoff < 0) {
        const char* e = tip_env("TIP_RNN_HANDOFF");   // 0 = arrival counter, 1 = sentinel polling (default)
        handoff = e ? atoi(e) : 1;
    }
    return handoff;
}

bool rnn_uses_sentinel(const Dims& d, int B, int T, int cluster) {
    return d.R == 512 && (long long)B * T * 512 * 4 <= 0x7fffffffLL && cluster >= 4 && rnn_handoff_mode() == 1;
}

template <int WAVES, int KSPLIT>
static hipError_t launch_rnn_resident(const float* ih, const float* whh_frag, float* hall, unsigned* flags, int B, int T,
                                      int ntiles, int num_cus, bool hall_armed, const Guard& gd, hipStream_t s,
                                      const float* gate = nullptr, int wg_per_cu = 1) {
    constexpr int CLUSTER = (512 / 16) / (WAVES / KSPLIT);
    int groups = ntiles;
    const int maxg = num_cus * wg_per_cu / CLUSTER > 0 ? num_cus * wg_per_cu / CLUSTER : 1;   // keep every cluster co-resident
    if (groups > maxg) groups = maxg;
    const int handoff = rnn_handoff_mode();
    const size_t smem = ((size_t)kRnnTile * (512 + 4) + (size_t)(KSPLIT - 1) * (WAVES / KSPLIT) * 256) * sizeof(float);
    {
        // every member of a cluster must be resident while its partners wait for it: ask the runtime, do not assume
        static PerDeviceInt occ_dev; int& occ = occ_dev.cur();
        hipError_t ce = check_coresident(rnn_resident_kernel<WAVES, KSPLIT, 1>, WAVES * 64, smem, groups * CLUSTER, num_cus, &occ);
        if (ce != hipSuccess) return ce;
    }
    const long long hb = (long long)B * T * 512 * 4;
    if (hb > 0x7fffffffLL) return hipErrorInvalidValue;
    if (handoff == 0) {
        const int prepoll = 0;
        hipError_t e = hipMemsetAsync(flags, 0, (size_t)ntiles * T * sizeof(unsigned), s);
        if (e != hipSuccess) return e;
        if (gate)
            hipLaunchKernelGGL((rnn_resident_kernel<WAVES, KSPLIT, 0, false, true>), dim3(groups * CLUSTER), dim3(WAVES * 64), smem,
                               s, ih, whh_frag, hall, flags, B, T, ntiles, (int)hb, gate, gd, prepoll);
        else
            hipLaunchKernelGGL((rnn_resident_kernel<WAVES, KSPLIT, 0>), dim3(groups * CLUSTER), dim3(WAVES * 64), smem, s, ih,
                               whh_frag, hall, flags, B, T, ntiles, (int)hb, gate, gd, prepoll);
    } else {
        if (!hall_armed) {
            hipError_t e = hipMemsetAsync(hall, 0xFF, (size_t)hb, s);   // every word = kRnnSentinel
            if (e != hipSuccess) return e;
        }
        // XCC-id exchange words: tagged with a per-launch number instead of being zeroed (see the kernel)
        const unsigned etag = next_rnn_launch_tag();
        static int trace = -1;
        if (trace < 0) trace = tip_env("TIP_RNN_TRACE") ? 1 : 0;
        static int prepoll_env = -1;   // TIP_RNN_PREPOLL=0: no arrival probe before the tile pull (measurement)
        if (prepoll_env < 0) prepoll_env = (tip_env("TIP_RNN_PREPOLL") && tip_env("TIP_RNN_PREPOLL")[0] == '0') ? 0 : 1;
        static int rot_env = -1;       // TIP_RNN_ROTATE=0: every member sweeps the tile from row 0 (measurement)
        if (rot_env < 0) rot_env = (tip_env("TIP_RNN_ROTATE") && tip_env("TIP_RNN_ROTATE")[0] == '0') ? 0 : 2;
        const int prepoll = prepoll_env | rot_env | (int)(etag << 8);
        if (gate)
            hipLaunchKernelGGL((rnn_resident_kernel<WAVES, KSPLIT, 1, false, true>), dim3(groups * CLUSTER), dim3(WAVES * 64), smem,
                               s, ih, whh_frag, hall, flags, B, T, ntiles, (int)hb, gate, gd, prepoll);
        else if (trace)
            hipLaunchKernelGGL((rnn_resident_kernel<WAVES, KSPLIT, 1, true>), dim3(groups * CLUSTER), dim3(WAVES * 64), smem, s,
                               ih, whh_frag, hall, flags, B, T, ntiles, (int)hb, gate, gd, prepoll);
        else
            hipLaunchKernelGGL((rnn_resident_kernel<WAVES, KSPLIT, 1>), dim3(groups * CLUSTER), dim3(WAVES * 64), smem, s, ih,
                               whh_frag, hall, flags, B, T, ntiles, (int)hb, gate, gd, prepoll);
    }
    return hipGetLastError();
}

// four-row tiles on 4-workgroup clusters of 8-wave members, or (WAVES = 4) 8-workgroup clusters of 4-wave members, two per CU
// (rnn_rows4_kernel); sentinel hand-off only
template <int NT, int WAVES>
static hipError_t launch_rnn_rows4_nt(const float* ih, const float* whh_frag, float* hall, unsigned* flags, int B, int T, int ntiles,
                                      int groups, long long hb, const Guard& gd, hipStream_t s, const float* gate, unsigned etag, int num_cus) {
    constexpr int smem = 2 * NT * kQ4Rows * kQ4LD * (int)sizeof(float);
    constexpr int CLUSTER = 32 / WAVES, THREADS = WAVES * 64;
    static int trace = -1, abl = -1;
    if (trace < 0) trace = tip_env("TIP_RNN_TRACE") ? 1 : 0;
    if (abl < 0) abl = tip_env("TIP_RNN_ABLATE") ? atoi(tip_env("TIP_RNN_ABLATE")) : 0;   // measurement only (profiles/): never set in production
    static PerDeviceFlag attr_flag; bool& attr_set = attr_flag.cur();
    if (!attr_set) {
        for (const void* f : {reinterpret_cast<const void*>(rnn_rows4_kernel<NT, false, false, WAVES>), reinterpret_cast<const void*>(rnn_rows4_kernel<NT, true, false, WAVES>),
                              reinterpret_cast<const void*>(rnn_rows4_kernel<NT, false, true, WAVES>)}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            if (e != hipSuccess) return e;
        }
        attr_set = true;
    }
    static PerDeviceInt occ_dev; int& occ = occ_dev.cur();   // every member of a cluster must be resident while its partners wait for it: ask the runtime
    hipError_t ce = check_coresident(rnn_rows4_kernel<NT, false, false, WAVES>, THREADS, smem, groups * CLUSTER, num_cus, &occ);
    if (ce != hipSuccess) return ce;
    // Members of a cluster are taken 8 workgroup ids apart (one XCD); that needs a grid of whole rounds of 8 clusters.  The
    // workgroups of the clusters that pad the last round exit at once (they hold no resources anybody waits for).
    const dim3 grid((groups + 7) / 8 * 8 * CLUSTER), block(THREADS);
    if (gate)
        hipLaunchKernelGGL((rnn_rows4_kernel<NT, false, true, WAVES>), grid, block, smem, s, ih, whh_frag, hall, flags, B, T, ntiles, (int)hb, gate, gd, etag, abl, groups);
    else if (trace)
        hipLaunchKernelGGL((rnn_rows4_kernel<NT, true, false, WAVES>), grid, block, smem, s, ih, whh_frag, hall, flags, B, T, ntiles, (int)hb, gate, gd, etag, abl, groups);
    else
        hipLaunchKernelGGL((rnn_rows4_kernel<NT, false, false, WAVES>), grid, block, smem, s, ih, whh_frag, hall, flags, B, T, ntiles, (int)hb, gate, gd, etag, abl, groups);
    return hipGetLastError();
}

static hipError_t launch_rnn_rows4(const float* ih, const float* whh_frag, float* hall, unsigned* flags, int B, int T, int num_cus,
                                   bool hall_armed, const Guard& gd, hipStream_t s, const float* gate = nullptr) {
    const int ntiles = (B + kQ4Rows - 1) / kQ4Rows;
    // 4-wave members, 8 per cluster: when the batch leaves CUs idle (tiles x 8 <= #CUs, i.e. B <= 128 on a full part) every member
    // gets a CU of its own with half the matrix work per step: 62 vs 76 us at B = 100.  With two members per CU (B = 256) the hoped-for
    // overlap of one's hand-off with the other's MFMAs does not happen: 84 vs 78 us (profiles/r04/rnn_w4.txt), so the 8-wave members
    // stay there.  Bit-identical either way.  TIP_RNN_W4=0 / 1 forces one (measurement).
    static int w4 = -2;
    if (w4 == -2) w4 = tip_env("TIP_RNN_W4") ? (tip_env("TIP_RNN_W4")[0] == '1' ? 1 : 0) : -1;
    const bool use_w4 = (w4 == 1 || (w4 == -1 && ntiles * 8 <= num_cus));
    int groups = ntiles;
    const int maxg = num_cus / kQ4Cluster > 0 ? num_cus / kQ4Cluster : 1;   // keep every cluster co-resident (W4: 8 members, two per CU: the same count)
    if (groups > maxg) groups = maxg;
    const int tpg = (ntiles + groups - 1) / groups;                          // tiles per cluster
    const long long hb = (long long)B * T * 512 * 4;
    if (hb > 0x7fffffffLL) return hipErrorInvalidValue;
    // the kernel forms (tile * 4 + pad row) * row bytes in 32-bit unsigned arithmetic for up to 7 rows past the batch: those
    // offsets must stay above the descriptor's range WITHOUT wrapping back into it (a window of > 149 k frames would)
    if ((long long)(B + 7) * T * 512 * 4 > 0xffffffffLL) return hipErrorInvalidValue;
    if (!hall_armed) {
        hipError_t e = hipMemsetAsync(hall, 0xFF, (size_t)hb, s);   // every word = kRnnSentinel
        if (e != hipSuccess) return e;
    }
    const unsigned etag = next_rnn_launch_tag();
    // tiles a cluster advances together: as many as it owns, up to kQ4Tiles (3 -> 4: the pad tile is out of range and multiplies zeros)
    if (use_w4) {
        if (tpg <= 1) return launch_rnn_rows4_nt<1, 4>(ih, whh_frag, hall, flags, B, T, ntiles, groups, hb, gd, s, gate, etag, num_cus);
        if (tpg == 2) return launch_rnn_rows4_nt<2, 4>(ih, whh_frag, hall, flags, B, T, ntiles, groups, hb, gd, s, gate, etag, num_cus);
        return launch_rnn_rows4_nt<kQ4Tiles, 4>(ih, whh_frag, hall, flags, B, T, ntiles, groups, hb, gd, s, gate, etag, num_cus);
    }
    if (tpg <= 1) return launch_rnn_rows4_nt<1, 8>(ih, whh_frag, hall, flags, B, T, ntiles, groups, hb, gd, s, gate, etag, num_cus);
    if (tpg == 2) return launch_rnn_rows4_nt<2, 8>(ih, whh_frag, hall, flags, B, T, ntiles, groups, hb, gd, s, gate, etag, num_cus);
    return launch_rnn_rows4_nt<kQ4Tiles, 8>(ih, whh_frag, hall, flags, B, T, ntiles, groups, hb, gd, s, gate, etag, num_cus);
}

// one 16-window tile per `cluster` workgroups on the streaming kernel (rnn_kernel): any rnn_hidden that is a multiple of 64
template <bool BWD>
static hipError_t launch_rnn_stream(const float* ih, const float* whh_frag, float* hall, unsigned* flags, int B, int T, int R, int cluster,
                                    int num_cus, const Guard& gd, const float* gate, hipStream_t s) {
    const int ntiles = (B + kRnnTile - 1) / kRnnTile;
    if (cluster < 1) cluster = 1;
    if (cluster > 8) cluster = 8;
    const int KB = R / 16;
    while (cluster > 1 && (KB % (4 * cluster))) cluster >>= 1;
    if (KB % (4 * cluster)) return hipErrorInvalidValue;
    const int nbw = KB / (4 * cluster);
    int groups = ntiles;
    if (cluster > 1) {
        // every workgroup of a cluster must be co-resident: keep the grid within one workgroup per CU
        const int maxg = num_cus / cluster > 0 ? num_cus / cluster : 1;
        if (groups > maxg) groups = maxg;
        hipError_t e = hipMemsetAsync(flags, 0, (size_t)ntiles * T * sizeof(unsigned), s);
        if (e != hipSuccess) return e;
    }
    const size_t smem = (size_t)kRnnTile * (R + 4) * sizeof(float);
    const dim3 grid(groups * cluster), block(256);
#define TIP_RNN_CASE(N) \
    case N: hipLaunchKernelGGL((rnn_kernel<N, BWD>), grid, block, smem, s, ih, whh_frag, hall, flags, B, T, R, cluster, ntiles, gd, gate); break;
    switch (nbw) {
        TIP_RNN_CASE(1) TIP_RNN_CASE(2) TIP_RNN_CASE(3) TIP_RNN_CASE(4) TIP_RNN_CASE(5) TIP_RNN_CASE(6)
        TIP_RNN_CASE(7) TIP_RNN_CASE(8)
        default: return hipErrorInvalidValue;
    }
#undef TIP_RNN_CASE
    return hipGetLastError();
}

// backward recurrence of the training step: R = 512 on the register-resident clusters (cluster 4 / 8 / 16 / four-row tiles; see
// rnn_resident_kernel<.., BWD>), any other rnn_hidden on the streaming kernel (cluster 1 / 2 / 4 / 8)
hipError_t launch_rnn_bwd(const Dims& d, const float* dH, const float* whh_t_frag, const float* h_fwd, float* delta,
                          unsigned* flags, int B, int T, int cluster, int num_cus, const Guard& gd, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (d.R != 512 || cluster < 4)
        return launch_rnn_stream<true>(dH, whh_t_frag, delta, flags, B, T, d.R, cluster == kRnnRows4 ? 1 : cluster, num_cus, gd, h_fwd, s);
    if ((long long)B * T * 512 * 4 > 0x7fffffffLL) return hipErrorInvalidValue;
    const int ntiles = (B + kRnnTile - 1) / kRnnTile;
    if (cluster == kRnnRows4) {
        if (rnn_handoff_mode() == 1) return launch_rnn_rows4(dH, whh_t_frag, delta, flags, B, T, num_cus, false, gd, s, h_fwd);
        cluster = 16;
    }
    if (cluster >= 16) return launch_rnn_resident<8, 4>(dH, whh_t_frag, delta, flags, B, T, ntiles, num_cus, false, gd, s, h_fwd);
    if (cluster == 8) return launch_rnn_resident<4, 1>(dH, whh_t_frag, delta, flags, B, T, ntiles, num_cus, false, gd, s, h_fwd);
    return launch_rnn_resident<8, 1>(dH, whh_t_frag, delta, flags, B, T, ntiles, num_cus, false, gd, s, h_fwd);
}

hipError_t launch_rnn(const Dims& d, const float* ih, const float* whh_frag, float* hall, unsigned* flags, int B,
                      int T, int cluster, int num_cus, bool hall_armed, const Guard& gd, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    const int R = d.R;
    const int ntiles = (B + kRnnTile - 1) / kRnnTile;
    if (cluster < 1) cluster = 1;
    if (R == 512 && (long long)B * T * 512 * 4 <= 0x7fffffffLL) {
        if (cluster == kRnnRows4) {
            if (rnn_handoff_mode() == 1) return launch_rnn_rows4(ih, whh_frag, hall, flags, B, T, num_cus, hall_armed, gd, s, nullptr);
            cluster = 16;   // (TIP_RNN_HANDOFF=0, measurement: the counter protocol exists for the 16-row kernels only)
        }
        // register-resident clustered kernel: W_hh slice lives in VGPRs, 4/8/16 workgroups per window tile
        if (cluster >= 32)   // two 4-wave workgroups per CU, each one column block of a tile: one computes while the other waits
            return launch_rnn_resident<4, 4>(ih, whh_frag, hall, flags, B, T, ntiles, num_cus, hall_armed, gd, s, nullptr, 2);
        if (cluster >= 16) {
            static int w8 = -1;   // TIP_RNN_C16=4 selects the 4-wave variant (measurement)
            if (w8 < 0) w8 = (tip_env("TIP_RNN_C16") && tip_env("TIP_RNN_C16")[0] == '4') ? 0 : 1;
            return w8 ? launch_rnn_resident<8, 4>(ih, whh_frag, hall, flags, B, T, ntiles, num_cus, hall_armed, gd, s)
                      : launch_rnn_resident<4, 2>(ih, whh_frag, hall, flags, B, T, ntiles, num_cus, hall_armed, gd, s);
        }
        if (cluster == 8) return launch_rnn_resident<4, 1>(ih, whh_frag, hall, flags, B, T, ntiles, num_cus, hall_armed, gd, s);
        if (cluster == 4) return launch_rnn_resident<8, 1>(ih, whh_frag, hall, flags, B, T, ntiles, num_cus, hall_armed, gd, s);
    }
    return launch_rnn_stream<false>(ih, whh_frag, hall, flags, B, T, R, cluster, num_cus, gd, nullptr, s);
}

}  // namespace tip

namespace tip {
hipError_t read_spin_timeouts_general(unsigned* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_spin_timeouts_general), sizeof(unsigned));
}
}  // namespace tip

extern "C" int tip_debug_read_rnn_trace(unsigned long long* out, int n) {
    if (!out || n < 0 || n > 2048) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tip::g_rnn_trace), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -5;
}

// measurement only: one lane writes {s_memtime (shader-clock ticks), wall_clock64 (constant 100 MHz), XCC id}.  Two probes
// bracketing a stretch of work on the same stream give the shader clock actually held over it (bench.py, sustained pass).
namespace tip {
__global__ void clock_probe_kernel(unsigned long long* out) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    out[0] = __builtin_amdgcn_s_memtime();
    out[1] = wall_clock64();
    out[2] = xcc & 0xf;
}
}  // namespace tip

extern "C" int tip_debug_clock_probe(unsigned long long* dev_out, void* stream) {
    if (!dev_out) return -1;
    hipLaunchKernelGGL(tip::clock_probe_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), dev_out);
    return hipGetLastError() == hipSuccess ? 0 : -5;
}

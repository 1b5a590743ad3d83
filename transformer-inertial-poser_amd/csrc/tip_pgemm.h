// tip_pgemm.h — register-ring / fragment-order GEMM building blocks shared by the two-window fused kernel
// (tip_fused2.hip), the general plan's panel GEMM and the training step's GEMMs (tip_train.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace tip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ldfrag2(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
    return make_float4(f.x, f.y, f.z, f.w);
}

template <int NBW>
struct WRing2 {
    float4 w0[NBW], w1[NBW];
};

template <int NBW>
__device__ __forceinline__ void ring2_prefetch(WRing2<NBW>& g, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff, int nstride_b) {
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
        g.w0[n] = ldfrag2(rsrc, voff, soff + n * nstride_b);
        g.w1[n] = ldfrag2(rsrc, voff, soff + n * nstride_b + 1024);
    }
}

// SWAP: operands exchanged — the instruction then delivers the TRANSPOSED tile (lane (l15, lg) holds row l15, columns 4 lg .. 4 lg + 3
// instead of column l15, rows 4 lg .. + 3), bit for bit the same sums: an epilogue can then move 16 bytes per lane.
template <int NRB, int NBW, bool SWAP = false>
__device__ __forceinline__ void mfma_block2(f32x4 (&acc)[NRB][NBW], const float4 (&a)[NRB], const float4 (&w)[NBW]) {
#define TIP_MFMA2_STEP(c)                                                                                               \
    _Pragma("unroll") for (int r = 0; r < NRB; ++r) _Pragma("unroll") for (int n = 0; n < NBW; ++n)                      \
        acc[r][n] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x4f32(w[n].c, a[r].c, acc[r][n], 0, 0, 0)                     \
                         : __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].c, w[n].c, acc[r][n], 0, 0, 0);
    TIP_MFMA2_STEP(x)
    TIP_MFMA2_STEP(y)
    TIP_MFMA2_STEP(z)
    TIP_MFMA2_STEP(w)
#undef TIP_MFMA2_STEP
}

template <int NRB, int NBW>
__device__ __forceinline__ void zero_acc2(f32x4 (&acc)[NRB][NBW]) {
#pragma unroll
    for (int r = 0; r < NRB; ++r)
#pragma unroll
        for (int n = 0; n < NBW; ++n) acc[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// =====================================================================================================================
// pgemm: the fused kernels' GEMM structure as a stand-alone kernel for BIG linears (the scaled configuration's d = 1024,
// ffn = 4096): C[M,N] = epi(A[M,K] W^T + bias (+ res)).  An 80-row x 512-column panel per 512-thread workgroup; the A
// panel goes through LDS in K-chunks of 128 (double-buffered, ONE barrier per chunk = per 640 MFMAs of a wave), the
// weights never touch LDS: they are packed in 16x16x4 B-fragment order and each wave streams the fragments of its four
// column blocks from L2 into a register ring.  Against the LDS-tiled tgemm16 (both operands staged, a barrier every 32
// MFMAs, 16 flop per L2 byte) this reads 40 flop per L2 byte and keeps the matrix pipe fed between barriers.
// =====================================================================================================================
namespace pg {
constexpr int ROWS = 80, RB = 5, NBW = 4, KC = 128, LDA = KC + 8, THREADS = 512, COLS = 8 * NBW * 16;   // 512
constexpr int LDS_BYTES = 2 * ROWS * LDA * 4;                                                          // 87 040
}  // namespace pg

// Epi finishes and stores the outputs in groups of FOUR consecutive columns of one row (the tiles are computed transposed):
//   Col  begin(col)                       what a column group needs whatever the row: the bias (loaded once per group, not per element)
//   Aux  load(row, col)                   the group's other inputs (residual, gate): requested for all five row blocks before the first
//                                         is used
//   void store(row, col, acc, Col, Aux)   16-byte store
// (round 3: the first form, `epi(row, col, acc)` per ELEMENT with its bias / residual loads inside, compiled to 80 dependent
// load -> wait -> store round trips per lane — the optimiser may not move a load across a store it cannot prove disjoint — and cost
// 13-28 us of every 80 x 512 panel: the big linears of the scaled configuration ran at 71-87 % of the matrix peak because of it.)
template <typename Epi>
__device__ __forceinline__ void pgemm_body(const float* __restrict__ A, int lda, const float* __restrict__ wfrag, int wbytes, int M,
                                           int N, int K, const Epi& epi) {
    using namespace pg;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wfrag), 0, wbytes, 0x00020000);
    const int voff = lane * 16;
    const int row0 = blockIdx.y * ROWS;
    const int nb0 = blockIdx.x * (COLS / 16) + wave * NBW;          // first 16-column block of this wave
    const int KB = K >> 4, nchunks = K / KC;
    const int wsoff = nb0 * KB * 1024;                              // byte offset of (nb0, kb = 0); next nb: + KB*1024
    // A staging: 80 x 128 floats = 2560 float4 = 5 per thread
    float4 st[5];
    auto fetch = [&](int c) {
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int f = tid + u * THREADS, r = f >> 5, k4 = f & 31;
            st[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row0 + r < M) st[u] = *reinterpret_cast<const float4*>(A + (size_t)(row0 + r) * lda + c * KC + k4 * 4);
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int f = tid + u * THREADS, r = f >> 5, k4 = f & 31;
            *reinterpret_cast<float4*>(smem + (buf * ROWS + r) * LDA + k4 * 4) = st[u];
        }
    };
    f32x4 acc[RB][NBW];
    zero_acc2<RB, NBW>(acc);
    WRing2<NBW> g;
    ring2_prefetch<NBW>(g, rsrc, voff, wsoff, KB * 1024);
    fetch(0);
    stage(0);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) fetch(c + 1);
        const float* As = smem + ((c & 1) * ROWS + l15) * LDA + lg * 4;
        float4 a0[RB], a1[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) a0[r] = *reinterpret_cast<const float4*>(As + r * 16 * LDA);
#pragma unroll 1
        for (int kb = 0; kb < KC / 16; kb += 2) {
            const int gkb = c * (KC / 16) + kb + 2;                 // the k-blocks the ring fetches next (may run past K: padded image)
#pragma unroll
            for (int r = 0; r < RB; ++r) a1[r] = *reinterpret_cast<const float4*>(As + r * 16 * LDA + (kb + 1) * 16);
            mfma_block2<RB, NBW, true>(acc, a0, g.w0);
#pragma unroll
            for (int n = 0; n < NBW; ++n) g.w0[n] = ldfrag2(rsrc, voff, wsoff + (n * KB + gkb) * 1024);
            __builtin_amdgcn_sched_barrier(0);   // keep each refill behind the block that freed its registers (not next to its use)
#pragma unroll
            for (int r = 0; r < RB; ++r) a0[r] = *reinterpret_cast<const float4*>(As + r * 16 * LDA + ((kb + 2) & (KC / 16 - 1)) * 16);
            __builtin_amdgcn_sched_barrier(0);   // the next pair's A fragments stay here, half a trip ahead of their first MFMA (gemm_phase_h)
            mfma_block2<RB, NBW, true>(acc, a1, g.w1);
#pragma unroll
            for (int n = 0; n < NBW; ++n) g.w1[n] = ldfrag2(rsrc, voff, wsoff + (n * KB + gkb + 1) * 1024);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c + 1 < nchunks) stage((c + 1) & 1);
        __syncthreads();
    }
    // epilogue.  Transposed tiles: lane (l15, lg) holds row l15, columns 4 lg .. 4 lg + 3 of its 16 x 16 block.
    // (N is a multiple of the panel width: every column group exists.)  The inputs of group n + 1 are requested before group n is
    // stored — different columns, so this order is right even for an in-place residual — and a group's five row blocks together.
    typename Epi::Aux aux[2][RB];
    auto request = [&](int n, typename Epi::Aux (&a)[RB]) {
        const int col = (nb0 + n) * 16 + lg * 4;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int row = row0 + r * 16 + l15;
            a[r] = epi.load(row < M ? row : M - 1, col);   // clamped, unconditional
        }
    };
    request(0, aux[0]);
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
        const int col = (nb0 + n) * 16 + lg * 4;
        const typename Epi::Col cb = epi.begin(col);
        if (n + 1 < NBW) request(n + 1, aux[(n + 1) & 1]);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int row = row0 + r * 16 + l15;
            if (row < M) epi.store(row, col, acc[r][n], cb, aux[n & 1][r]);
        }
    }
}

inline bool pgemm_ok(int M, int N, int K) { return N % pg::COLS == 0 && K % pg::KC == 0 && M >= 4 * pg::ROWS; }

}  // namespace tip

// tip_fused.hip — fused execution plan for the paper configuration (d=256, 16 heads x 16, ffn=1024, T<=40).
//
// One 512-thread workgroup (8 waves, 2 per SIMD) carries ONE 40-frame window through the prologue, in_linear
// and all encoder layers without leaving the CU: the activations [48 x 256] live in LDS for the whole pass,
// every GEMM is v_mfma_f32_16x16x4_f32 with the A operand read from LDS (ds_read_b128 = 4 k-steps) and the B
// operand streamed straight from L2 in pre-packed fragment order (one coalesced 1-KiB global_load_dwordx4 per
// 16x16 weight block per wave, register double-buffered), attention runs one head per wave between the QKV
// GEMM and the out-projection with no inter-wave dependency, LayerNorm uses wave shuffles.  HBM sees only the
// window's inputs (35 KB) and the encoder output (40 KB); weights (12.6 MB) are L2/Infinity-Cache resident and
// shared by all 256 CUs.  Reference: /root/reference/simple_transformer_with_state.py:63-91.
//
// LDS map (floats):  X [48][260] residual stream | C chunk region: Q [48][132] | K [48][132] | V^T [128][52] of 8
//                    heads, reused as U [48][228] (prologue) and Hc [48][260] (FFN hidden chunk).  127,232 B.
#include <stdlib.h>
#include <string.h>

#include "tip_internal.h"
#include "tip_attention.h"
#include "tip_layernorm.h"

namespace tip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace fz {
constexpr int D = 256, H = 16, DH = 16, F = 1024, RP = 48, RB = 3, TMAX = 40, R = 512;
constexpr int KIN = 224;            // in_linear K, zero padded (221 with acc-sum, 203 without)
// Row strides of the A-operand planes are = 8 (mod 64) dwords, not the usual "+ 4".  An A fragment is one ds_read_b128 per lane: row
// l15, dwords 4 lg .. 4 lg + 3 of the k-block.  The LDS serves a b128 read in four groups of 16 lanes that are NOT contiguous —
// {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS) — i.e. rows {0-3, 12-15} at one k offset together with rows
// {4-11} at the NEXT k offset.  In units of 16-byte slots (16 per LDS line) lane (l15, lg) hits slot (s * l15 + lg) mod 16, s = stride / 4
// mod 16: with s = 1 (stride 260) the two row sets land on {0-3, 12-15} and {5-12} — slot 12 twice, every read 5 LDS cycles instead of
// 4; with s = 2 (stride 264) on the even and the odd slots.  SQ_LDS_BANK_CONFLICT of the hybrid encoder: 26.4 M -> 3.6 M cycles per
// launch (the tail fragments' four rows x two k offsets likewise: {0-3} and {1-4} -> {0, 2, 4, 6} and {1, 3, 5, 7}).
constexpr int LDX = D + 8;          // 264
constexpr int LDC = 128 + 4;        // 132: one Q / K plane of an 8-head chunk, [48 rows][128 channels]
constexpr int LDV = RP + 4;         // 52: V is kept TRANSPOSED, [128 channels][48 keys], so P.V reads B fragments as b128
constexpr int LDU = KIN + 8;        // 232
constexpr int X_FLOATS = RP * LDX;              // 12480
constexpr int C_FLOATS = 2 * RP * LDC + 128 * LDV;   // 19328
constexpr int LDS_BYTES = (X_FLOATS + C_FLOATS) * 4;
constexpr int THREADS = 512;
// packed section (floats)
constexpr size_t IN_W = 0;
constexpr size_t IN_B = IN_W + (size_t)D * KIN;
constexpr size_t LAYER0 = IN_B + D;
constexpr size_t QKV_W = 0;
constexpr size_t QKV_B = QKV_W + (size_t)3 * D * D;
constexpr size_t WO_W = QKV_B + 3 * D;
constexpr size_t WO_B = WO_W + (size_t)D * D;
constexpr size_t W1_W = WO_B + D;
constexpr size_t W1_B = W1_W + (size_t)F * D;
constexpr size_t W2_W = W1_B + F;
constexpr size_t W2_B = W2_W + (size_t)D * F;
constexpr size_t G1 = W2_B + D;
constexpr size_t BE1 = G1 + D;
constexpr size_t G2 = BE1 + D;
constexpr size_t BE2 = G2 + D;
constexpr size_t LAYER_FLOATS = BE2 + D;
constexpr size_t TAIL_PAD = 4096;      // the k-loop prefetches up to 2 blocks (2 KiB) past a wave's last block
}  // namespace fz

bool fused_supported(const Dims& d, int T) {
    return d.D == fz::D && d.H == fz::H && d.F == fz::F && d.In <= fz::KIN && T >= 1 && T <= fz::TMAX && d.L >= 1;
}

// RNN input projection (R = 512) rides at the end of the fused section: [32 nb][16 kb] fragments + (b_ih + b_hh)
bool fused_has_rnn_ih(const Dims& d) { return d.with_rnn && d.R == fz::R; }
static size_t fused_ih_off(const Dims& d) { return fz::LAYER0 + (size_t)d.L * fz::LAYER_FLOATS; }

size_t fused_packed_floats(const Dims& d) {
    if (!(d.D == fz::D && d.H == fz::H && d.F == fz::F && d.In <= fz::KIN)) return 0;
    return fused_ih_off(d) + (fused_has_rnn_ih(d) ? (size_t)fz::R * fz::D + fz::R : 0) + fz::TAIL_PAD;
}

// W [N][K] row-major (K <= Kpad) -> 16x16x4 B-fragment order [N/16][Kpad/16][64 lanes][4]:
//   dst[((nb*KB + kb)*64 + lane)*4 + s] = W[nb*16 + (lane&15)][kb*16 + 4*(lane>>4) + s]
static void pack_frag(float* dst, const float* W, int N, int K, int Kpad, const int* row_map, float scale_rows_lt,
                      int scale_n) {
    const int KB = Kpad / 16;
    for (int nb = 0; nb < N / 16; ++nb)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < 4; ++s) {
                    const int n = nb * 16 + (lane & 15), k = kb * 16 + 4 * (lane >> 4) + s;
                    const int src = row_map ? row_map[n] : n;
                    float v = k < K ? W[(size_t)src * K + k] : 0.f;
                    if (n < scale_n) v *= scale_rows_lt;
                    dst[((size_t)(nb * KB + kb) * 64 + lane) * 4 + s] = v;
                }
}

void fused_pack(const Dims& d, const float* const* t, float* dst) {
    using namespace fz;
    // in_linear: channel shuffle (:88-89) folded into rows, root-velocity columns (:75) zeroed
    std::vector<int> map(D);
    for (int a = 0; a < DH; ++a)
        for (int b = 0; b < H; ++b) map[a * H + b] = b * DH + a;
    std::vector<float> win((size_t)D * d.In);
    memcpy(win.data(), t[0], sizeof(float) * win.size());
    for (int n = 0; n < D; ++n)
        for (int c = d.rootv0; c < d.rootv1; ++c) win[(size_t)n * d.In + d.n_imu_total + c] = 0.f;
    pack_frag(dst + IN_W, win.data(), D, d.In, KIN, map.data(), 1.f, 0);
    for (int n = 0; n < D; ++n) dst[IN_B + n] = t[1][map[n]];
    for (int l = 0; l < d.L; ++l) {
        const float* const* lw = t + 2 + 12 * l;
        float* L = dst + LAYER0 + (size_t)l * LAYER_FLOATS;
        pack_frag(L + QKV_W, lw[0], 3 * D, D, D, nullptr, 0.25f, D);  // 1/sqrt(16) folded into W_q (exact)
        for (int i = 0; i < 3 * D; ++i) L[QKV_B + i] = lw[1][i] * (i < D ? 0.25f : 1.f);
        pack_frag(L + WO_W, lw[2], D, D, D, nullptr, 1.f, 0);
        memcpy(L + WO_B, lw[3], sizeof(float) * D);
        pack_frag(L + W1_W, lw[4], F, D, D, nullptr, 1.f, 0);
        memcpy(L + W1_B, lw[5], sizeof(float) * F);
        pack_frag(L + W2_W, lw[6], D, F, F, nullptr, 1.f, 0);
        memcpy(L + W2_B, lw[7], sizeof(float) * D);
        memcpy(L + G1, lw[8], sizeof(float) * D);
        memcpy(L + BE1, lw[9], sizeof(float) * D);
        memcpy(L + G2, lw[10], sizeof(float) * D);
        memcpy(L + BE2, lw[11], sizeof(float) * D);
    }
    if (fused_has_rnn_ih(d)) {
        const float* const* tw = t + 2 + 12 * d.L;
        float* I = dst + fused_ih_off(d);
        pack_frag(I, tw[0], R, D, D, nullptr, 1.f, 0);
        for (int i = 0; i < R; ++i) I[(size_t)R * D + i] = tw[2][i] + tw[3][i];
    }
}

static PackOp mk_op(const float* src, size_t dst, int N, int K, int src_rows, int src_cols, int frag) {
    PackOp o;
    o.src = src; o.src2 = nullptr; o.dst_off = dst; o.N = N; o.K = K; o.src_rows = src_rows; o.src_cols = src_cols; o.frag = frag;
    o.shuffle_h = 0; o.shuffle_dh = 0; o.z0 = 0; o.z1 = 0; o.scale_rows = 0; o.scale = 1.f; o.transpose = 0;
    return o;
}

// the same image as fused_pack() above, as descriptors for the on-device packer
void fused_pack_ops(const Dims& d, const float* const* t, size_t base, std::vector<PackOp>& ops) {
    using namespace fz;
    {
        PackOp o = mk_op(t[0], base + IN_W, D, KIN, D, d.In, 1);
        o.shuffle_h = H; o.shuffle_dh = DH;
        o.z0 = d.n_imu_total + d.rootv0; o.z1 = d.n_imu_total + d.rootv1;
        ops.push_back(o);
        PackOp b = mk_op(t[1], base + IN_B, D, 1, D, 1, 0);
        b.shuffle_h = H; b.shuffle_dh = DH;
        ops.push_back(b);
    }
    for (int l = 0; l < d.L; ++l) {
        const float* const* lw = t + 2 + 12 * l;
        const size_t L = base + LAYER0 + (size_t)l * LAYER_FLOATS;
        PackOp q = mk_op(lw[0], L + QKV_W, 3 * D, D, 3 * D, D, 1);
        q.scale = 0.25f; q.scale_rows = D;
        ops.push_back(q);
        PackOp qb = mk_op(lw[1], L + QKV_B, 3 * D, 1, 3 * D, 1, 0);
        qb.scale = 0.25f; qb.scale_rows = D;
        ops.push_back(qb);
        ops.push_back(mk_op(lw[2], L + WO_W, D, D, D, D, 1));
        ops.push_back(mk_op(lw[3], L + WO_B, D, 1, D, 1, 0));
        ops.push_back(mk_op(lw[4], L + W1_W, F, D, F, D, 1));
        ops.push_back(mk_op(lw[5], L + W1_B, F, 1, F, 1, 0));
        ops.push_back(mk_op(lw[6], L + W2_W, D, F, D, F, 1));
        ops.push_back(mk_op(lw[7], L + W2_B, D, 1, D, 1, 0));
        ops.push_back(mk_op(lw[8], L + G1, D, 1, D, 1, 0));
        ops.push_back(mk_op(lw[9], L + BE1, D, 1, D, 1, 0));
        ops.push_back(mk_op(lw[10], L + G2, D, 1, D, 1, 0));
        ops.push_back(mk_op(lw[11], L + BE2, D, 1, D, 1, 0));
    }
    if (fused_has_rnn_ih(d)) {
        const float* const* tw = t + 2 + 12 * d.L;
        const size_t I = base + fused_ih_off(d);
        ops.push_back(mk_op(tw[0], I, R, D, R, D, 1));
        PackOp b = mk_op(tw[2], I + (size_t)R * D, R, 1, R, 1, 0);
        b.src2 = tw[3];
        ops.push_back(b);
    }
}

// ------------------------------------------------------------------------------------------------------------
// device pieces
// ------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wsum(float v) { return wave64_sum(v); }   // DPP + permlane swaps, no LDS round trips

// acc[r][n] += A(rows of block r) x W(block n).  For the first SWAPN column blocks the MFMA operands are swapped (weights
// as A, activations as B): the accumulator then holds the TRANSPOSED tile — (channels 4*lg + e, row l15) — which is the
// fragment layout a following MFMA wants for Q and K (tip_attention.h, attention_head_regs).
template <int NBW, int SWAPN = 0>
__device__ __forceinline__ void mfma_block(f32x4 (&acc)[fz::RB][NBW], const float4 (&a)[fz::RB], const float4 (&w)[NBW]) {
#define TIP_MFMA_STEP(c)                                                                                             \
    _Pragma("unroll") for (int r = 0; r < fz::RB; ++r) _Pragma("unroll") for (int n = 0; n < NBW; ++n)                \
        acc[r][n] = n < SWAPN ? __builtin_amdgcn_mfma_f32_16x16x4f32(w[n].c, a[r].c, acc[r][n], 0, 0, 0)              \
                              : __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].c, w[n].c, acc[r][n], 0, 0, 0);
    TIP_MFMA_STEP(x)
    TIP_MFMA_STEP(y)
    TIP_MFMA_STEP(z)
    TIP_MFMA_STEP(w)
#undef TIP_MFMA_STEP
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// One 1-KiB fragment block: lane l gets bytes [16 l, 16 l + 16) of the block at byte offset `soff`.
// raw buffer loads (SGPR descriptor + scalar offset) instead of flat loads: the optimiser treats them as opaque
// calls, so the software-pipelined prefetch below survives (flat loads from __restrict__ memory were re-sunk to
// their use, exposing the L2 latency every k-step), and addresses cost one VGPR.
__device__ __forceinline__ float4 load_frag(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    // NB: bit-cast the whole vector — __builtin_bit_cast(float, v[i]) on a vector ELEMENT is miscompiled by this
    // clang (every element reads lane 0 of the vector).
    const f32x4 f = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
    return make_float4(f.x, f.y, f.z, f.w);
}

// Register ring of weight fragments: k-blocks (even, odd) of the NBW column blocks a wave owns.
template <int NBW>
struct WRing {
    float4 w0[NBW], w1[NBW];
};

// Prime a ring with k-blocks 0 and 1 of a phase.  Called EARLY (before the epilogue / barrier / LayerNorm /
// attention that precedes the phase) so the L2 latency of a phase's first fragments is never exposed.
template <int NBW>
__device__ __forceinline__ void ring_prefetch(WRing<NBW>& g, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff, int nstride_b) {
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
        g.w0[n] = load_frag(rsrc, voff, soff + n * nstride_b);
        g.w1[n] = load_frag(rsrc, voff, soff + n * nstride_b + 1024);
    }
}

// acc[r][n] += A[48 x 16*KB] (LDS, leading dim lda) * Wblock(n, kb)   for kb in [0, KB)
//   soff: byte offset (wave-uniform) of the wave's first block at kb = 0; block n is + n*nstride_b; k-block kb is + kb*1024.
//   The ring holds k-blocks 0,1 on entry.  The prefetch runs two k-blocks ahead and is unconditional (branch-free
//   loop => counted vmcnt waits); for the LAST pair it is redirected to k-blocks 0,1 of the NEXT phase
//   (nsoff / nnstride_b), so on exit the ring is already primed for a following phase of the same width.
//   Callers without such a successor pass their own soff (a harmless in-bounds reload).
template <int NBW, int KB, bool NOMMA = false, int SWAPN = 0, bool PIN = true>
__device__ __forceinline__ void gemm_phase(f32x4 (&acc)[fz::RB][NBW], const float* As, int lda, __amdgpu_buffer_rsrc_t rsrc,
                                           int voff, int soff, int nstride_b, WRing<NBW>& g, int nsoff, int nnstride_b) {
    static_assert(KB % 2 == 0, "k-blocks are processed in pairs");
    // A fragments are double-buffered too: the ds_read_b128 of k-block kb+1 is in flight while the MFMAs of kb issue.
    // (Reads past the last k-block stay inside the LDS allocation and are never used.)
    float4 a0[fz::RB], a1[fz::RB];
#pragma unroll
    for (int r = 0; r < fz::RB; ++r) a0[r] = *reinterpret_cast<const float4*>(As + r * 16 * lda);
#pragma unroll 1
    for (int kb = 0; kb < KB; kb += 2) {
        const bool last = kb + 2 >= KB;                       // wave-uniform: scalar selects, no branch
        const int o = last ? nsoff : soff + (kb + 2) * 1024;
        const int st = last ? nnstride_b : nstride_b;
#pragma unroll
        for (int r = 0; r < fz::RB; ++r) a1[r] = *reinterpret_cast<const float4*>(As + r * 16 * lda + (kb + 1) * 16);
        if (!NOMMA) mfma_block<NBW, SWAPN>(acc, a0, g.w0);
        else { asm volatile("" :: "v"(a0[0].x), "v"(g.w0[0].x)); }
#pragma unroll
        for (int n = 0; n < NBW; ++n) g.w0[n] = load_frag(rsrc, voff, o + n * st);
        // PIN: keep the refill here (the scheduler may sink it towards its use).  Measured: helps the training forward and the
        // backward kernels by ~1 %, costs the stash-free inference kernel 0.7 % — which therefore opts out
        if (PIN) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < fz::RB; ++r) a0[r] = *reinterpret_cast<const float4*>(As + r * 16 * lda + (kb + 2) * 16);
        if (!NOMMA) mfma_block<NBW, SWAPN>(acc, a1, g.w1);
        else { asm volatile("" :: "v"(a1[0].x), "v"(g.w1[0].x)); }
#pragma unroll
        for (int n = 0; n < NBW; ++n) g.w1[n] = load_frag(rsrc, voff, o + n * st + 1024);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NBW>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[fz::RB][NBW]) {
#pragma unroll
    for (int r = 0; r < fz::RB; ++r)
#pragma unroll
        for (int n = 0; n < NBW; ++n) acc[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// LayerNorm of the residual stream: tip_layernorm.h (16 lanes per row, DPP reductions), all 48 rows here — the pad rows of the
// padded kernels evolve like real ones and must stay finite.
// The training forward's stash stores have per-thread row / column offsets that are loop invariant; hoisted out of the layer
// loop they cost ~50 VGPRs for the whole kernel (hundreds of bytes of scratch per lane).  An opaque copy of the thread index at
// each use keeps them where they are consumed: a dozen VALU instructions per store loop.
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

// rows 0..T-1 of an LDS tile [rows][ld] (cols floats wide) -> HBM rows of stride dst_ld, 16-byte stores
__device__ __forceinline__ void rows_to_hbm(const float* lds, int ld, int cols, float* dst, int dst_ld, int T, int tid) {
    const int c4n = cols >> 2;
    for (int i = tid; i < T * c4n; i += fz::THREADS) {
        const int r = i / c4n, c4 = i - r * c4n;
        // streaming stores: the stash (2 MB per window) is read next by the backward, the weights it would push out of L2 in 0.1 us
        __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(lds + r * ld + c4 * 4), reinterpret_cast<f32x4*>(dst + (size_t)r * dst_ld + c4 * 4));
    }
}

// The same with a FIXED number of stores per thread (T <= 40: the hybrid kernel), rows >= T dropped by the buffer's range check instead
// of by the loop bound.  With the run-time trip count above the compiler cannot count the stores in flight, and the `s_waitcnt vmcnt(N)`
// of the weight ring behind such a loop (vector memory retires in order, stores included) degenerates to "all but the last few":
// every stash point waited for its own stores' acknowledgement before the next phase's first MFMA.
template <int COLS>
__device__ __forceinline__ void rows_to_hbm_fixed(const float* lds, int ld, float* dst, int dst_ld, int T, int tid) {
    constexpr int C4 = COLS / 4, NP = (40 * C4 + fz::THREADS - 1) / fz::THREADS;
    static_assert((NP * fz::THREADS + C4 - 1) / C4 <= fz::RP, "the passes stay inside the LDS tile's rows");
    const __amdgpu_buffer_rsrc_t rs = tip_rows_buffer(dst, T * dst_ld * 4);
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int i = tid + j * fz::THREADS, r = i / C4, c4 = i - r * C4;
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds + r * ld + c4 * 4);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ln_u32x4, v), rs, (r * dst_ld + c4 * 4) * 4, 0, kTipNT);
    }
}

// The training forward's hidden chunk [T][256] (after ReLU and dropout): rows to HBM as rows_to_hbm does, plus the chunk's gates
// (hidden > 0) as BITS for the fused FFN backward.  gb points at row 0 of the window in the gate-bit array [M][32 dwords]:
//   dword (row, f*8 + 2*e + (c4 >> 5)), bit (c4 & 31)  =  hidden[row][f*256 + 4*c4 + e] > 0      (c4 = 0..63: column quad)
// One wave iteration is one row (64 lanes = its 64 column quads): four ballots, 32 bytes stored by lane 0.  Round 5: ffn_bwd_kernel
// read the gates as the saved fp32 hidden rows — 42 MB per layer that arrive cold from HBM in front of each chunk's weight ring
// (vector memory returns in order: every chunk stalled ~4 000 cycles behind them); as bits they are 1.3 MB, staged into LDS once per window.
template <bool FIXED>   // FIXED: five stores per thread whatever T <= 40 is (rows_to_hbm_fixed), lanes other than 0 drop the bit stores
__device__ __forceinline__ void hidden_to_hbm(const float* lds, int ld, float* dst, int dst_ld, unsigned* gb, int f, int T, int tid) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int lane = tid & 63;
    if (FIXED) {
        const __amdgpu_buffer_rsrc_t rs = tip_rows_buffer(dst, T * dst_ld * 4);
        const __amdgpu_buffer_rsrc_t gs = tip_rows_buffer(reinterpret_cast<const float*>(gb), T * 128);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int r = (tid >> 6) + 8 * j;
            const f32x4 v = *reinterpret_cast<const f32x4*>(lds + r * ld + lane * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ln_u32x4, v), rs, (r * dst_ld + lane * 4) * 4, 0, kTipNT);
            const unsigned long long b0 = __builtin_amdgcn_ballot_w64(v[0] > 0.f), b1 = __builtin_amdgcn_ballot_w64(v[1] > 0.f);
            const unsigned long long b2 = __builtin_amdgcn_ballot_w64(v[2] > 0.f), b3 = __builtin_amdgcn_ballot_w64(v[3] > 0.f);
            const int off = lane == 0 ? r * 128 + f * 32 : 0x7fffff00;      // one lane writes the row's 32 bytes; the others fall outside the buffer
            const u32x4 lo = {(unsigned)b0, (unsigned)(b0 >> 32), (unsigned)b1, (unsigned)(b1 >> 32)};
            const u32x4 hi = {(unsigned)b2, (unsigned)(b2 >> 32), (unsigned)b3, (unsigned)(b3 >> 32)};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ln_u32x4, lo), gs, off, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ln_u32x4, hi), gs, off, 16, 0);
        }
        return;
    }
    for (int i = tid; i < T * 64; i += fz::THREADS) {
        const int r = i >> 6;
        const f32x4 v = *reinterpret_cast<const f32x4*>(lds + r * ld + lane * 4);
        __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst + (size_t)r * dst_ld + lane * 4));
        const unsigned long long b0 = __builtin_amdgcn_ballot_w64(v[0] > 0.f), b1 = __builtin_amdgcn_ballot_w64(v[1] > 0.f);
        const unsigned long long b2 = __builtin_amdgcn_ballot_w64(v[2] > 0.f), b3 = __builtin_amdgcn_ballot_w64(v[3] > 0.f);
        if (lane == 0) {
            u32x4* o = reinterpret_cast<u32x4*>(gb + (size_t)r * 32 + f * 8);
            o[0] = (u32x4){(unsigned)b0, (unsigned)(b0 >> 32), (unsigned)b1, (unsigned)(b1 >> 32)};
            o[1] = (u32x4){(unsigned)b2, (unsigned)(b2 >> 32), (unsigned)b3, (unsigned)(b3 >> 32)};
        }
    }
}

// ABL != 0 builds are MEASUREMENT-ONLY ablations (wrong results): 1 = no attention, 2 = no LayerNorm, 4 = no MFMA,
// 8 = no epilogue LDS traffic.  Selected with TIP_FUSED_ABLATE=<mask>; the product path always runs ABL = 0.
template <int ABL>
__global__ __launch_bounds__(fz::THREADS) void fused_encoder_kernel(
    const float* __restrict__ wts, const float* __restrict__ x_imu, const float* __restrict__ x_s,
    const float* __restrict__ keep_mask, float keep_scale, float* __restrict__ xout, float* __restrict__ ih_out,
    unsigned* __restrict__ hall_sentinel, int B, int T, int NI, int S, int L, int wbytes, int ih_off_b, FusedTrain tr) {
    using namespace fz;
    constexpr bool TR = (ABL & 8) != 0;   // training forward: stash activations, apply the encoder's dropout
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;
    float* C = smem + X_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wts), 0, wbytes, 0x00020000);
    const int voff = lane * 16;

    for (int win = blockIdx.x; win < B; win += gridDim.x) {
        // weight fragments of the first GEMM are requested before the window's inputs are even staged
        const int in_soff = (int)(IN_W * 4) + (wave * 2) * (KIN / 16) * 1024;
        WRing<2> g_in;
        ring_prefetch<2>(g_in, rsrc, voff, in_soff, (KIN / 16) * 1024);
        WRing<3> g_qkv;   // primed one phase ahead of every QKV projection
        // ---- P0 prologue (:63-78): U = [x_imu | scrub(x_s) * mask | 0], rows >= T zero ---------------------------
        float* U = C;
        for (int i = tid; i < RP * LDU; i += THREADS) U[i] = 0.f;
        __syncthreads();
        {
            const float* xi = x_imu + (size_t)win * T * NI;
            for (int i = tid; i < T * NI; i += THREADS) {
                const int r = i / NI, c = i - r * NI;
                U[r * LDU + c] = xi[i];
            }
            const float* xs = x_s + (size_t)win * T * S;
            const float* km = keep_mask ? keep_mask + (size_t)win * T * S : nullptr;
            for (int i = tid; i < T * S; i += THREADS) {
                const int r = i / S, c = i - r * S;
                float v = xs[i];
                if (v != v) v = 0.f;                  // :65
                if (km) v = v * km[i] * keep_scale;   // :77 with an explicit keep-mask
                U[r * LDU + NI + c] = v;
            }
        }
        __syncthreads();
        // ---- P1 in_linear (:79) + channel shuffle (:88-89, folded into the packed rows) ---------------------------
        {
            f32x4 acc[RB][2];
            zero_acc<2>(acc);
            gemm_phase<2, KIN / 16, (ABL & 4) != 0, 0, TR>(acc, U + l15 * LDU + lg * 4, LDU, rsrc, voff, in_soff, (KIN / 16) * 1024, g_in, in_soff,
                                    (KIN / 16) * 1024);
            // layer 0, chunk 0 QKV fragments fly during the epilogue + barrier
            ring_prefetch<3>(g_qkv, rsrc, voff, (int)(LAYER0 * 4) + (int)(QKV_W * 4) + wave * 16 * 1024, 16 * 16 * 1024);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int col = (wave * 2 + n) * 16 + l15;
                const float bv = wts[IN_B + col];
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) X[(r * 16 + lg * 4 + e) * LDX + col] = acc[r][n][e] + bv;
            }
        }
        __syncthreads();
        const size_t grow0 = (size_t)win * T;           // first global row (b*T + t) of this window
        if (TR) rows_to_hbm(X, LDX, D, tr.sv + (size_t)tr.x0 * 64 + grow0 * D, D, T, tid);

#pragma unroll 1
        for (int layer = 0; layer < L; ++layer) {
            const float* LW = wts + LAYER0 + (size_t)layer * LAYER_FLOATS;
            const int lbase = (int)((LAYER0 + (size_t)layer * LAYER_FLOATS) * 4);
            float* svl = TR ? tr.sv + (size_t)layer * tr.layer_stride * 64 : nullptr;   // this layer's stash
            // dropout keys of this layer's sites 1..3 (site 0, the attention probabilities, is keyed inside attention_head_regs)
            const unsigned dk1 = TR ? tip_drop_key_s(tr.seed, (unsigned)(layer * 4 + 1)) : 0u;
            const unsigned dk2 = TR ? tip_drop_key_s(tr.seed, (unsigned)(layer * 4 + 2)) : 0u;
            const unsigned dk3 = TR ? tip_drop_key_s(tr.seed, (unsigned)(layer * 4 + 3)) : 0u;
            float* Qc = C;   // attention output of one 8-head chunk [48 rows][128 channels]: the out-projection's A operand
            // ---- self-attention block: two chunks of 8 heads; wave w owns head 8c + w end to end -----------------
            f32x4 acc_o[RB][2];
            zero_acc<2>(acc_o);
            WRing<2> g_o, g_f;

#pragma unroll 1
            for (int c = 0; c < 2; ++c) {
                {
                    const int head = c * 8 + wave;
                    f32x4 acc[RB][3];
                    zero_acc<3>(acc);
                    // column blocks of this head in the packed [48 nb][16 kb] QKV matrix: Q = head, K = 16+head, V = 32+head
                    const int qsoff = lbase + (int)(QKV_W * 4) + head * 16 * 1024;
                    // Q and K are projected with swapped MFMA operands: their accumulators come out as (channels 4*lg + e,
                    // row l15) — the fragments S^T = K Q^T wants — and V in the plain layout P V wants, so this wave's head
                    // runs projection -> scores -> softmax -> P V entirely in registers: no Q/K/V planes, no barrier here.
                    gemm_phase<3, 16, (ABL & 4) != 0, 2, TR>(acc, X + l15 * LDX + lg * 4, LDX, rsrc, voff, qsoff, 16 * 16 * 1024, g_qkv,
                                                         qsoff, 16 * 16 * 1024);
                    // out-projection fragments of this chunk fly during the attention
                    ring_prefetch<2>(g_o, rsrc, voff, lbase + (int)(WO_W * 4) + ((wave * 2) * 16 + c * 8) * 1024, 16 * 1024);
                    const f32x4 bq = *reinterpret_cast<const f32x4*>(LW + QKV_B + head * 16 + lg * 4);
                    const f32x4 bk = *reinterpret_cast<const f32x4*>(LW + QKV_B + D + head * 16 + lg * 4);
                    const float bv = LW[QKV_B + 2 * D + head * 16 + l15];
                    f32x4 qt[RB], kt[RB], vv[RB];
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        qt[r] = acc[r][0] + bq;
                        kt[r] = acc[r][1] + bk;
                        vv[r] = acc[r][2] + bv;
                    }
                    if (TR) {
                        // raw q (the packed W_q carries the 1/sqrt(d_head) fold: undo it exactly), k, v -> [M, 3D]
                        float* qp = svl + (size_t)tr.qkv * 64 + grow0 * (3 * D) + head * 16;
#pragma unroll
                        for (int r = 0; r < RB; ++r) {
                            const int row = r * 16 + l15;
                            if (row < T) {
                                __builtin_nontemporal_store(qt[r] * 4.0f, reinterpret_cast<f32x4*>(qp + (size_t)row * (3 * D) + lg * 4));
                                __builtin_nontemporal_store(kt[r], reinterpret_cast<f32x4*>(qp + (size_t)row * (3 * D) + D + lg * 4));
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const int vr = r * 16 + lg * 4 + e;
                                if (vr < T) __builtin_nontemporal_store(vv[r][e], qp + (size_t)vr * (3 * D) + 2 * D + l15);
                            }
                        }
                        attention_head_regs<LDC, true>(qt, kt, vv, Qc, wave * 16, lane, 48, svl + (size_t)tr.ast * 64,
                                                       (unsigned long long)win * H + head, T, tr.seed, (unsigned)(layer * 4 + 0),
                                                       tr.thresh, tr.scale);
                    } else if (!(ABL & 1)) {
                        attention_head_regs<LDC>(qt, kt, vv, Qc, wave * 16, lane);
                    }
                }
                __syncthreads();
                if (TR) rows_to_hbm(Qc, LDC, 128, svl + (size_t)tr.att * 64 + grow0 * D + c * 128, D, T, tid);
                // the next consumer's fragments go out before this phase's MFMAs: chunk 1's QKV, or the first FFN chunk
                if (c == 0)
                    ring_prefetch<3>(g_qkv, rsrc, voff, lbase + (int)(QKV_W * 4) + (8 + wave) * 16 * 1024, 16 * 16 * 1024);
                else
                    ring_prefetch<2>(g_f, rsrc, voff, lbase + (int)(W1_W * 4) + (wave * 2) * 16 * 1024, 16 * 1024);
                // out-projection partial: acc_o += O_chunk[48 x 128] * Wo[:, 128c .. 128c+127]^T
                {
                    const int osoff = lbase + (int)(WO_W * 4) + ((wave * 2) * 16 + c * 8) * 1024;
                    gemm_phase<2, 8, (ABL & 4) != 0, 0, TR>(acc_o, Qc + l15 * LDC + lg * 4, LDC, rsrc, voff, osoff, 16 * 1024, g_o, osoff, 16 * 1024);
                }
                __syncthreads();
            }
            // residual + bias, then LayerNorm1
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int col = (wave * 2 + n) * 16 + l15;
                const float bv = LW[WO_B + col];
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc_o[r][n][e] + bv;
                        if (TR && tr.thresh)
                            v = tip_drop_hash_k(dk1, (grow0 + r * 16 + lg * 4 + e) * D + col) >= tr.thresh
                                    ? v * tr.scale : 0.f;
                        X[(r * 16 + lg * 4 + e) * LDX + col] += v;
                    }
            }
            __syncthreads();
            if (TR)
                layernorm_rows16<fz::RP, fz::LDX, true>(X, LW + G1, LW + BE1, wave, lane, svl + (size_t)tr.z1 * 64 + grow0 * D, svl + (size_t)tr.st1 * 64 + grow0 * 2,
                                     svl + (size_t)tr.x1 * 64 + grow0 * D, T);
            else if (!(ABL & 2)) layernorm_rows16<fz::RP, fz::LDX>(X, LW + G1, LW + BE1, wave, lane);
            __syncthreads();
            // ---- feed-forward block: hidden processed in 4 chunks of 256, second GEMM accumulates in registers -----
            float* Hc = C;
            f32x4 acc_f[RB][2];
            zero_acc<2>(acc_f);
#pragma unroll 1
            for (int f = 0; f < 4; ++f) {
                {
                    f32x4 acc[RB][2];
                    zero_acc<2>(acc);
                    const int nb0 = f * 16 + wave * 2;
                    const int w1off = lbase + (int)(W1_W * 4) + nb0 * 16 * 1024;
                    const int w2off = lbase + (int)(W2_W * 4) + ((wave * 2) * 64 + f * 16) * 1024;
                    // chained: the tail of linear1(f) primes the ring with linear2(f)'s first fragments
                    gemm_phase<2, 16, (ABL & 4) != 0, 0, TR>(acc, X + l15 * LDX + lg * 4, LDX, rsrc, voff, w1off, 16 * 1024, g_f, w2off, 64 * 1024);
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const int col = (wave * 2 + n) * 16 + l15;
                        const float bv = LW[W1_B + f * 256 + col];
#pragma unroll
                        for (int r = 0; r < RB; ++r)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float v = fmaxf(acc[r][n][e] + bv, 0.f);
                                if (TR && tr.thresh)
                                    v = tip_drop_hash_k(dk2, (grow0 + r * 16 + lg * 4 + e) * F + f * 256 + col) >= tr.thresh ? v * tr.scale : 0.f;
                                Hc[(r * 16 + lg * 4 + e) * LDX + col] = v;
                            }
                    }
                }
                __syncthreads();
                if (TR)
                    hidden_to_hbm<false>(Hc, LDX, svl + (size_t)tr.hid * 64 + grow0 * F + f * 256, F,
                                  reinterpret_cast<unsigned*>(svl + (size_t)tr.hid * 64 + (size_t)B * T * F) + grow0 * 32, f, T, opaque(tid));
                {
                    const int w2off = lbase + (int)(W2_W * 4) + ((wave * 2) * 64 + f * 16) * 1024;
                    // ... and the tail of linear2(f) primes it with linear1(f+1)'s (its own again after the last chunk)
                    const int nxt = f < 3 ? lbase + (int)(W1_W * 4) + ((f + 1) * 16 + wave * 2) * 16 * 1024 : w2off;
                    gemm_phase<2, 16, (ABL & 4) != 0, 0, TR>(acc_f, Hc + l15 * LDX + lg * 4, LDX, rsrc, voff, w2off, 64 * 1024, g_f, nxt,
                                      f < 3 ? 16 * 1024 : 64 * 1024);
                }
                __syncthreads();
            }
            // next layer's first QKV fragments (or the RNN input projection's) fly during the epilogue + LayerNorm2
            if (layer + 1 < L)
                ring_prefetch<3>(g_qkv, rsrc, voff, lbase + (int)(LAYER_FLOATS * 4) + (int)(QKV_W * 4) + wave * 16 * 1024,
                                 16 * 16 * 1024);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int col = (wave * 2 + n) * 16 + l15;
                const float bv = LW[W2_B + col];
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc_f[r][n][e] + bv;
                        if (TR && tr.thresh)
                            v = tip_drop_hash_k(dk3, (grow0 + r * 16 + lg * 4 + e) * D + col) >= tr.thresh
                                    ? v * tr.scale : 0.f;
                        X[(r * 16 + lg * 4 + e) * LDX + col] += v;
                    }
            }
            __syncthreads();
            if (TR)
                layernorm_rows16<fz::RP, fz::LDX, true>(X, LW + G2, LW + BE2, wave, lane, svl + (size_t)tr.z2 * 64 + grow0 * D, svl + (size_t)tr.st2 * 64 + grow0 * 2,
                                     svl + (size_t)tr.xo * 64 + grow0 * D, T);
            else if (!(ABL & 2)) layernorm_rows16<fz::RP, fz::LDX>(X, LW + G2, LW + BE2, wave, lane);
            __syncthreads();
        }
        // ---- RNN input projection (:99, first half of nn.RNN): IH = X W_ih^T + (b_ih + b_hh), rows 0..T-1 -> HBM ----
        if (ih_out) {
            f32x4 acc[RB][4];
            zero_acc<4>(acc);
            const int isoff = ih_off_b + (wave * 4) * 16 * 1024;
            WRing<4> g_ih;   // once per window: primed in place (one exposed L2 round trip per window)
            ring_prefetch<4>(g_ih, rsrc, voff, isoff, 16 * 1024);

            gemm_phase<4, 16, (ABL & 4) != 0, 0, TR>(acc, X + l15 * LDX + lg * 4, LDX, rsrc, voff, isoff, 16 * 1024, g_ih, isoff, 16 * 1024);
            float* io = ih_out + (size_t)win * T * R;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int col = (wave * 4 + n) * 16 + l15;
                const float bv = wts[ih_off_b / 4 + R * D + col];
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = r * 16 + lg * 4 + e;
                        if (row < T) io[(size_t)row * R + col] = acc[r][n][e] + bv;
                    }
            }
        }
        // ---- arm the RNN hand-off: this window's rows of HALL start as the all-ones sentinel (saves a 21-MB memset) ---
        if (hall_sentinel) {
            uint4* hp = reinterpret_cast<uint4*>(hall_sentinel + (size_t)win * T * R);
            for (int i = tid; i < T * (R / 4); i += THREADS) hp[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        }
        // ---- encoder output rows 0..T-1 -> HBM (only when a caller wants it) ---------------------------------------
        if (xout) {
            float* out = xout + (size_t)win * T * D;
            for (int i = tid; i < T * (D / 4); i += THREADS) {
                const int r = i / (D / 4), c4 = i - r * (D / 4);
                *reinterpret_cast<float4*>(out + (size_t)r * D + c4 * 4) = *reinterpret_cast<const float4*>(X + r * LDX + c4 * 4);
            }
        }
        __syncthreads();
    }
}

hipError_t launch_fused_encoder(const Dims& d, const float* fused_w, const float* x_imu, const float* x_s,
                                const float* keep_mask, float keep_scale, float* xout, float* ih_out, float* hall_sentinel,
                                int B, int T, int num_cus, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    static int abl = -1;
    if (abl < 0) {
        const char* e = tip_env("TIP_FUSED_ABLATE");   // measurement-only (profiles/): never set in production
        abl = e ? atoi(e) : 0;
    }
    const int grid = B < num_cus ? B : num_cus;
    float* iho = fused_has_rnn_ih(d) ? ih_out : nullptr;
    const int wb = (int)(fused_packed_floats(d) * 4), iob = (int)(fused_ih_off(d) * 4);
#define TIP_FUSED_LAUNCH(A)                                                                                              \
    {                                                                                                                    \
        static PerDeviceFlag attr_flag; bool& attr_set = attr_flag.cur();                                                                                    \
        if (!attr_set) {                                                                                                 \
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fused_encoder_kernel<A>),                   \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, fz::LDS_BYTES);               \
            if (e != hipSuccess) return e;                                                                               \
            attr_set = true;                                                                                             \
        }                                                                                                                \
        hipLaunchKernelGGL(fused_encoder_kernel<A>, dim3(grid), dim3(fz::THREADS), fz::LDS_BYTES, s, fused_w, x_imu, x_s, \
                           keep_mask, keep_scale, xout, iho, reinterpret_cast<unsigned*>(hall_sentinel), B, T, d.n_imu_total, d.S, d.L, wb, \
                           iob, FusedTrain{});                    \
    }
    switch (abl) {
        case 0: TIP_FUSED_LAUNCH(0) break;
        case 1: TIP_FUSED_LAUNCH(1) break;
        case 2: TIP_FUSED_LAUNCH(2) break;
        case 3: TIP_FUSED_LAUNCH(3) break;
        case 4: TIP_FUSED_LAUNCH(4) break;
        case 7: TIP_FUSED_LAUNCH(7) break;
        default: return hipErrorInvalidValue;
    }
#undef TIP_FUSED_LAUNCH
    return hipGetLastError();
}

hipError_t launch_fused_train(const Dims& d, const float* fused_w, const float* x_imu, const float* x_s, const float* keep_mask,
                              float keep_scale, float* ih_out, float* hall_sentinel, const FusedTrain& tr, int B, int T,
                              int num_cus, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (!fused_supported(d, T) || !fused_has_rnn_ih(d)) return hipErrorInvalidValue;
    static PerDeviceFlag attr_flag; bool& attr_set = attr_flag.cur();
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fused_encoder_kernel<8>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, fz::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    const int grid = B < num_cus ? B : num_cus;
    hipLaunchKernelGGL(fused_encoder_kernel<8>, dim3(grid), dim3(fz::THREADS), fz::LDS_BYTES, s, fused_w, x_imu, x_s, keep_mask,
                       keep_scale, (float*)nullptr, ih_out, reinterpret_cast<unsigned*>(hall_sentinel), B, T, d.n_imu_total, d.S, d.L,
                       (int)(fused_packed_floats(d) * 4), (int)(fused_ih_off(d) * 4), tr);
    return hipGetLastError();
}

// =====================================================================================================================
// "fusedh": the one-window kernel with a HYBRID row tiling — no matrix-core work on padded rows.
//
// T = 40 rows do not fill three 16-row MFMA blocks: in fused_encoder_kernel every weight fragment multiplies rows 0-15,
// 16-31 and 32-47, and a third of that last block's cycles (17 % of all MFMA issue time) is spent on the eight zero rows
// 40-47.  Here rows 0-31 keep v_mfma_f32_16x16x4_f32 (two row blocks), and rows 32-39 go through
// v_mfma_f32_4x4x1_16b_f32 — sixteen independent 4x4x1 blocks per instruction, 512 FLOP in 2 passes (measured 8.4-9 cycles
// per instruction against 32 for a 16x16x4: tools/probes/mfma4x4_probe.hip) — fed by THE SAME B-fragment registers:
//   the 16x16x4 weight fragment has lane (l15, lg) = W[col 16 nb + l15][k = 16 kb + 4 lg + s] in component s.  Read as the
//   B operand of a 4x4x1, lane 4 b + j is column j of block b: block b = 4 lg + (l15 >> 2), j = l15 & 3, i.e. block
//   (lg, cb) multiplies columns 4 cb .. 4 cb + 3 at k = 16 kb + 4 lg + s.  Its A operand (lane 4 b + i = row i of block b)
//   is X[row0 + i][16 kb + 4 lg + s]: one ds_read_b128 per (4-row block, k-block), lanes that differ only in cb reading the
//   same address (LDS broadcast).  The accumulator of lane (lg, cb, j) then holds, in register i, the partial sum over the
//   k == 4 lg .. 4 lg + 3 (mod 16) of output (row0 + i, column 16 nb + 4 cb + j): four k-partials per output, one per lg,
//   combined once per phase by a two-step exchange (xor 32, xor 16) that leaves row row0 + lg in lane (lg, l15).
// Per (16-column block, k-block): 8 big + 8 small MFMAs = 8 x 32 + 8 x 9 = 328 issue cycles instead of 12 x 32 = 384.
// The QKV projection does the same; its rows 32..39 reach the attention's register fragments (attention_head_regs: Q^T / K^T
// tiles with the row in the lane index) through a 1.3-KB per-wave LDS patch that transposes the tail's (row, channel) result.
// Rows 40-47 of X are zeroed once and never written again (LayerNorm covers rows 0-39): finite pad keys/values for the attention.
// Numerics: rows 0-31 are bit-identical to TIP_PLAN_FUSED (same instructions; causality keeps rows >= 32 out of them);
// rows 32-39 differ in summation order only (four k-chains per output instead of one).
// =====================================================================================================================
namespace fzh {
constexpr int RBM = 2;    // 16-row MFMA blocks (rows 0-31)
constexpr int RBT = 2;    // 4-row blocks of the tail (rows 32-35, 36-39)
constexpr int TAIL0 = 32;
}  // namespace fzh

// (the first SWAPN column blocks of the 16-row part with swapped operands, as in mfma_block: transposed accumulators)
template <int NBW, int SWAPN = 0>
__device__ __forceinline__ void mfma_block_h(f32x4 (&acc)[fzh::RBM][NBW], f32x4 (&acct)[fzh::RBT][NBW], const float4 (&a)[fzh::RBM],
                                             const float4 (&at)[fzh::RBT], const float4 (&w)[NBW]) {
#define TIP_MFMA_STEP_H(c)                                                                                            \
    _Pragma("unroll") for (int r = 0; r < fzh::RBM; ++r) _Pragma("unroll") for (int n = 0; n < NBW; ++n)               \
        acc[r][n] = n < SWAPN ? __builtin_amdgcn_mfma_f32_16x16x4f32(w[n].c, a[r].c, acc[r][n], 0, 0, 0)              \
                              : __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].c, w[n].c, acc[r][n], 0, 0, 0);             \
    _Pragma("unroll") for (int r = 0; r < fzh::RBT; ++r) _Pragma("unroll") for (int n = 0; n < NBW; ++n)               \
        acct[r][n] = __builtin_amdgcn_mfma_f32_4x4x1f32(at[r].c, w[n].c, acct[r][n], 0, 0, 0);
    TIP_MFMA_STEP_H(x)
    TIP_MFMA_STEP_H(y)
    TIP_MFMA_STEP_H(z)
    TIP_MFMA_STEP_H(w)
#undef TIP_MFMA_STEP_H
}

// acc  [r][n] += A[rows 16 r .. 16 r + 15] x Wblock(n, kb)      (r = 0, 1)
// acct [r][n] += A[rows 32 + 4 r .. + 3]  x Wblock(n, kb), per-lg k-partials (see above)
//   Am: this lane's LDS address for the 16-row blocks (base + l15 * lda + lg * 4);  At: for the tail (base + (32 + (lane & 3)) *
//   lda + lg * 4).  Weight ring semantics exactly as gemm_phase.
template <int NBW, int KB, int SWAPN = 0, bool PIN = false>
__device__ __forceinline__ void gemm_phase_h(f32x4 (&acc)[fzh::RBM][NBW], f32x4 (&acct)[fzh::RBT][NBW], const float* Am, const float* At,
                                             int lda, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff, int nstride_b, WRing<NBW>& g,
                                             int nsoff, int nnstride_b) {
    static_assert(KB % 2 == 0, "k-blocks are processed in pairs");
    float4 a0[fzh::RBM], a1[fzh::RBM], t0[fzh::RBT], t1[fzh::RBT];
#pragma unroll
    for (int r = 0; r < fzh::RBM; ++r) a0[r] = *reinterpret_cast<const float4*>(Am + r * 16 * lda);
#pragma unroll
    for (int r = 0; r < fzh::RBT; ++r) t0[r] = *reinterpret_cast<const float4*>(At + r * 4 * lda);
#pragma unroll 1
    for (int kb = 0; kb < KB; kb += 2) {
        const bool last = kb + 2 >= KB;
        const int o = last ? nsoff : soff + (kb + 2) * 1024;
        const int st = last ? nnstride_b : nstride_b;
#pragma unroll
        for (int r = 0; r < fzh::RBM; ++r) a1[r] = *reinterpret_cast<const float4*>(Am + r * 16 * lda + (kb + 1) * 16);
#pragma unroll
        for (int r = 0; r < fzh::RBT; ++r) t1[r] = *reinterpret_cast<const float4*>(At + r * 4 * lda + (kb + 1) * 16);
        mfma_block_h<NBW, SWAPN>(acc, acct, a0, t0, g.w0);
#pragma unroll
        for (int n = 0; n < NBW; ++n) g.w0[n] = load_frag(rsrc, voff, o + n * st);
        if (PIN) __builtin_amdgcn_sched_barrier(0);   // (as in gemm_phase: keeps the refill ahead of the second half's MFMAs)
#pragma unroll
        for (int r = 0; r < fzh::RBM; ++r) a0[r] = *reinterpret_cast<const float4*>(Am + r * 16 * lda + (kb + 2) * 16);
#pragma unroll
        for (int r = 0; r < fzh::RBT; ++r) t0[r] = *reinterpret_cast<const float4*>(At + r * 4 * lda + (kb + 2) * 16);
        // the NEXT pair's first A fragments are requested HERE, half an iteration ahead: left to itself the scheduler sinks these reads
        // to the top of the next iteration, right in front of the MFMA that needs them, and both waves of a SIMD — in step behind every
        // barrier — then sit out an LDS round trip per k-block pair with the matrix pipe idle
        __builtin_amdgcn_sched_barrier(0);
        mfma_block_h<NBW, SWAPN>(acc, acct, a1, t1, g.w1);
#pragma unroll
        for (int n = 0; n < NBW; ++n) g.w1[n] = load_frag(rsrc, voff, o + n * st + 1024);
        if (PIN) __builtin_amdgcn_sched_barrier(0);
    }
}

template <int NBW>
__device__ __forceinline__ void zero_acc_h(f32x4 (&acc)[fzh::RBM][NBW], f32x4 (&acct)[fzh::RBT][NBW]) {
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
#pragma unroll
        for (int r = 0; r < fzh::RBM; ++r) acc[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < fzh::RBT; ++r) acct[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
}

// Combine the four k-partials of a tail accumulator (registers = rows row0 .. row0 + 3, one partial per lg): two exchange
// steps, after which lane (lg, l15) holds the finished sum of row row0 + lg, column l15 of the block.  Every row is summed
// as (p[lg] + p[lg ^ 2]) + (p[lg ^ 1] + p[lg ^ 3]) — the same pairing for all four rows.
// Three permlane swaps (tip_layernorm.h), no LDS crossbar: swap32(v0, v2) leaves {v0.lo, v2.lo} and {v0.hi, v2.hi}, whose sum is
// row 0 in lanes 0-31 and row 2 in lanes 32-63 — exactly what "keep one row, hand the other to lane ^ 32" produces.
__device__ __forceinline__ float tail_reduce(const f32x4& v, int lg) {
    (void)lg;
    float a = v[0], b = v[2];
    swap32(a, b);
    float k0 = a + b;              // rows {0,1} live on with lg in {0,1}, rows {2,3} with lg in {2,3}
    a = v[1], b = v[3];
    swap32(a, b);
    float k1 = a + b;
    swap16(k0, k1);
    return k0 + k1;                // row 2 * (lg >> 1) + (lg & 1) = lg
}

// The V tile of rows 32..47 in the plain accumulator layout (lane (l15, lg) = rows 32 + 4 lg + e, channel l15) from the two tail
// accumulators (a0: rows 32..35, a1: rows 36..39; four k-partials each, one per lg): all four rows of block lg must end up in
// ONE lane — an all-reduce over lg rather than tail_reduce's reduce-scatter.  swap16(a0, a1) pairs lg with lg ^ 1 and parks
// block 0 in the even, block 1 in the odd lane groups; swap32 of the sums finishes both.  Lanes lg = 2, 3 (pad rows 40..47,
// multiplied by P = 0 in the attention) get zeros.
__device__ __forceinline__ f32x4 tail_gather_v(const f32x4& a0, const f32x4& a1, int lg) {
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float x = a0[e], y = a1[e];
        swap16(x, y);
        float s = x + y, t = s;
        swap32(s, t);
        o[e] = s + t;
    }
    return lg < 2 ? o : (f32x4){0.f, 0.f, 0.f, 0.f};
}

// measurement only (TIP_FUSEDH_TRACE=1): s_memtime stamps of workgroup 0 / thread 0 at the phase boundaries of layer 1
__device__ unsigned long long g_fh_trace[64];
__device__ unsigned long long g_fh_wg[2 * 1024];   // TRACE: [workgroup][entry, exit] in s_memrealtime ticks (100 MHz, device-wide)
#define FH_STAMP(slot) do { if (TRACE && blockIdx.x == 0 && tid == 0 && (layer == 1 || (slot) < 4 || (slot) >= 40)) g_fh_trace[slot] = __builtin_amdgcn_s_memtime(); } while (0)

// TR: training forward (tip_train_forward) — the encoder's four dropout sites and the activation stash of the backward, exactly
// as fused_encoder_kernel<8> writes them (same arrays, same per-element dropout keys: the masks do not depend on the tiling).
template <bool TRACE, bool TR>
__global__ __launch_bounds__(fz::THREADS) void fused_encoder_h_kernel(
    const float* __restrict__ wts, const float* __restrict__ x_imu, const float* __restrict__ x_s,
    const float* __restrict__ keep_mask, float keep_scale, float* __restrict__ xout, float* __restrict__ ih_out,
    unsigned* __restrict__ hall_sentinel, int B, int T, int NI, int S, int L, int wbytes, int ih_off_b, FusedTrain tr) {
    using namespace fz;
    using namespace fzh;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;
    float* C = smem + X_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wts), 0, wbytes, 0x00020000);
    const int voff = lane * 16;
    // this lane's LDS offsets (floats) inside a [rows][ld] operand: 16-row blocks / tail blocks
    auto am = [&](int ld) { return l15 * ld + lg * 4; };
    auto at = [&](int ld) { return (TAIL0 + (lane & 3)) * ld + lg * 4; };

    // rows 40..47 of the residual stream: zero once, never written again (finite pad keys / values for the attention)
    for (int i = tid; i < 8 * LDX; i += THREADS) X[TMAX * LDX + i] = 0.f;

    for (int win = blockIdx.x; win < B; win += gridDim.x) {
        WRing<3> g_qkv;
        {
        const int layer = -1;
        FH_STAMP(0);
        if (TRACE && tid == 0 && win == (int)blockIdx.x && blockIdx.x < 1024) g_fh_wg[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
        const int in_soff = (int)(IN_W * 4) + (wave * 2) * (KIN / 16) * 1024;
        WRing<2> g_in;
        ring_prefetch<2>(g_in, rsrc, voff, in_soff, (KIN / 16) * 1024);
        // ---- P0 prologue (:63-78) ------------------------------------------------------------------------------------
        float* U = C;
        for (int i = tid; i < RP * LDU / 4; i += THREADS) reinterpret_cast<float4*>(U)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        {
            // Round 3: rows wave, wave + 8, ... (five per wave at T = 40), a lane per column: unit-stride loads without an index
            // division, ALL of a wave's loads in flight before the first is used, unconditional (clamped addresses), and the LDS
            // stores through one base + COMPILE-TIME offsets.  The element loop this replaces (i = tid, tid + 512, ...: 18 trips,
            // each a dependent HBM round trip behind an integer division) took 17.2 k cycles per window; a first attempt at
            // batching the loads (round 2) spilled its forty store addresses and was slower.
            constexpr int NR = (TMAX + 7) / 8;
            float vi[NR][2], vs[NR][3], vk[NR][3];
            const float* kmb = keep_mask ? keep_mask : x_s;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int r = wave + 8 * j;
                const int rc = r < T ? r : T - 1;
                const float* xi = x_imu + ((size_t)win * T + rc) * NI;
                const float* xs = x_s + ((size_t)win * T + rc) * S;
                const float* km = kmb + ((size_t)win * T + rc) * S;
#pragma unroll
                for (int q = 0; q < 2; ++q) vi[j][q] = xi[lane + 64 * q < NI ? lane + 64 * q : NI - 1];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int c = lane + 64 * q < S ? lane + 64 * q : S - 1;
                    vs[j][q] = xs[c];
                    vk[j][q] = km[c];
                }
            }
            float* pu_i = U + wave * LDU + lane;
            float* pu_s = pu_i + NI;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                if (wave + 8 * j < T) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        if (lane + 64 * q < NI) pu_i[8 * j * LDU + 64 * q] = vi[j][q];
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        if (lane + 64 * q < S) {
                            float v = vs[j][q];
                            if (v != v) v = 0.f;                               // :65
                            if (keep_mask) v = v * vk[j][q] * keep_scale;     // :77 with an explicit keep-mask
                            pu_s[8 * j * LDU + 64 * q] = v;
                        }
                }
            }
        }
        __syncthreads();
        FH_STAMP(1);
        // training forward: the staged input rows are dW_in's operand in the backward — stashed from here instead of by a launch of
        // their own (tip_train.hip)
        if (TR && tr.u) rows_to_hbm_fixed<KIN>(U, LDU, tr.sv + (size_t)tr.u * 64 + (size_t)win * T * KIN, KIN, T, opaque(tid));
        // ---- P1 in_linear (:79) + channel shuffle (folded) -------------------------------------------------------------
        {
            f32x4 acc[RBM][2], acct[RBT][2];
            zero_acc_h<2>(acc, acct);
            gemm_phase_h<2, KIN / 16, 0, TR>(acc, acct, U + am(LDU), U + at(LDU), LDU, rsrc, voff, in_soff, (KIN / 16) * 1024, g_in, in_soff,
                                      (KIN / 16) * 1024);
            ring_prefetch<3>(g_qkv, rsrc, voff, (int)(LAYER0 * 4) + (int)(QKV_W * 4) + wave * 16 * 1024, 16 * 16 * 1024);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int col = (wave * 2 + n) * 16 + l15;
                const float bv = wts[IN_B + col];
#pragma unroll
                for (int r = 0; r < RBM; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) X[(r * 16 + lg * 4 + e) * LDX + col] = acc[r][n][e] + bv;
#pragma unroll
                for (int r = 0; r < RBT; ++r) X[(TAIL0 + 4 * r + lg) * LDX + col] = tail_reduce(acct[r][n], lg) + bv;
            }
        }
        __syncthreads();
        FH_STAMP(2);
        }
        const size_t grow0 = (size_t)win * T;           // first global row (b*T + t) of this window
        if (TR) rows_to_hbm_fixed<D>(X, LDX, tr.sv + (size_t)tr.x0 * 64 + grow0 * D, D, T, opaque(tid));
#pragma unroll 1
        for (int layer = 0; layer < L; ++layer) {
            FH_STAMP(8);
            const float* LW = wts + LAYER0 + (size_t)layer * LAYER_FLOATS;
            const int lbase = (int)((LAYER0 + (size_t)layer * LAYER_FLOATS) * 4);
            float* svl = TR ? tr.sv + (size_t)layer * tr.layer_stride * 64 : nullptr;   // this layer's stash
            const unsigned dk1 = TR ? tip_drop_key_s(tr.seed, (unsigned)(layer * 4 + 1)) : 0u;
            const unsigned dk2 = TR ? tip_drop_key_s(tr.seed, (unsigned)(layer * 4 + 2)) : 0u;
            const unsigned dk3 = TR ? tip_drop_key_s(tr.seed, (unsigned)(layer * 4 + 3)) : 0u;
            float bvo[2], bv2[2];   // out-projection / linear2 biases of this wave's columns: requested a phase (or eight) ahead of their use
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                bvo[n] = LW[WO_B + (wave * 2 + n) * 16 + l15];
                bv2[n] = LW[W2_B + (wave * 2 + n) * 16 + l15];
            }
            WRing<2> g_o, g_f;
            // ---- self-attention block.  Wave w owns heads w and 8 + w END TO END and runs them back to back with no workgroup
            // barrier in between: projection -> attention -> projection -> attention.  The two waves that share a SIMD drift apart
            // in this stretch, so one's softmax (VALU, LDS) runs under the other's projection MFMAs instead of both idling the
            // matrix pipe at the same time in front of a barrier; then ONE out-projection over all 16 heads (K = 256).
            float* Oc = C;   // attention output of all heads [48 rows][256 channels], leading dimension LDX: the out-projection's A operand
#pragma unroll 1
            for (int c = 0; c < 2; ++c) {
                const int head = c * 8 + wave;
                f32x4 acc[RBM][3], acct[RBT][3];
                zero_acc_h<3>(acc, acct);
                const int qsoff = lbase + (int)(QKV_W * 4) + head * 16 * 1024;
                // rows 0..31: Q and K with swapped operands (their accumulators are the attention's transposed fragments), V plain;
                // rows 32..39: all three through the 4x4x1 tail in the plain orientation.  The tail of head w's product primes
                // the ring with head 8 + w's first fragments (they fly during head w's attention).
                const int nxt = c == 0 ? qsoff + 8 * 16 * 1024 : qsoff;
                gemm_phase_h<3, 16, 2, TR>(acc, acct, X + am(LDX), X + at(LDX), LDX, rsrc, voff, qsoff, 16 * 16 * 1024, g_qkv, nxt,
                                       16 * 16 * 1024);
                if (c == 1)   // out-projection fragments fly during the second attention and the barrier
                    ring_prefetch<2>(g_o, rsrc, voff, lbase + (int)(WO_W * 4) + (wave * 2) * 16 * 1024, 16 * 1024);
                const f32x4 bq = *reinterpret_cast<const f32x4*>(LW + QKV_B + head * 16 + lg * 4);
                const f32x4 bk = *reinterpret_cast<const f32x4*>(LW + QKV_B + D + head * 16 + lg * 4);
                const float bv = LW[QKV_B + 2 * D + head * 16 + l15];
                f32x4 qt[RB], kt[RB], vv[RB];
#pragma unroll
                for (int r = 0; r < RBM; ++r) {
                    qt[r] = acc[r][0] + bq;
                    kt[r] = acc[r][1] + bk;
                    vv[r] = acc[r][2] + bv;
                }
                // Third row block (rows 32..47, eight of them real).  V: gather the tail into the plain layout.  Q, K: the tail
                // yields (row 4 rb + lg, channel l15) per lane; the attention wants (row l15, channels 4 lg ..): a 1.3-KB per-wave
                // LDS patch behind the O plane does the transpose, no workgroup barrier — a wave's LDS operations complete in
                // program order.
                vv[2] = tail_gather_v(acct[0][2], acct[1][2], lg);
                if (lg < 2) vv[2] += bv;
                {
                    constexpr int SLD = 20;                                   // 16 channels + 4: b128 reads of 8 rows conflict-free
                    float* scr = C + RP * LDX + wave * (2 * 8 * SLD);
                    const float bqp = LW[QKV_B + head * 16 + l15], bkp = LW[QKV_B + D + head * 16 + l15];
#pragma unroll
                    for (int rb = 0; rb < RBT; ++rb) {
                        scr[(4 * rb + lg) * SLD + l15] = tail_reduce(acct[rb][0], lg) + bqp;
                        scr[8 * SLD + (4 * rb + lg) * SLD + l15] = tail_reduce(acct[rb][1], lg) + bkp;
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    qt[2] = l15 < 8 ? *reinterpret_cast<const f32x4*>(scr + l15 * SLD + lg * 4) : z;
                    kt[2] = l15 < 8 ? *reinterpret_cast<const f32x4*>(scr + 8 * SLD + l15 * SLD + lg * 4) : z;
                    __builtin_amdgcn_wave_barrier();                          // (the patch is rewritten by this wave's next head)
                }
                FH_STAMP(9 + 4 * c);    // QKV projection issued and its results consumed
                if (TR) {
                    // raw q (the packed W_q carries the 1/sqrt(d_head) fold: undo it exactly), k, v -> [M, 3D]; the third block has
                    // the same fragment layout as the padded kernel's (rows 32 + l15 / rows 32 + 4 lg + e), pad rows masked by T
                    // (a buffer of this window's T rows: the pad rows' stores are dropped by the range check, not by a branch each)
                    const __amdgpu_buffer_rsrc_t qp_rs = tip_rows_buffer(svl + (size_t)tr.qkv * 64 + grow0 * (3 * D), T * 3 * D * 4);
                    const int lo = opaque(lane), l15o = lo & 15, lgo = lo >> 4;
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        const int qo = ((r * 16 + l15o) * (3 * D) + head * 16 + lgo * 4) * 4;
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, qt[r] * 4.0f), qp_rs, qo, 0, kTipNT);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, kt[r]), qp_rs, qo + D * 4, 0, kTipNT);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(vv[r][e]), qp_rs,
                                                                  ((r * 16 + lgo * 4 + e) * (3 * D) + 2 * D + head * 16 + l15o) * 4, 0, kTipNT);
                    }
                    attention_head_regs<LDX, true>(qt, kt, vv, Oc, head * 16, lo, TMAX, svl + (size_t)tr.ast * 64,
                                                   (unsigned long long)win * H + head, T, tr.seed, (unsigned)(layer * 4 + 0), tr.thresh,
                                                   tr.scale);
                } else {
                    attention_head_regs<LDX>(qt, kt, vv, Oc, head * 16, lane, TMAX);   // rows 40..47 of the O plane are never read
                }
                FH_STAMP(10 + 4 * c);
            }
            __syncthreads();
            FH_STAMP(15);
            if (TR) rows_to_hbm_fixed<D>(Oc, LDX, svl + (size_t)tr.att * 64 + grow0 * D, D, T, opaque(tid));
            ring_prefetch<2>(g_f, rsrc, voff, lbase + (int)(W1_W * 4) + (wave * 2) * 16 * 1024, 16 * 1024);
            f32x4 acc_o[RBM][2], acc_ot[RBT][2];
            zero_acc_h<2>(acc_o, acc_ot);
            {
                const int osoff = lbase + (int)(WO_W * 4) + (wave * 2) * 16 * 1024;
                gemm_phase_h<2, 16, 0, TR>(acc_o, acc_ot, Oc + am(LDX), Oc + at(LDX), LDX, rsrc, voff, osoff, 16 * 1024, g_o, osoff, 16 * 1024);
            }
            __syncthreads();
            FH_STAMP(16);
            // residual + bias, then LayerNorm1.  The residual values of this lane's 20 elements are READ before the first is written
            // back: `X[i] += v` per element compiles to a chain of ds_read -> wait -> ds_write round trips (2-3 k cycles per epilogue
            // with the matrix pipe idle; the optimiser may not move a read across a write it cannot prove disjoint).
            float xr[2][RBM][4], xt[2][RBT];
#pragma unroll
            for (int n = 0; n < 2; ++n) {   // (the unroll pragma above binds to this loop)
                const int col = (wave * 2 + n) * 16 + l15;
#pragma unroll
                for (int r = 0; r < RBM; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xr[n][r][e] = X[(r * 16 + lg * 4 + e) * LDX + col];
#pragma unroll
                for (int r = 0; r < RBT; ++r) xt[n][r] = X[(TAIL0 + 4 * r + lg) * LDX + col];
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int col = (wave * 2 + n) * 16 + l15;
                const float bv = bvo[n];
#pragma unroll
                for (int r = 0; r < RBM; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc_o[r][n][e] + bv;
                        if (TR && tr.thresh)
                            v = tip_drop_hash_k(dk1, (grow0 + r * 16 + lg * 4 + e) * D + col) >= tr.thresh ? v * tr.scale : 0.f;
                        X[(r * 16 + lg * 4 + e) * LDX + col] = xr[n][r][e] + v;
                    }
#pragma unroll
                for (int r = 0; r < RBT; ++r) {
                    float v = tail_reduce(acc_ot[r][n], lg) + bv;
                    if (TR && tr.thresh)
                        v = tip_drop_hash_k(dk1, (grow0 + TAIL0 + 4 * r + lg) * D + col) >= tr.thresh ? v * tr.scale : 0.f;
                    X[(TAIL0 + 4 * r + lg) * LDX + col] = xt[n][r] + v;
                }
            }
            __syncthreads();
            FH_STAMP(17);
            if (TR)
                layernorm_rows16<fz::TMAX, fz::LDX, true>(X, LW + G1, LW + BE1, wave, opaque(lane), svl + (size_t)tr.z1 * 64 + grow0 * D,
                                                          svl + (size_t)tr.st1 * 64 + grow0 * 2, svl + (size_t)tr.x1 * 64 + grow0 * D, T);
            else
                layernorm_rows16<fz::TMAX, fz::LDX>(X, LW + G1, LW + BE1, wave, lane);   // rows 40..47 stay zero
            __syncthreads();
            FH_STAMP(18);
            // ---- feed-forward block: hidden in 4 chunks of 256, linear2 accumulates in registers -------------------------
            float* Hc = C;
            f32x4 acc_f[RBM][2], acc_ft[RBT][2];
            zero_acc_h<2>(acc_f, acc_ft);
#pragma unroll 1
            for (int f = 0; f < 4; ++f) {
                {
                    f32x4 acc[RBM][2], acct[RBT][2];
                    zero_acc_h<2>(acc, acct);
                    const int nb0 = f * 16 + wave * 2;
                    const int w1off = lbase + (int)(W1_W * 4) + nb0 * 16 * 1024;
                    const int w2off = lbase + (int)(W2_W * 4) + ((wave * 2) * 64 + f * 16) * 1024;
                    // (the epilogue's bias is requested BEFORE the product: after it, the L2 round trip would be exposed 16 times a layer)
                    float bv1[2];
#pragma unroll
                    for (int n = 0; n < 2; ++n) bv1[n] = LW[W1_B + f * 256 + (wave * 2 + n) * 16 + l15];
                    gemm_phase_h<2, 16, 0, TR>(acc, acct, X + am(LDX), X + at(LDX), LDX, rsrc, voff, w1off, 16 * 1024, g_f, w2off, 64 * 1024);
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        const int col = (wave * 2 + n) * 16 + l15;
                        const float bv = bv1[n];
#pragma unroll
                        for (int r = 0; r < RBM; ++r)
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float v = fmaxf(acc[r][n][e] + bv, 0.f);
                                if (TR && tr.thresh)
                                    v = tip_drop_hash_k(dk2, (grow0 + r * 16 + lg * 4 + e) * F + f * 256 + col) >= tr.thresh ? v * tr.scale : 0.f;
                                Hc[(r * 16 + lg * 4 + e) * LDX + col] = v;
                            }
#pragma unroll
                        for (int r = 0; r < RBT; ++r) {
                            float v = fmaxf(tail_reduce(acct[r][n], lg) + bv, 0.f);
                            if (TR && tr.thresh)
                                v = tip_drop_hash_k(dk2, (grow0 + TAIL0 + 4 * r + lg) * F + f * 256 + col) >= tr.thresh ? v * tr.scale : 0.f;
                            Hc[(TAIL0 + 4 * r + lg) * LDX + col] = v;
                        }
                    }
                }
                FH_STAMP(19 + 3 * f);
                __syncthreads();
                FH_STAMP(20 + 3 * f);
                if (TR)   // (the gate bits sit behind the layer's [M][F] hidden rows: saved_layout, tip_train.hip)
                    hidden_to_hbm<true>(Hc, LDX, svl + (size_t)tr.hid * 64 + grow0 * F + f * 256, F,
                                  reinterpret_cast<unsigned*>(svl + (size_t)tr.hid * 64 + (size_t)B * T * F) + grow0 * 32, f, T, opaque(tid));
                {
                    const int w2off = lbase + (int)(W2_W * 4) + ((wave * 2) * 64 + f * 16) * 1024;
                    const int nxt = f < 3 ? lbase + (int)(W1_W * 4) + ((f + 1) * 16 + wave * 2) * 16 * 1024 : w2off;
                    gemm_phase_h<2, 16, 0, TR>(acc_f, acc_ft, Hc + am(LDX), Hc + at(LDX), LDX, rsrc, voff, w2off, 64 * 1024, g_f, nxt,
                                        f < 3 ? 16 * 1024 : 64 * 1024);
                }
                __syncthreads();
                FH_STAMP(21 + 3 * f);
            }
            if (layer + 1 < L)
                ring_prefetch<3>(g_qkv, rsrc, voff, lbase + (int)(LAYER_FLOATS * 4) + (int)(QKV_W * 4) + wave * 16 * 1024, 16 * 16 * 1024);
#pragma unroll
            for (int n = 0; n < 2; ++n) {   // residual values first (as after the out-projection)
                const int col = (wave * 2 + n) * 16 + l15;
#pragma unroll
                for (int r = 0; r < RBM; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xr[n][r][e] = X[(r * 16 + lg * 4 + e) * LDX + col];
#pragma unroll
                for (int r = 0; r < RBT; ++r) xt[n][r] = X[(TAIL0 + 4 * r + lg) * LDX + col];
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int col = (wave * 2 + n) * 16 + l15;
                const float bv = bv2[n];
#pragma unroll
                for (int r = 0; r < RBM; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = acc_f[r][n][e] + bv;
                        if (TR && tr.thresh)
                            v = tip_drop_hash_k(dk3, (grow0 + r * 16 + lg * 4 + e) * D + col) >= tr.thresh ? v * tr.scale : 0.f;
                        X[(r * 16 + lg * 4 + e) * LDX + col] = xr[n][r][e] + v;
                    }
#pragma unroll
                for (int r = 0; r < RBT; ++r) {
                    float v = tail_reduce(acc_ft[r][n], lg) + bv;
                    if (TR && tr.thresh)
                        v = tip_drop_hash_k(dk3, (grow0 + TAIL0 + 4 * r + lg) * D + col) >= tr.thresh ? v * tr.scale : 0.f;
                    X[(TAIL0 + 4 * r + lg) * LDX + col] = xt[n][r] + v;
                }
            }
            __syncthreads();
            FH_STAMP(31);
            if (TR)
                layernorm_rows16<fz::TMAX, fz::LDX, true>(X, LW + G2, LW + BE2, wave, opaque(lane), svl + (size_t)tr.z2 * 64 + grow0 * D,
                                                          svl + (size_t)tr.st2 * 64 + grow0 * 2, svl + (size_t)tr.xo * 64 + grow0 * D, T);
            else
                layernorm_rows16<fz::TMAX, fz::LDX>(X, LW + G2, LW + BE2, wave, lane);
            __syncthreads();
            FH_STAMP(32);
        }
        { const int layer = -1; FH_STAMP(40); }
        // ---- RNN input projection: IH = X W_ih^T + (b_ih + b_hh), rows 0..T-1 -> HBM -------------------------------------
        if (ih_out) {
            f32x4 acc[RBM][4], acct[RBT][4];
            zero_acc_h<4>(acc, acct);
            const int isoff = ih_off_b + (wave * 4) * 16 * 1024;
            WRing<4> g_ih;
            ring_prefetch<4>(g_ih, rsrc, voff, isoff, 16 * 1024);
            // Round 3: the 16-row blocks with SWAPPED operands — transposed accumulators, (columns 16 n + 4 lg + e, row 16 r + l15): a
            // lane stores four consecutive columns of one row with ONE 16-byte store instead of four scattered 4-byte ones (the
            // store tail is issue-bound: 32 -> 8 store instructions per wave for rows 0-31).  Same products, same sums.
            gemm_phase_h<4, 16, 4, TR>(acc, acct, X + am(LDX), X + at(LDX), LDX, rsrc, voff, isoff, 16 * 1024, g_ih, isoff, 16 * 1024);
            // This window's IH rows as a buffer of T rows: the stores of the pad rows are dropped by the range check, not by a branch
            // around each store.  The biases of all four column blocks are requested BEFORE the first store: loaded inside the loop, each
            // block's bias load sat behind the previous block's stores (the optimiser may not hoist a load over a store it cannot prove
            // disjoint) — four exposed L2 round trips at the end of every window.
            const __amdgpu_buffer_rsrc_t io_rs = __builtin_amdgcn_make_buffer_rsrc(ih_out + (size_t)win * T * R, 0, T * R * 4, 0x00020000);
            // (opaque lane index: the store offsets are formed here, not hoisted to the top of the kernel and carried in scratch)
            const int lo = opaque(lane), l15o = lo & 15, lgo = lo >> 4;
            float bvs[4];
            f32x4 bv4s[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                bvs[n] = wts[ih_off_b / 4 + R * D + (wave * 4 + n) * 16 + l15o];
                bv4s[n] = *reinterpret_cast<const f32x4*>(wts + ih_off_b / 4 + R * D + (wave * 4 + n) * 16 + lgo * 4);
            }
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int col = (wave * 4 + n) * 16 + l15o;
#pragma unroll
                for (int r = 0; r < RBM; ++r) {
                    const int row = r * 16 + l15o;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[r][n] + bv4s[n]), io_rs, (row * R + (wave * 4 + n) * 16 + lgo * 4) * 4, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < RBT; ++r) {
                    const float v = tail_reduce(acct[r][n], lg) + bvs[n];
                    const int row = TAIL0 + 4 * r + lgo;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), io_rs, (row * R + col) * 4, 0, 0);
                }
            }
        }
        if (hall_sentinel) {
            uint4* hp = reinterpret_cast<uint4*>(hall_sentinel + (size_t)win * T * R);
            for (int i = opaque(tid); i < T * (R / 4); i += THREADS) hp[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        }
        if (xout) {
            float* out = xout + (size_t)win * T * D;
            for (int i = tid; i < T * (D / 4); i += THREADS) {
                const int r = i / (D / 4), c4 = i - r * (D / 4);
                *reinterpret_cast<float4*>(out + (size_t)r * D + c4 * 4) = *reinterpret_cast<const float4*>(X + r * LDX + c4 * 4);
            }
        }
        __syncthreads();
        { const int layer = -1; FH_STAMP(41); }
        if (TRACE && tid == 0 && blockIdx.x < 1024) g_fh_wg[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}
#undef FH_STAMP

hipError_t launch_fused_encoder_h(const Dims& d, const float* fused_w, const float* x_imu, const float* x_s,
                                  const float* keep_mask, float keep_scale, float* xout, float* ih_out, float* hall_sentinel,
                                  int B, int T, int num_cus, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    static PerDeviceFlag attr_flag; bool& attr_set = attr_flag.cur();
    if (!attr_set) {
        for (const void* f : {reinterpret_cast<const void*>(fused_encoder_h_kernel<false, false>), reinterpret_cast<const void*>(fused_encoder_h_kernel<true, false>)}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, fz::LDS_BYTES);
            if (e != hipSuccess) return e;
        }
        attr_set = true;
    }
    static int trace = -1;
    if (trace < 0) trace = tip_env("TIP_FUSEDH_TRACE") ? 1 : 0;
    const int grid = B < num_cus ? B : num_cus;
    float* iho = fused_has_rnn_ih(d) ? ih_out : nullptr;
    if (trace)
        hipLaunchKernelGGL((fused_encoder_h_kernel<true, false>), dim3(grid), dim3(fz::THREADS), fz::LDS_BYTES, s, fused_w, x_imu, x_s, keep_mask,
                           keep_scale, xout, iho, reinterpret_cast<unsigned*>(hall_sentinel), B, T, d.n_imu_total, d.S, d.L,
                           (int)(fused_packed_floats(d) * 4), (int)(fused_ih_off(d) * 4), FusedTrain{});
    else
        hipLaunchKernelGGL((fused_encoder_h_kernel<false, false>), dim3(grid), dim3(fz::THREADS), fz::LDS_BYTES, s, fused_w, x_imu, x_s, keep_mask,
                           keep_scale, xout, iho, reinterpret_cast<unsigned*>(hall_sentinel), B, T, d.n_imu_total, d.S, d.L,
                           (int)(fused_packed_floats(d) * 4), (int)(fused_ih_off(d) * 4), FusedTrain{});
    return hipGetLastError();
}

// training forward on the hybrid tiling (TIP_TRAIN_FWD_PADDED=1 selects fused_encoder_kernel<8>, the 48-row kernel, for A/B runs)
hipError_t launch_fused_train_h(const Dims& d, const float* fused_w, const float* x_imu, const float* x_s, const float* keep_mask,
                                float keep_scale, float* ih_out, float* hall_sentinel, const FusedTrain& tr, int B, int T,
                                int num_cus, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (!fused_supported(d, T) || !fused_has_rnn_ih(d)) return hipErrorInvalidValue;
    static PerDeviceFlag attr_flag; bool& attr_set = attr_flag.cur();
    if (!attr_set) {
        for (const void* f : {reinterpret_cast<const void*>(fused_encoder_h_kernel<false, true>), reinterpret_cast<const void*>(fused_encoder_h_kernel<true, true>)}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, fz::LDS_BYTES);
            if (e != hipSuccess) return e;
        }
        attr_set = true;
    }
    const int grid = B < num_cus ? B : num_cus;
    static int trace = -1;
    if (trace < 0) trace = tip_env("TIP_FUSEDH_TRACE") ? 1 : 0;
    if (trace) {   // measurement: the same phase stamps as the inference kernel (tools/fh_trace.py --train)
        hipLaunchKernelGGL((fused_encoder_h_kernel<true, true>), dim3(grid), dim3(fz::THREADS), fz::LDS_BYTES, s, fused_w, x_imu, x_s,
                           keep_mask, keep_scale, (float*)nullptr, ih_out, reinterpret_cast<unsigned*>(hall_sentinel), B, T, d.n_imu_total,
                           d.S, d.L, (int)(fused_packed_floats(d) * 4), (int)(fused_ih_off(d) * 4), tr);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((fused_encoder_h_kernel<false, true>), dim3(grid), dim3(fz::THREADS), fz::LDS_BYTES, s, fused_w, x_imu, x_s,
                       keep_mask, keep_scale, (float*)nullptr, ih_out, reinterpret_cast<unsigned*>(hall_sentinel), B, T, d.n_imu_total,
                       d.S, d.L, (int)(fused_packed_floats(d) * 4), (int)(fused_ih_off(d) * 4), tr);
    return hipGetLastError();
}

}  // namespace tip
extern "C" int tip_debug_read_fh_wg(unsigned long long* out, int n) {
    if (!out || n < 0 || n > 2048) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tip::g_fh_wg), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -5;
}
extern "C" int tip_debug_read_fh_trace(unsigned long long* out, int n) {
    if (!out || n < 0 || n > 64) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tip::g_fh_trace), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -5;
}
namespace tip {
// =====================================================================================================================
// Training step, backward of one encoder layer's feed-forward block, fused per window.
//   dy = dL/d(layer output)  ->  LayerNorm2 backward  ->  dz2 (kept: it is also the residual path into the block's input)
//   dff2 = dz2 * keep3                      (gradient into linear2's output; also linear2's bias-gradient rows)
//   dpre = (dff2 W2) * [hid > 0] / (1 - p)  (hid was saved AFTER ReLU and dropout: its sign is the gate)
//   dx1  = dz2 + dpre W1
// Same structure as the forward FFN phases of fused_encoder_kernel — activations [48 x 256] resident in LDS, the hidden
// dimension walked in 4 chunks of 256, weight fragments of W2^T ([F][D]) and W1^T ([D][F]) streamed from the backward
// image with the same ring/offset scheme as W1 / W2 in the forward — so the two 10 240-row GEMMs and the LayerNorm
// kernel of the layer-by-layer backward (and the HBM round trips between them) become one launch.
// dff2 and dpre still go to HBM: the weight-gradient GEMMs (reduction over all rows of the batch) read them.
// =====================================================================================================================
namespace fb {
constexpr size_t W2T = 0;                                  // [F][D] fragments: B operand of dhid = dff2 W2
constexpr size_t W1T = W2T + (size_t)fz::F * fz::D;        // [D][F] fragments: B operand of dx1 = dpre W1
constexpr size_t WOT = W1T + (size_t)fz::D * fz::F;        // [D][D]  fragments of Wo^T:   datt = datt_o Wo
constexpr size_t WQT = WOT + (size_t)fz::D * fz::D;        // [D][3D] fragments of Wqkv^T: dx_in = dqkv Wqkv
constexpr size_t LAYER_FLOATS2 = WQT + (size_t)fz::D * 3 * fz::D;   // per-layer stride of the backward image

}  // namespace fb

size_t fused_bwd_image_floats(const Dims& d) {
    if (fused_packed_floats(d) == 0) return 0;
    return (size_t)d.L * fb::LAYER_FLOATS2 + fz::TAIL_PAD;
}

void fused_bwd_pack_ops(const Dims& d, const float* const* t, size_t base, std::vector<PackOp>& ops) {
    using namespace fz;
    for (int l = 0; l < d.L; ++l) {
        const float* const* lw = t + 2 + 12 * l;
        const size_t L = base + (size_t)l * fb::LAYER_FLOATS2;
        PackOp a = mk_op(lw[6], L + fb::W2T, F, D, F, D, 1);   // linear2.weight is [D][F]: logical W2^T [F][D]
        a.transpose = 1;
        ops.push_back(a);
        PackOp b = mk_op(lw[4], L + fb::W1T, D, F, D, F, 1);   // linear1.weight is [F][D]: logical W1^T [D][F]
        b.transpose = 1;
        ops.push_back(b);
        PackOp o = mk_op(lw[2], L + fb::WOT, D, D, D, D, 1);         // out_proj.weight [D][D]: logical Wo^T
        o.transpose = 1;
        ops.push_back(o);
        PackOp q = mk_op(lw[0], L + fb::WQT, D, 3 * D, D, 3 * D, 1);  // in_proj_weight [3D][D]: logical Wqkv^T [D][3D]
        q.transpose = 1;
        ops.push_back(q);
    }
}

// measurement only (TIP_BWD_TRACE=1): s_memtime stamps of workgroup 0 / thread 0 at the phase boundaries of the two fused
// backward kernels ([0..7] ffn_bwd, [8..15] attn_bwd), read back with tip_debug_read_bwd_trace() (tools/bwd_trace.py)
__device__ unsigned long long g_bwd_trace[16];
__device__ __forceinline__ void bwd_stamp(int on, int slot) {
    if (on && blockIdx.x == 0 && threadIdx.x == 0) g_bwd_trace[slot] = __builtin_amdgcn_s_memtime();
}


// BITS: the ReLU / dropout gates come as the bit array the fused training forward wrote (hidden_to_hbm), staged into LDS once per
// window; otherwise (layer-by-layer forward) from the saved fp32 hidden rows.
constexpr int kGateLd = 36;   // dwords per row of the LDS copy (32 + 4: the 16 rows a lane group reads sit on distinct bank quads)
template <bool BITS>
__global__ __launch_bounds__(fz::THREADS) void ffn_bwd_kernel(FfnBwdArgs a, int B, int T) {
    using namespace fz;
    using namespace fzh;   // hybrid row tiling as in fused_encoder_h_kernel: rows 0-31 on 16x16x4, rows 32-39 on 4x4x1 MFMAs, nothing on pad rows
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* G = smem;                       // dy -> dz2 -> dx1
    float* Mb = smem + RP * LDX;           // dff2 (masked dz2)
    float* Hc = smem + 2 * RP * LDX;       // dpre chunk; scratch for the LayerNorm parameter partials before that
    unsigned* GB = reinterpret_cast<unsigned*>(smem + 3 * RP * LDX);   // BITS: this window's gate bits [RP][kGateLd], dword f*8 + half*4 + e
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wimg), 0, a.wbytes, 0x00020000);
    const int voff = lane * 16;
    const int lbase = (int)(((size_t)a.layer * fb::LAYER_FLOATS2) * 4);
    auto am = [&](int ld) { return l15 * ld + lg * 4; };                       // this lane's LDS offsets: 16-row blocks / tail blocks
    auto at = [&](int ld) { return (TAIL0 + (lane & 3)) * ld + lg * 4; };
    // the saved hidden activations through a buffer descriptor (byte offsets must fit 32 bits: checked by the launcher)
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.hid), 0, a.hid_bytes, 0x00020000);

    const unsigned dkey = tip_drop_key_s(a.seed, a.site);
    for (int win = blockIdx.x; win < B; win += gridDim.x) {
        const size_t grow0 = (size_t)win * T;
        bwd_stamp(a.trace, 0);
        WRing<2> g_f;
        ring_prefetch<2>(g_f, rsrc, voff, lbase + (int)(fb::W2T * 4) + (wave * 2) * 16 * 1024, 16 * 1024);
        u32x4 gq = {0u, 0u, 0u, 0u};   // BITS: 16 bytes of this window's 40 x 128 B of gate bits per thread (tid < 8 T), to LDS below
        if (BITS && tid < 8 * T) gq = *reinterpret_cast<const u32x4*>(a.gbits + (grow0 + (tid >> 3)) * 32 + (tid & 7) * 4);
        // ---- LayerNorm2 backward, one wave per row (rows w, w+8, ...) ------------------------------------------------------
        {
            const float4 gg = *reinterpret_cast<const float4*>(a.g2 + lane * 4);
            float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg, dm = dg;
            // every row's operands are requested BEFORE the first row is worked on: written as one load -> compute -> store body
            // per row, the stores of row i stood between the optimiser and the loads of row i + 1 (it may not move a load across a
            // store it cannot prove disjoint), and the phase was three HBM round trips long (14 000 cycles for 5 rows per wave)
            constexpr int NR = RP / 8;
            float4 zq[NR], dq_[NR];
            float2 sq_[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int r = wave + 8 * i;
                const size_t gr = grow0 + (r < T ? r : 0);      // pad rows: a valid address, the values are not used
                zq[i] = *reinterpret_cast<const float4*>(a.z2 + gr * D + lane * 4);
                dq_[i] = *reinterpret_cast<const float4*>(a.dy + gr * D + lane * 4);
                sq_[i] = *reinterpret_cast<const float2*>(a.st2 + gr * 2);
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int r = wave + 8 * i;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f), m = o;
                if (r < T) {
                    const size_t gr = grow0 + r;
                    const float4 zv = zq[i];
                    const float4 dyv = dq_[i];
                    const float mean = sq_[i].x, rstd = sq_[i].y;
                    float4 xh;
                    xh.x = (zv.x - mean) * rstd; xh.y = (zv.y - mean) * rstd; xh.z = (zv.z - mean) * rstd; xh.w = (zv.w - mean) * rstd;
                    const float ax = dyv.x * gg.x, ay = dyv.y * gg.y, az = dyv.z * gg.z, aw = dyv.w * gg.w;
                    const float m1 = wsum((ax + ay) + (az + aw)) * (1.f / D);
                    const float m2 = wsum((ax * xh.x + ay * xh.y) + (az * xh.z + aw * xh.w)) * (1.f / D);
                    dg.x += dyv.x * xh.x; dg.y += dyv.y * xh.y; dg.z += dyv.z * xh.z; dg.w += dyv.w * xh.w;
                    db.x += dyv.x; db.y += dyv.y; db.z += dyv.z; db.w += dyv.w;
                    o.x = rstd * (ax - m1 - xh.x * m2); o.y = rstd * (ay - m1 - xh.y * m2);
                    o.z = rstd * (az - m1 - xh.z * m2); o.w = rstd * (aw - m1 - xh.w * m2);
                    m = o;
                    if (a.thresh) {
                        const unsigned long long idx = gr * D + lane * 4;
                        m.x = tip_drop_hash_k(dkey, idx) >= a.thresh ? o.x * a.scale : 0.f;
                        m.y = tip_drop_hash_k(dkey, idx + 1) >= a.thresh ? o.y * a.scale : 0.f;
                        m.z = tip_drop_hash_k(dkey, idx + 2) >= a.thresh ? o.z * a.scale : 0.f;
                        m.w = tip_drop_hash_k(dkey, idx + 3) >= a.thresh ? o.w * a.scale : 0.f;
                    }
                    *reinterpret_cast<float4*>(a.dff2 + gr * D + lane * 4) = m;
                    dm.x += m.x; dm.y += m.y; dm.z += m.z; dm.w += m.w;
                }
                *reinterpret_cast<float4*>(G + r * LDX + lane * 4) = o;      // padded rows: zeros
                *reinterpret_cast<float4*>(Mb + r * LDX + lane * 4) = m;
            }
            // per-window partials of dgamma2 | dbeta2 | d(linear2 bias): waves -> LDS -> one [3*D] row per window
            float* red = Hc;   // [8 waves][3][D]
            *reinterpret_cast<float4*>(red + (wave * 3 + 0) * D + lane * 4) = dg;
            *reinterpret_cast<float4*>(red + (wave * 3 + 1) * D + lane * 4) = db;
            *reinterpret_cast<float4*>(red + (wave * 3 + 2) * D + lane * 4) = dm;
        }
        if (BITS && tid < 8 * T) {
            // global dwords (f*8 + 2 e + half) -> LDS dwords (f*8 + 4 half + e): a lane's four e of one half are then one 16-byte read
            const int row = tid >> 3, q = tid & 7, fq = q >> 1, e0 = (q & 1) * 2;
            unsigned* g = GB + row * kGateLd + fq * 8;
            g[e0] = gq[0]; g[4 + e0] = gq[1]; g[e0 + 1] = gq[2]; g[4 + e0 + 1] = gq[3];
        }
        __syncthreads();
        for (int i = tid; i < 3 * D; i += THREADS) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += Hc[w * 3 * D + i];
            a.lnpart[(size_t)win * (3 * D + F) + i] = s;
        }
        __syncthreads();
        bwd_stamp(a.trace, 1);
        // ---- hidden chunks ------------------------------------------------------------------------------------------------
        f32x4 acc_x[RBM][2], acc_xt[RBT][2];
        zero_acc_h<2>(acc_x, acc_xt);
#pragma unroll 1
        for (int f = 0; f < 4; ++f) {
            {
                f32x4 acc[RBM][2], acct[RBT][2];
                zero_acc_h<2>(acc, acct);
                const int w2off = lbase + (int)(fb::W2T * 4) + (f * 16 + wave * 2) * 16 * 1024;
                const int w1off = lbase + (int)(fb::W1T * 4) + ((wave * 2) * 64 + f * 16) * 1024;
                // The product is computed TRANSPOSED (weights as the A operand): a lane then holds 4 consecutive hidden
                // channels of ONE row, so the ReLU/dropout gate is one 16-byte load of the saved hidden row and the chunk goes
                // to LDS as one 16-byte store per tile.  The gate loads are issued before the MFMAs (buffer loads: opaque to the
                // optimiser, so they stay here) and consumed after.
                f32x4 gate[RBM][2];
                float gatet[RBT][2];   // the tail comes out in the plain orientation: lane (lg, l15) = row 32 + 4 rb + lg, channel l15
                u32x4 gw[RBM];         // BITS: row r*16 + l15, dword e: bit ((wave & 3)*2 + n)*4 + lg is the gate of element (r, n, e)
                unsigned gwt[RBT];     // BITS: row 32 + 4 rb + lg, dword of e = l15 & 3: bit ((wave & 3)*2 + n)*4 + (l15 >> 2)
                if (BITS) {
                    const int half = wave >> 2;
#pragma unroll
                    for (int r = 0; r < RBM; ++r) gw[r] = *reinterpret_cast<const u32x4*>(GB + (r * 16 + l15) * kGateLd + f * 8 + half * 4);
#pragma unroll
                    for (int rb = 0; rb < RBT; ++rb) gwt[rb] = GB[(TAIL0 + 4 * rb + lg) * kGateLd + f * 8 + half * 4 + (l15 & 3)];
                }
                if (!BITS) {
#pragma unroll
                for (int n = 0; n < 2; ++n) {
#pragma unroll
                    for (int r = 0; r < RBM; ++r) {
                        const int row = r * 16 + l15;
                        const long long off = ((long long)(grow0 + (row < T ? row : 0)) * F + f * 256 + (wave * 2 + n) * 16 + lg * 4) * 4;
                        gate[r][n] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hrs, (int)off, 0, 0));
                    }
#pragma unroll
                    for (int rb = 0; rb < RBT; ++rb) {
                        const int row = TAIL0 + 4 * rb + lg;
                        const long long off = ((long long)(grow0 + (row < T ? row : 0)) * F + f * 256 + (wave * 2 + n) * 16 + l15) * 4;
                        gatet[rb][n] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hrs, (int)off, 0, 0));
                    }
                }
                }
                // the tail of this phase primes the ring with the dx1 phase's first fragments
                gemm_phase_h<2, 16, 2, true>(acc, acct, Mb + am(LDX), Mb + at(LDX), LDX, rsrc, voff, w2off, 16 * 1024, g_f, w1off, 64 * 1024);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    f32x4 bs = {0.f, 0.f, 0.f, 0.f};   // this window's share of linear1's bias gradient: column sums of the chunk
#pragma unroll
                    for (int r = 0; r < RBM; ++r) {
                        const int row = r * 16 + l15;
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const bool open = BITS ? ((gw[r][e] >> (((wave & 3) * 2 + n) * 4 + lg)) & 1u) != 0u : gate[r][n][e] > 0.f;
                            v[e] = (row < T && open) ? acc[r][n][e] * a.gate_scale : 0.f;
                        }
                        *reinterpret_cast<f32x4*>(Hc + row * LDX + (wave * 2 + n) * 16 + lg * 4) = v;
                        bs += v;
                    }
                    float ts = 0.f;   // rows 32..39: lane (lg, l15) = (row 32 + 4 rb + lg, channel l15) after the tail's reduce-scatter
#pragma unroll
                    for (int rb = 0; rb < RBT; ++rb) {
                        const int row = TAIL0 + 4 * rb + lg;
                        const float x = tail_reduce(acct[rb][n], lg);
                        const bool open = BITS ? ((gwt[rb] >> (((wave & 3) * 2 + n) * 4 + (l15 >> 2))) & 1u) != 0u : gatet[rb][n] > 0.f;
                        const float v = (row < T && open) ? x * a.gate_scale : 0.f;
                        Hc[row * LDX + (wave * 2 + n) * 16 + l15] = v;
                        ts += v;
                    }
                    ts = lg4_sum(ts);   // every lane (any lg, l15): the tail rows' sum of channel l15
                    // the lane holds 4 consecutive hidden channels of rows l15 and 16 + l15: the sum over rows is a butterfly over the 16
                    // lanes that share lg, straight from the registers; the tail's share of channel 4 lg + e sits in lane l15 = 4 lg + e
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float x = bs[e];
                        x = row16_sum(x);
                        bs[e] = x + __shfl(ts, (lane & 48) | (lg * 4 + e), 64);
                    }
                    if (l15 == 0)
                        *reinterpret_cast<f32x4*>(a.lnpart + (size_t)win * (3 * D + F) + 3 * D + f * 256 + (wave * 2 + n) * 16 + lg * 4) = bs;
                }
            }
            if (f == 0) bwd_stamp(a.trace, 2);     // first chunk: d(hidden) product done
            __syncthreads();
            if (f == 0) bwd_stamp(a.trace, 3);
            rows_to_hbm_fixed<256>(Hc, LDX, a.dpre + grow0 * F + f * 256, F, T, tid);   // (a counted number of stores in front of the dx1 phase's ring)
            {
                const int w1off = lbase + (int)(fb::W1T * 4) + ((wave * 2) * 64 + f * 16) * 1024;
                const int nxt = f < 3 ? lbase + (int)(fb::W2T * 4) + ((f + 1) * 16 + wave * 2) * 16 * 1024 : w1off;
                if (f == 0) bwd_stamp(a.trace, 4);     // first chunk: rows + bias partials out
                gemm_phase_h<2, 16, 0, true>(acc_x, acc_xt, Hc + am(LDX), Hc + at(LDX), LDX, rsrc, voff, w1off, 64 * 1024, g_f, nxt,
                                             f < 3 ? 16 * 1024 : 64 * 1024);
                if (f == 0) bwd_stamp(a.trace, 5);     // first chunk: dx1 partial product done
            }
            __syncthreads();
        }
        bwd_stamp(a.trace, 6);
        // ---- dx1 = dz2 + dpre W1 -------------------------------------------------------------------------------------------
#pragma unroll
        for (int n = 0; n < 2; ++n) {   // dz2 values read before the first write-back (tip_pgemm.h: `G[i] += v` serialises read -> wait -> write)
            const int col = (wave * 2 + n) * 16 + l15;
            float gr_[RBM][4], gt_[RBT];
#pragma unroll
            for (int r = 0; r < RBM; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) gr_[r][e] = G[(r * 16 + lg * 4 + e) * LDX + col];
#pragma unroll
            for (int rb = 0; rb < RBT; ++rb) gt_[rb] = G[(TAIL0 + 4 * rb + lg) * LDX + col];
#pragma unroll
            for (int r = 0; r < RBM; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) G[(r * 16 + lg * 4 + e) * LDX + col] = gr_[r][e] + acc_x[r][n][e];
#pragma unroll
            for (int rb = 0; rb < RBT; ++rb) G[(TAIL0 + 4 * rb + lg) * LDX + col] = gt_[rb] + tail_reduce(acc_xt[rb][n], lg);
        }
        __syncthreads();
        rows_to_hbm_fixed<D>(G, LDX, a.dx1 + grow0 * D, D, T, tid);
        bwd_stamp(a.trace, 7);
        __syncthreads();
    }
}

hipError_t launch_ffn_bwd(const Dims& d, const FfnBwdArgs& a, int B, int T, int num_cus, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (!fused_supported(d, T) || (long long)B * T * d.F * 4 > 0x7fffffffLL) return hipErrorInvalidValue;
    constexpr int lds = (3 * fz::RP * fz::LDX + fz::RP * kGateLd) * 4;
    static_assert(lds <= 160 * 1024, "LDS of one CU");
    static PerDeviceFlag attr_flag; bool& attr_set = attr_flag.cur();
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    FfnBwdArgs aa = a;
    aa.hid_bytes = (int)((long long)B * T * d.F * 4);
    static int trace = -1;
    if (trace < 0) trace = tip_env("TIP_BWD_TRACE") ? 1 : 0;
    aa.trace = trace;
    if (aa.gbits) hipLaunchKernelGGL(ffn_bwd_kernel<true>, dim3(B < num_cus ? B : num_cus), dim3(fz::THREADS), lds, s, aa, B, T);
    else hipLaunchKernelGGL(ffn_bwd_kernel<false>, dim3(B < num_cus ? B : num_cus), dim3(fz::THREADS), lds, s, aa, B, T);
    return hipGetLastError();
}

// =====================================================================================================================
// Training step, backward of one encoder layer's ATTENTION block, fused per window:
//   dx1 = dL/d(LayerNorm1 output)  ->  LayerNorm1 backward  ->  dz1 (to HBM: it is also the residual path)
//   datt_o = dz1 * keep1 (gradient into out_proj's output)  ->  dO = datt_o Wo, per head, in registers, in BOTH layouts
//   attention backward per head in registers (as tip_attn.hip, mattn_bwd_kernel) -> dq | dk | dv planes in LDS (+ HBM)
//   dx_in = dz1 + [dq | dk | dv] W_qkv
// One wave owns one head end to end; 2 chunks of 8 heads as in the forward.  datt_o and dqkv also go to HBM for the
// weight-gradient GEMMs.
// =====================================================================================================================

// DROP: attention / out-proj dropout on (a.thresh != 0).  A template parameter, and the masked elements below are SELECTS, not branches:
// with `if (valid) { ... }` around each element (and a uniform `if (a.thresh)` inside it) the attention part was 450 basic blocks,
// MFMA -> branch -> exp -> MFMA strictly in program order, nothing for the scheduler to overlap (the heads ran at 50 % of their MFMA time).
template <bool DROP>
__global__ __launch_bounds__(fz::THREADS) void attn_bwd_kernel(AttnBwdArgs a, int B, int T) {
    using namespace fz;
    constexpr int LDT = 20;                             // per-wave transposition tiles [48][16 + 4]
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Mb = smem;                                   // datt_o [48][260]
    float* Sc = smem + RP * LDX;                        // per-wave scratch: Q tile | K tile (LayerNorm partials before that)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wimg), 0, a.wbytes, 0x00020000);
    const int voff = lane * 16;
    const int lbase = (int)(((size_t)a.layer * fb::LAYER_FLOATS2) * 4);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    float* Qs = Sc + wave * (3 * RP * LDT);             // this wave's head: Q rows [48][16] (row-major) ...
    float* Ks = Qs + RP * LDT;                          // ... and K rows: read back as (row 4*lg + e, channel l15) B fragments
    float* Gs = Ks + RP * LDT;                          // dO tile: written in the accumulator layout, read back as row fragments
    float* Ts = Gs;                                     // ... and, once it has been read, the P | dS transposition tiles [2][16][20]
    const unsigned dkey0 = tip_drop_key_s(a.seed, a.site0), dkey1 = tip_drop_key_s(a.seed, a.site1);

    for (int win = blockIdx.x; win < B; win += gridDim.x) {
        const size_t grow0 = (size_t)win * T;
        bwd_stamp(a.trace, 8);
        WRing<1> g_o;   // Wo^T fragments of head c*8 + wave
        ring_prefetch<1>(g_o, rsrc, voff, lbase + (int)(fb::WOT * 4) + wave * 16 * 1024, 0);
        // ---- LayerNorm1 backward --------------------------------------------------------------------------------------------
        {
            const float4 gg = *reinterpret_cast<const float4*>(a.g1 + lane * 4);
            float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg, dm = dg;
            // all rows' operands first (ffn_bwd_kernel: the stores of row i otherwise fence the loads of row i + 1)
            constexpr int NR = RP / 8;
            float4 zq[NR], dq_[NR];
            float2 sq_[NR];
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int r = wave + 8 * i;
                const size_t gr = grow0 + (r < T ? r : 0);
                zq[i] = *reinterpret_cast<const float4*>(a.z1 + gr * D + lane * 4);
                dq_[i] = *reinterpret_cast<const float4*>(a.dx1 + gr * D + lane * 4);
                sq_[i] = *reinterpret_cast<const float2*>(a.st1 + gr * 2);
            }
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int r = wave + 8 * i;
                float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
                if (r < T) {
                    const size_t gr = grow0 + r;
                    const float4 zv = zq[i];
                    const float4 dyv = dq_[i];
                    const float mean = sq_[i].x, rstd = sq_[i].y;
                    float4 xh;
                    xh.x = (zv.x - mean) * rstd; xh.y = (zv.y - mean) * rstd; xh.z = (zv.z - mean) * rstd; xh.w = (zv.w - mean) * rstd;
                    const float ax = dyv.x * gg.x, ay = dyv.y * gg.y, az = dyv.z * gg.z, aw = dyv.w * gg.w;
                    const float m1 = wsum((ax + ay) + (az + aw)) * (1.f / D);
                    const float m2 = wsum((ax * xh.x + ay * xh.y) + (az * xh.z + aw * xh.w)) * (1.f / D);
                    dg.x += dyv.x * xh.x; dg.y += dyv.y * xh.y; dg.z += dyv.z * xh.z; dg.w += dyv.w * xh.w;
                    db.x += dyv.x; db.y += dyv.y; db.z += dyv.z; db.w += dyv.w;
                    float4 o;
                    o.x = rstd * (ax - m1 - xh.x * m2); o.y = rstd * (ay - m1 - xh.y * m2);
                    o.z = rstd * (az - m1 - xh.z * m2); o.w = rstd * (aw - m1 - xh.w * m2);
                    *reinterpret_cast<float4*>(a.dz1 + gr * D + lane * 4) = o;
                    m = o;
                    if (DROP) {
                        const unsigned long long idx = gr * D + lane * 4;
                        m.x = tip_drop_hash_k(dkey1, idx) >= a.thresh ? o.x * a.scale : 0.f;
                        m.y = tip_drop_hash_k(dkey1, idx + 1) >= a.thresh ? o.y * a.scale : 0.f;
                        m.z = tip_drop_hash_k(dkey1, idx + 2) >= a.thresh ? o.z * a.scale : 0.f;
                        m.w = tip_drop_hash_k(dkey1, idx + 3) >= a.thresh ? o.w * a.scale : 0.f;
                    }
                    *reinterpret_cast<float4*>(a.datt_o + gr * D + lane * 4) = m;
                    dm.x += m.x; dm.y += m.y; dm.z += m.z; dm.w += m.w;
                }
                *reinterpret_cast<float4*>(Mb + r * LDX + lane * 4) = m;      // padded rows: zeros
            }
            float* red = Sc;   // [8 waves][3][D] = 6144 floats
            *reinterpret_cast<float4*>(red + (wave * 3 + 0) * D + lane * 4) = dg;
            *reinterpret_cast<float4*>(red + (wave * 3 + 1) * D + lane * 4) = db;
            *reinterpret_cast<float4*>(red + (wave * 3 + 2) * D + lane * 4) = dm;
        }
        __syncthreads();
        for (int i = tid; i < 3 * D; i += THREADS) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += Sc[w * 3 * D + i];
            a.lnpart[(size_t)win * 6 * D + i] = s;
        }
        __syncthreads();

        bwd_stamp(a.trace, 9);
        // ---- per head (no workgroup barrier in here: a wave only touches Mb (read), its own scratch and its own HBM columns) ---
#pragma unroll 1
        for (int c = 0; c < 2; ++c) {
            bwd_stamp(a.trace, 10 + c);
            const int head = c * 8 + wave;
            const unsigned long long bh = (unsigned long long)win * H + head;
            const __amdgpu_buffer_rsrc_t q_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.qkv) + grow0 * (3 * D), 0, T * 3 * D * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.att) + grow0 * D, 0, T * D * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t st_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.ast) + bh * T * 2, 0, T * 8, 0x00020000);
            // this window's dq | dk | dv rows as a buffer whose extent is T rows: stores to the padded rows are dropped by the
            // hardware's range check instead of by a branch around every store (each one ended a basic block)
            const __amdgpu_buffer_rsrc_t dq_rs = __builtin_amdgcn_make_buffer_rsrc(a.dqkv + grow0 * (3 * D), 0, T * 3 * D * 4, 0x00020000);
            const int dq_col = (head * 16 + l15) * 4;
            // q, k, v, O row fragments and the softmax statistics are requested before the dO product
            f32x4 qf[RB], kf[RB], vf[RB], of[RB];
            float mq[RB], iq[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int row = r * 16 + l15;
                // rows >= T: out of the buffers' extent -> zeros, no branch
                const int qo = row * (3 * D * 4) + head * 64 + lg * 16;
                qf[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(q_rs, qo, 0, 0));
                kf[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(q_rs, qo + D * 4, 0, 0));
                vf[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(q_rs, qo + 2 * D * 4, 0, 0));
                of[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(o_rs, row * (D * 4) + head * 64 + lg * 16, 0, 0));
                typedef float tf2 __attribute__((ext_vector_type(2)));
                const tf2 mi = __builtin_bit_cast(tf2, __builtin_amdgcn_raw_buffer_load_b64(st_rs, row * 8, 0, 0));
                mq[r] = mi[0];
                iq[r] = mi[1];
            }
            // dO of this head: [48 x 16] = datt_o [48 x 256] * Wo^T(:, head): gp = accumulator layout (rows 4*lg + e, channel l15);
            // gs = the same tile as row fragments (row l15, channels 4*lg + e), obtained through the wave's scratch
            f32x4 gs[RB], gp[RB];
            {
                f32x4 acc[RB][1];
                zero_acc<1>(acc);
                const int osoff = lbase + (int)(fb::WOT * 4) + head * 16 * 1024;
                const int nxt = c == 0 ? lbase + (int)(fb::WOT * 4) + (8 + wave) * 16 * 1024 : osoff;
                gemm_phase<1, 16>(acc, Mb + l15 * LDX + lg * 4, LDX, rsrc, voff, osoff, 0, g_o, nxt, 0);
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    gp[r] = acc[r][0];
#pragma unroll
                    for (int e = 0; e < 4; ++e) Gs[(r * 16 + lg * 4 + e) * LDT + l15] = gp[r][e];
                }
            }
            // Q and K rows through the wave's scratch: written as row fragments, read back transposed (row 4*lg + e, channel l15)
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                *reinterpret_cast<f32x4*>(Qs + (r * 16 + l15) * LDT + lg * 4) = qf[r];
                *reinterpret_cast<f32x4*>(Ks + (r * 16 + l15) * LDT + lg * 4) = kf[r];
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < RB; ++r) gs[r] = *reinterpret_cast<const f32x4*>(Gs + (r * 16 + l15) * LDT + lg * 4);
            float dd[RB];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                float d_ = (gs[r][0] * of[r][0] + gs[r][1] * of[r][1]) + (gs[r][2] * of[r][2] + gs[r][3] * of[r][3]);
                d_ = lg4_sum(d_);
                dd[r] = d_;
            }
            f32x4 dk[RB], dv[RB];
            float sq = 0.f;   // this lane's share of the in-projection bias gradient (column sums of dq)
#pragma unroll
            for (int r = 0; r < RB; ++r) dk[r] = dv[r] = zero4;
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int q = r * 16 + l15;
                f32x4 dq = zero4;
#pragma unroll
                for (int cb = 0; cb <= r; ++cb) {
                    // (key 4*lg + e, query l15): scores and dP once, on the matrix pipe
                    f32x4 s1 = zero4, p1 = zero4;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[cb][e], qf[r][e], s1, 0, 0, 0);
                        p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[cb][e], gs[r][e], p1, 0, 0, 0);
                    }
                    f32x4 pd, ds;   // dropped probabilities, dS: the operands of dV / dK in the transposed layout
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int kk = cb * 16 + lg * 4 + e;
                        // masked (kk > q) or padded (q >= T) elements: the select sits on the exponent (exp(-inf) = 0 -> p = 0 -> v = 0
                        // exactly), where both sides are cheap; a select on the RESULT is turned back into a branch around the exp
                        const float p = __expf((kk <= q && q < T) ? s1[e] * a.q_scale - mq[r] : -INFINITY) * iq[r];
                        float kf_ = 1.f;
                        if (DROP) kf_ = tip_drop_hash_k(dkey0, (bh * T + q) * T + kk) >= a.thresh ? a.scale : 0.f;
                        pd[e] = p * kf_;
                        ds[e] = p * (p1[e] * kf_ - dd[r]);
                    }
                    // The same two tiles with the roles of the lane and the register swapped — (query 4*lg + e, key l15), what dV and
                    // dK reduce over — through the wave's scratch (the dO tile's, free since gs was read): 8 LDS writes + 2 reads.
                    // Vector-ALU work does not issue under fp32 MFMAs on this part (tools/probes/coissue_probe.hip), LDS traffic
                    // does: until round 5 this layout was COMPUTED a second time (8 MFMAs, 4 exponentials, 4 dropout hashes and 12
                    // cross-lane reads per block pair), and the heads ran at 52 % of their MFMA time.
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        Ts[(lg * 4 + e) * LDT + l15] = pd[e];
                        Ts[(16 + lg * 4 + e) * LDT + l15] = ds[e];
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        dq = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[e], Ks[(cb * 16 + lg * 4 + e) * LDT + l15], dq, 0, 0, 0);
                    __builtin_amdgcn_wave_barrier();
                    const f32x4 pdt = *reinterpret_cast<const f32x4*>(Ts + l15 * LDT + lg * 4);
                    const f32x4 dst = *reinterpret_cast<const f32x4*>(Ts + (16 + l15) * LDT + lg * 4);
                    __builtin_amdgcn_wave_barrier();   // (the next block pair's writes stay behind these reads)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int qq = r * 16 + lg * 4 + e;
                        dv[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(pdt[e], gp[r][e], dv[cb], 0, 0, 0);
                        dk[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(dst[e], Qs[qq * LDT + l15], dk[cb], 0, 0, 0);
                    }
                }
                // dQ rows of block r: (queries 4*lg + e, channel l15)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int qq = r * 16 + lg * 4 + e;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dq[e] * a.q_scale), dq_rs, qq * (3 * D * 4) + dq_col, 0, 0);
                    sq += dq[e] * a.q_scale;   // rows >= T are exactly zero (their dS is masked)
                }
            }
            float sk = 0.f, sv = 0.f;
#pragma unroll
            for (int cb = 0; cb < RB; ++cb)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kk = cb * 16 + lg * 4 + e;
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dk[cb][e] * a.q_scale), dq_rs, kk * (3 * D * 4) + D * 4 + dq_col, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(dv[cb][e]), dq_rs, kk * (3 * D * 4) + 2 * D * 4 + dq_col, 0, 0);
                    sk += dk[cb][e] * a.q_scale;
                    sv += dv[cb][e];
                }
            sq = lg4_sum(sq);
            sk = lg4_sum(sk);
            sv = lg4_sum(sv);
            if (lg == 0) {
                float* lp = a.lnpart + (size_t)win * 6 * D + 3 * D + head * 16 + l15;
                lp[0] = sq;
                lp[D] = sk;
                lp[2 * D] = sv;
            }
            __builtin_amdgcn_wave_barrier();   // the scratch tiles are rewritten by the next head
        }
        bwd_stamp(a.trace, 12);
        __syncthreads();   // every head's dq | dk | dv rows of this window are in HBM/L2 (and visible to the other waves)
        bwd_stamp(a.trace, 13);
        // ---- dx_in = dz1 + dqkv [48 x 768] * Wqkv^T -------------------------------------------------------------------------------
        // The dq | dk | dv rows just written (L2) are staged through LDS in 6 chunks of 128 columns, double-buffered in the
        // per-wave scratch region (free by now): coalesced 512-byte row segments once per workgroup instead of every wave
        // gathering 64-byte pieces of all 48 rows per k-block (that form ran this product at 55 % of its MFMA time).
        {
            constexpr int KC = 128, LDA = 136, NCH = 3 * D / KC;
            static_assert(2 * RP * LDA <= 8 * 3 * RP * 20, "chunk buffers fit the scratch region");
            // hybrid row tiling (fused_encoder_h_kernel): rows 0-31 on 16x16x4, rows 32-39 on 4x4x1 MFMAs, nothing on the pad rows
            using namespace fzh;
            // The accumulators START at dz1 (the residual path of dx_in = dz1 + dqkv Wqkv^T) instead of at zero, so that the epilogue
            // only stores: written as `dx_in[i] = dz1[i] + acc` per element it compiled to one dependent load -> wait -> store round
            // trip per element, 20 per lane at the end of every window (the optimiser may not move a load across a store it cannot
            // prove disjoint, tip_pgemm.h), and this kernel has no registers left to batch the loads in.  Tail blocks: lane group 0
            // carries dz1 in its k-partial, the other three start at zero (tail_reduce sums the four).
            f32x4 acc_i[RBM][2], acc_it[RBT][2];
            const __amdgpu_buffer_rsrc_t z_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dz1) + grow0 * D, 0, T * D * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(a.dx_in + grow0 * D, 0, T * D * 4, 0x00020000);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int col = (wave * 2 + n) * 16 + l15;
#pragma unroll
                for (int r = 0; r < RBM; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = r * 16 + lg * 4 + e;      // rows >= T: outside the buffer -> 0
                        acc_i[r][n][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(z_rs, (row * D + col) * 4, 0, 0));
                    }
#pragma unroll
                for (int rb = 0; rb < RBT; ++rb)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = TAIL0 + 4 * rb + e;
                        const float zv = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(z_rs, (row * D + col) * 4, 0, 0));
                        acc_it[rb][n][e] = lg == 0 ? zv : 0.f;
                    }
            }
            float* Ab = Sc;
            const int wq = lbase + (int)(fb::WQT * 4) + (wave * 2) * 48 * 1024;
            WRing<2> g_q;
            ring_prefetch<2>(g_q, rsrc, voff, wq, 48 * 1024);
            f32x4 st[3];   // this thread's 3 float4 of a chunk: element i = tid + 512 j -> row i / 32, columns 4 (i % 32) ..
            auto fetch = [&](int ch) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int i = tid + THREADS * j, row = i >> 5, c4 = i & 31;
                    st[j] = zero4;
                    if (row < T) st[j] = *reinterpret_cast<const f32x4*>(a.dqkv + (grow0 + row) * (3 * D) + ch * KC + c4 * 4);
                }
            };
            auto stash = [&](float* buf) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int i = tid + THREADS * j, row = i >> 5, c4 = i & 31;
                    *reinterpret_cast<f32x4*>(buf + row * LDA + c4 * 4) = st[j];
                }
            };
            fetch(0);
            stash(Ab);
            __syncthreads();
#pragma unroll 1
            for (int ch = 0; ch < NCH; ++ch) {
                if (ch + 1 < NCH) fetch(ch + 1);           // in flight during this chunk's MFMAs
                const int nx = ch + 1 < NCH ? wq + (ch + 1) * (KC / 16) * 1024 : wq;
                const float* Ac = Ab + (ch & 1) * RP * LDA;
                gemm_phase_h<2, KC / 16, 0, true>(acc_i, acc_it, Ac + l15 * LDA + lg * 4, Ac + (TAIL0 + (lane & 3)) * LDA + lg * 4, LDA, rsrc, voff,
                                                  wq + ch * (KC / 16) * 1024, 48 * 1024, g_q, nx, 48 * 1024);
                if (ch + 1 < NCH) stash(Ab + ((ch + 1) & 1) * RP * LDA);
                __syncthreads();
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int col = (wave * 2 + n) * 16 + l15;
#pragma unroll
                for (int r = 0; r < RBM; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int row = r * 16 + lg * 4 + e;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc_i[r][n][e]), x_rs, (row * D + col) * 4, 0, 0);
                    }
#pragma unroll
                for (int rb = 0; rb < RBT; ++rb) {
                    const int row = TAIL0 + 4 * rb + lg;
                    const float v = tail_reduce(acc_it[rb][n], lg);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), x_rs, (row * D + col) * 4, 0, 0);
                }
            }
        }
        bwd_stamp(a.trace, 14);
        __syncthreads();
    }
}

hipError_t launch_attn_bwd(const Dims& d, const AttnBwdArgs& a, int B, int T, int num_cus, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (!fused_supported(d, T)) return hipErrorInvalidValue;
    constexpr int lds = (fz::RP * fz::LDX + 8 * 3 * fz::RP * 20) * 4;   // datt_o + per-wave Q / K / dO transposition tiles
    static_assert(8 * 3 * fz::RP * 20 >= 8 * 3 * fz::D, "the LayerNorm partial scratch lives in the per-wave scratch region");
    if ((long long)B * T * 3 * d.D * 4 > 0x7fffffffLL) return hipErrorInvalidValue;
    static PerDeviceFlag attr_flag; bool& attr_set = attr_flag.cur();
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    AttnBwdArgs aa = a;
    aa.dqkv_bytes = (int)((long long)B * T * 3 * d.D * 4);
    static int trace = -1;
    if (trace < 0) trace = tip_env("TIP_BWD_TRACE") ? 1 : 0;
    aa.trace = trace;
    if (aa.thresh) hipLaunchKernelGGL(attn_bwd_kernel<true>, dim3(B < num_cus ? B : num_cus), dim3(fz::THREADS), lds, s, aa, B, T);
    else hipLaunchKernelGGL(attn_bwd_kernel<false>, dim3(B < num_cus ? B : num_cus), dim3(fz::THREADS), lds, s, aa, B, T);
    return hipGetLastError();
}

// =====================================================================================================================
// Training step, the dX products outside the encoder layers, per window:  out[T][N] = A[T][K] W'^T  with W' [N][16 KB] in
// fragment order.  These were batch-wide LDS-tiled GEMMs (tgemm16: dH = dy W_out 29.7 us, d_enc = delta W_ih 36.6 us at B = 256 —
// 46 / 73 TFLOP/s: 10 240 rows x 256-512 columns is 640-1 280 tiles of 64 x 64 on 256 CUs with both operands staged per tile);
// here a workgroup takes one window: its 40 rows go to LDS once (whole K), each wave owns NBW column blocks and streams their
// weight fragments through the register ring of the fused kernels, rows 0-31 on 16x16x4 and rows 32-39 on 4x4x1 MFMAs
// (gemm_phase_h) — nothing on pad rows, no per-tile operand staging.
//   <2, 32>: N = 256, K = 512 (d_enc);   <4, 10>: N = 512, K = 160 (dH: dy padded to 144 columns, the k-blocks come in pairs)
// =====================================================================================================================
template <int NBW, int KB>
__global__ __launch_bounds__(fz::THREADS) void win_gemm_kernel(WinGemmArgs a, int B, int T) {
    using namespace fz;
    using namespace fzh;
    constexpr int K = KB * 16, LDA = K + 8, C4 = K / 4, NI = (RP * C4 + THREADS - 1) / THREADS;
    extern __shared__ __attribute__((aligned(16))) float smem[];   // A rows [RP][LDA]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wfrag), 0, a.wbytes, 0x00020000);
    const int voff = lane * 16;
    const int wsoff = wave * NBW * KB * 1024;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int win = blockIdx.x; win < B; win += gridDim.x) {
        WRing<NBW> g;
        ring_prefetch<NBW>(g, rsrc, voff, wsoff, KB * 1024);
        // the window's rows -> LDS: every load in flight before the first store (clamped addresses, zeros afterwards)
        f32x4 v[NI];
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int i = tid + j * THREADS, row = i / C4, c = (i - row * C4) * 4;
            const int rc = row < T ? row : T - 1, cc = c < a.kvalid ? c : a.kvalid - 4;
            v[j] = *reinterpret_cast<const f32x4*>(a.A + ((size_t)win * T + rc) * a.lda + cc);
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int i = tid + j * THREADS, row = i / C4, c = (i - row * C4) * 4;
            if (i < RP * C4) *reinterpret_cast<f32x4*>(smem + row * LDA + c) = (row < T && c < a.kvalid) ? v[j] : zero4;
        }
        __syncthreads();
        f32x4 acc[RBM][NBW], acct[RBT][NBW];
        zero_acc_h<NBW>(acc, acct);
        // rows 0-31 with swapped operands (transposed accumulators: lane = row l15, columns 4 lg ..): one 16-byte store per tile instead
        // of four 4-byte ones; the window's output rows as a buffer of T rows, so the pad rows are dropped by the range check instead
        // of by a branch around every store (as the hybrid encoder's IH epilogue).  Same products, same sums.
        gemm_phase_h<NBW, KB, NBW, true>(acc, acct, smem + l15 * LDA + lg * 4, smem + (TAIL0 + (lane & 3)) * LDA + lg * 4, LDA, rsrc, voff, wsoff,
                                         KB * 1024, g, wsoff, KB * 1024);
        const __amdgpu_buffer_rsrc_t o_rs = tip_rows_buffer(a.out + (size_t)win * T * a.ldo, T * a.ldo * 4);
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
#pragma unroll
            for (int r = 0; r < RBM; ++r)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[r][n]), o_rs,
                                                       ((r * 16 + l15) * a.ldo + (wave * NBW + n) * 16 + lg * 4) * 4, 0, 0);
#pragma unroll
            for (int rb = 0; rb < RBT; ++rb)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(tail_reduce(acct[rb][n], lg)), o_rs,
                                                      ((TAIL0 + 4 * rb + lg) * a.ldo + (wave * NBW + n) * 16 + l15) * 4, 0, 0);
        }
        __syncthreads();   // the next window's rows overwrite the LDS image
    }
}

template <int NBW, int KB>
static hipError_t launch_win_gemm_t(const WinGemmArgs& a, int B, int T, int num_cus, hipStream_t s) {
    constexpr int lds = fz::RP * (KB * 16 + 8) * 4;
    static PerDeviceFlag attr_flag; bool& attr_set = attr_flag.cur();
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(win_gemm_kernel<NBW, KB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((win_gemm_kernel<NBW, KB>), dim3(B < num_cus ? B : num_cus), dim3(fz::THREADS), lds, s, a, B, T);
    return hipGetLastError();
}

// N = 16 * 8 * nbw columns, K = 16 * kb (weights [N][K] in fragment order, zero padded); hipErrorInvalidValue for other shapes
hipError_t launch_win_gemm(int N, int K, const WinGemmArgs& a, int B, int T, int num_cus, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (T < 1 || T > fz::TMAX || a.kvalid % 4 || a.kvalid < 4 || a.kvalid > K || a.lda % 4 || a.ldo % 4 || (reinterpret_cast<uintptr_t>(a.out) & 15) || (long long)T * a.ldo * 4 > 0x7fffffffLL || (long long)N * K * 4 != (long long)a.wbytes)
        return hipErrorInvalidValue;
    if (N == 256 && K == 512) return launch_win_gemm_t<2, 32>(a, B, T, num_cus, s);
    if (N == 512 && K == 160) return launch_win_gemm_t<4, 10>(a, B, T, num_cus, s);
    return hipErrorInvalidValue;
}

}  // namespace tip

extern "C" int tip_debug_read_bwd_trace(unsigned long long* out, int n) {
    if (!out || n < 0 || n > 16) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tip::g_bwd_trace), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -5;
}

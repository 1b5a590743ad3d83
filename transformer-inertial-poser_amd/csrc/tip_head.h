// tip_head.h — the register-resident output projection (see tip_head.hip for the design) as a device function shared by the
// stand-alone kernel (tip_head.hip) and the recurrence's fused epilogue (rnn_rows4_kernel, tip_general.hip).
#pragma once
#include "tip_internal.h"
#include "tip_layernorm.h"

namespace tip {

namespace hd {
constexpr int THREADS = 512, WAVES = 8, NB = 9, KBW = 4;   // K = 512 = 32 k-blocks of 16, four per wave
constexpr int SLOT = 64 * 4;                               // floats of one (wave, column block) fragment image
constexpr int PART_FLOATS = WAVES * NB * SLOT;             // one buffer of partials: 18 432 floats = 73 728 B
constexpr int C8_FLOATS = 48 * WAVES;                      // column block 8 (3 valid columns), compact: [value][wave]
constexpr int LDS_BYTES = 2 * (PART_FLOATS + C8_FLOATS) * 4;   // 150 528 B: one workgroup per CU
}  // namespace hd

typedef float hf4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ hf4 hd_load(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(hf4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// four k-partials (one per lg) of a 4-row tile -> lane (lg, l15) = row lg, column l15 (tail_reduce of tip_fused.hip)
__device__ __forceinline__ float hd_tail_reduce(const hf4& v) {
    float a = v[0], b = v[2];
    swap32(a, b);
    float k0 = a + b;
    a = v[1], b = v[3];
    swap32(a, b);
    float k1 = a + b;
    swap16(k0, k1);
    return k0 + k1;
}

// measurement only (TIP_HEAD_TRACE=1): s_memtime stamps of wave 0 of workgroups 0 and 100 — [16 wg + 0] entry, [1] weight loads
// issued, [2 + i] after the barrier of tile i (i < 12), [15] exit
extern __device__ unsigned long long g_hd_trace[32];
extern __device__ unsigned long long g_hd_wg[2 * 1024];   // [wg][entry, exit] in s_memrealtime ticks (100 MHz, device-wide counter)
#define HD_STAMP(slot) do { if (TRACE && (blockIdx.x == 0 || blockIdx.x == 100) && threadIdx.x == 0) \
        g_hd_trace[(blockIdx.x ? 16 : 0) + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)

// The whole kernel as a device function, so that rnn_rows4_kernel (tip_general.hip) can run it as its epilogue.  `ctl` says which
// row groups this workgroup takes — first(), then + stride() while < ngroups — and gets one call, after_weights(), once the weight
// loads and the first tile's A loads are in flight (the fused caller waits there for the last row of its window).  AUX = cache
// policy of the A loads (16 = sc1 when the rows were written by other workgroups of the SAME launch).  The dynamic LDS region
// (hd::LDS_BYTES from its start) is the partial-sum buffer.  (The group range is a functor, not two integers: with `int g0, int
// gstride` parameters — even unused ones — the register allocator, which sits at the 256-VGPR limit here, spills 64 B per lane into
// the tile loop instead of 8.)
struct HdGridStride {
    __device__ __forceinline__ int first() const { return blockIdx.x; }
    __device__ __forceinline__ int stride() const { return gridDim.x; }
    __device__ __forceinline__ void after_weights() const {}
};
template <int MODE, bool TRACE, int AUX, class Ctl>
__device__ __forceinline__ void head_ksplit_body(const float* __restrict__ A, unsigned lda_b, unsigned a_bytes,
                                                 const float* __restrict__ wfrag, const float* __restrict__ bias, float* __restrict__ Y,
                                                 int ldy, int M, int N, int ngroups, Ctl ctl) {
    using namespace hd;
    extern __shared__ float part[];   // hd::LDS_BYTES of dynamic LDS (the caller's dynamic region, from its start)
    constexpr int NPH = MODE == 0 ? 4 : 1;
    constexpr int RG = MODE == 0 ? 40 : 4;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wfrag), 0, NB * 32 * 1024, 0x00020000);
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(Y, 0, (unsigned)M * (unsigned)ldy * 4u, 0x00020000);

    // per-lane byte offsets of the A fragments inside a tile (rows past M fall outside the descriptor: loads return 0)
    const unsigned aoff_b = (unsigned)l15 * lda_b + (unsigned)(wave * KBW * 64 + lg * 16);          // 16-row tile: row l15
    const unsigned aoff_t = (unsigned)(lane & 3) * lda_b + (unsigned)(wave * KBW * 64 + lg * 16);   // 4-row tile: row lane & 3

    auto tile_is_tail = [](int p) { return MODE != 0 || p >= 2; };
    auto tile_row = [](int p) { return MODE == 0 ? (p < 2 ? 16 * p : 32 + 4 * (p - 2)) : 0; };

    hf4 acur[KBW], anext[KBW];
    auto load_a = [&](hf4 (&a)[KBW], int g, int p) {
        const unsigned base = (unsigned)(g * RG + tile_row(p)) * lda_b + (tile_is_tail(p) ? aoff_t : aoff_b);
#pragma unroll
        for (int j = 0; j < KBW; ++j) a[j] = __builtin_bit_cast(hf4, __builtin_amdgcn_raw_buffer_load_b128(ars, (int)(base + j * 64), 0, AUX));
    };

    int g = ctl.first();
    if (g >= ngroups) return;
    HD_STAMP(0);
    if (TRACE && threadIdx.x == 0 && blockIdx.x < 1024) g_hd_wg[2 * blockIdx.x] = __builtin_amdgcn_s_memrealtime();
    int nstamp = 2;
    load_a(acur, g, 0);
    // the wave's slice of the weight, in the order the MFMAs consume it (k-block major)
    hf4 wreg[NB][KBW];
#pragma unroll
    for (int j = 0; j < KBW; ++j)
#pragma unroll
        for (int n = 0; n < NB; ++n) wreg[n][j] = hd_load(wrs, lane * 16, (n * 32 + wave * KBW + j) * 1024);

    HD_STAMP(1);
    ctl.after_weights();
    // column block a wave reduces: its own index; block 8 (columns 128..) is the last wave's second job
    const float bias_w = (wave * 16 + l15) < N ? bias[wave * 16 + l15] : 0.f;

    // ---- reduction of a finished tile, in two halves so that it can sit INSIDE the next tile's MFMA stream: the LDS reads are
    // issued after the first k-block's MFMAs, the adds and stores after the second's (the matrix pipe keeps running) ----
    auto red_addr = [&](int row, int n) -> unsigned {
        const int col = n * 16 + l15;
        return (unsigned)row * (unsigned)ldy * 4u + (col < N ? (unsigned)col * 4u : 0x80000000u);   // bad column: out of range
    };
    // half h (0 / 1) of the eight partials: waves 4 h .. 4 h + 3, summed (a + b) + (c + d); the two halves are added last, so a
    // tile's sum is ((w0 + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7)) whatever the interleaving
    auto half_load = [&](bool tail, hf4 (&sv)[4], int buf, int n, int h) {
        if (tail) {
            const float* src = part + buf * PART_FLOATS + n * SLOT + lane + h * 4 * NB * SLOT;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) sv[w2][0] = src[w2 * NB * SLOT];
        } else {
            const hf4* src = reinterpret_cast<const hf4*>(part + buf * PART_FLOATS) + n * 64 + lane + h * 4 * NB * 64;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) sv[w2] = src[w2 * NB * 64];
        }
    };
    auto half_sum = [&](bool tail, const hf4 (&sv)[4]) -> hf4 {
        if (tail) {
            hf4 r = (hf4){0.f, 0.f, 0.f, 0.f};
            r[0] = (sv[0][0] + sv[1][0]) + (sv[2][0] + sv[3][0]);
            return r;
        }
        return (sv[0] + sv[1]) + (sv[2] + sv[3]);
    };
    auto red_store = [&](bool tail, const hf4& v, int row0, int n, float bv) {
        if (tail) {
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v[0] + bv), yrs, (int)red_addr(row0 + lg, n), 0, 0);
        } else {
            const unsigned off = red_addr(row0 + lg * 4, n);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v[e] + bv), yrs, (int)(off + (unsigned)(e * ldy) * 4u), 0, 0);
        }
    };

    float* const c8 = part + 2 * PART_FLOATS;
    const float bias_c8 = (128 + lane % 3) < N ? bias[128 + lane % 3] : 0.f;
    auto c8_reduce = [&](bool tail, int buf, int row0) {
        if (lane < (tail ? 12 : 48)) {
            const hf4* src = reinterpret_cast<const hf4*>(c8 + buf * C8_FLOATS + lane * WAVES);
            const hf4 a = src[0], b = src[1];
            const float v = ((a[0] + a[1]) + (a[2] + a[3])) + ((b[0] + b[1]) + (b[2] + b[3]));
            const int col = 128 + lane % 3;
            const unsigned off = (unsigned)(row0 + lane / 3) * (unsigned)ldy * 4u + (col < N ? (unsigned)col * 4u : 0x80000000u);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v + bias_c8), yrs, (int)off, 0, 0);
        }
    };
    int buf = 0;
    int prev_row0 = 0;
    // one tile: p = its index in the group (compile time), has_prev = a finished tile waits in part[buf ^ 1]
    auto phase = [&](auto PC, int gg, bool has_prev) {
        constexpr int p = decltype(PC)::value;
        constexpr bool tail = MODE != 0 || p >= 2;
        constexpr bool prev_tail = MODE != 0 || ((p + NPH - 1) % NPH) >= 2;
        const int row0 = gg * RG + tile_row(p);
        // the buffer index is kept opaque: with its parity known per unrolled tile the compiler hoists one LDS address register
        // per (buffer, access kind) out of the loop — a dozen VGPRs the kernel does not have (it spilled them)
        int bo = buf;
        asm volatile("" : "+s"(bo));
        hf4 acc[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[n] = (hf4){0.f, 0.f, 0.f, 0.f};
        auto mfma_kb = [&](int j) {
#define TIP_HD_STEP(c)                                                                                                   \
    _Pragma("unroll") for (int n = 0; n < NB; ++n) acc[n] = tail                                                         \
        ? __builtin_amdgcn_mfma_f32_4x4x1f32(acur[j].c, wreg[n][j].c, acc[n], 0, 0, 0)                                  \
        : __builtin_amdgcn_mfma_f32_16x16x4f32(acur[j].c, wreg[n][j].c, acc[n], 0, 0, 0);
            TIP_HD_STEP(x) TIP_HD_STEP(y) TIP_HD_STEP(z) TIP_HD_STEP(w)
#undef TIP_HD_STEP
        };
        // the finished tile's reduction rides in this tile's MFMA stream: the wave's own column block in two halves, then (last
        // wave) the three valid columns of block 8 from their compact image — 8 partials of one output per lane, two 16-byte reads
        {
            hf4 sv[4], q0;
            if (has_prev) half_load(prev_tail, sv, bo ^ 1, wave, 0);
            mfma_kb(0);
            if (has_prev) {
                q0 = half_sum(prev_tail, sv);
                half_load(prev_tail, sv, bo ^ 1, wave, 1);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_kb(1);
            if (has_prev) red_store(prev_tail, q0 + half_sum(prev_tail, sv), prev_row0, wave, bias_w);
        }
        if (has_prev && wave == WAVES - 1) c8_reduce(prev_tail, bo ^ 1, prev_row0);
        // A fragments of the next tile (the next group's first tile after the last; past the end: out of range, zeros) — half a
        // tile ahead, not a whole one: issued any earlier they are live together with the reduction's registers and spill
        if (p + 1 < NPH) load_a(anext, gg, p + 1);
        else load_a(anext, gg + ctl.stride(), 0);
        __builtin_amdgcn_sched_barrier(0);
        mfma_kb(2);
        mfma_kb(3);
        float* dst = part + bo * PART_FLOATS + wave * NB * SLOT;
        if (tail) {
            // The permlane swaps sit in inline asm, and the compiler does not insert the MFMA-write -> VALU-read wait states in
            // front of an asm statement (seen: rows lg = 2 wrong and changing from run to run).  This asm takes every
            // accumulator as an in/out operand — so it is ordered after the MFMAs and before the swaps — and supplies the
            // wait states itself (a 2-pass MFMA needs 5; s_nop 15 = 16).
            asm volatile("s_nop 15"
                         : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                           "+v"(acc[7]), "+v"(acc[8]));
#pragma unroll
            for (int n = 0; n < NB - 1; ++n) dst[n * SLOT + lane] = hd_tail_reduce(acc[n]);
            const float t8 = hd_tail_reduce(acc[NB - 1]);
            if (l15 < 3) c8[bo * C8_FLOATS + (lg * 3 + l15) * WAVES + wave] = t8;          // value (row lg, column 128 + l15)
        } else {
#pragma unroll
            for (int n = 0; n < NB - 1; ++n) reinterpret_cast<hf4*>(dst)[n * 64 + lane] = acc[n];
            if (l15 < 3) {
#pragma unroll
                for (int e = 0; e < 4; ++e) c8[bo * C8_FLOATS + ((lg * 4 + e) * 3 + l15) * WAVES + wave] = acc[NB - 1][e];
            }
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);   // nothing of the next tile above the barrier (its accumulators would double the live set)
        if (TRACE) { if (nstamp < 14) HD_STAMP(nstamp); ++nstamp; }
        prev_row0 = row0;
        buf ^= 1;
#pragma unroll
        for (int j = 0; j < KBW; ++j) acur[j] = anext[j];
    };
    using std::integral_constant;
    // the first tile is peeled: its waits are counted against the 36 weight loads still in flight (inside the loop the compiler
    // has to assume the loop-carried state and would drain them all before the first MFMA)
    phase(integral_constant<int, 0>{}, g, false);
#pragma unroll 1
    for (;;) {
        if (NPH == 4) {
            phase(integral_constant<int, 1 % NPH>{}, g, true);
            phase(integral_constant<int, 2 % NPH>{}, g, true);
            phase(integral_constant<int, 3 % NPH>{}, g, true);
        }
        g += ctl.stride();
        if (g >= ngroups) break;
        phase(integral_constant<int, 0>{}, g, true);
    }
    // the last tile
    {
        constexpr bool lt = MODE != 0 || NPH - 1 >= 2;
        hf4 sv[4];
        half_load(lt, sv, buf ^ 1, wave, 0);
        hf4 q = half_sum(lt, sv);
        half_load(lt, sv, buf ^ 1, wave, 1);
        red_store(lt, q + half_sum(lt, sv), prev_row0, wave, bias_w);
        if (wave == WAVES - 1) c8_reduce(lt, buf ^ 1, prev_row0);
    }
    HD_STAMP(15);
    if (TRACE) {
        __syncthreads();
        if (threadIdx.x == 0 && blockIdx.x < 1024) g_hd_wg[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
    }
}


}  // namespace tip

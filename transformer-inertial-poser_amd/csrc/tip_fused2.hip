// tip_fused2.hip — two-window variant of the fused plan (paper configuration), used when there are at least two
// windows per CU (B >= 512 on a 256-CU part).
//
// The one-window kernel (tip_fused.hip) pads T = 40 rows to 48 (3 MFMA row blocks): 17 % of its matrix-core issue
// slots multiply zero rows.  Two windows are 80 rows = exactly 5 row blocks, so here ONE 512-thread workgroup carries
// TWO windows through the encoder with no padding at all, and every weight fragment streamed from L2 is used for
// 5 row blocks instead of 3.  Same packed weight image, same math, same summation order per output element.
//
// LDS (161,024 B): X [80][260] residual stream of both windows | C: per 4-head "quad" Q [96][68], K [96][68],
// V^T [64][100] (window 1 starts at plane row 48 so each window has its own 8 pad rows for the 3x16-row attention
// blocks), reused as U [80][232] (prologue) and Hc [80][136] (FFN hidden chunk of 128).
// Per layer: 4 quads x { Q|K projection (one 16-column block per wave), V projection (head = wave&3, waves 0-3 take
// row blocks 0-2, waves 4-7 row blocks 3-4), attention (wave = head x window), out-projection partial } ->
// LayerNorm1 -> 8 x { linear1 chunk + ReLU, linear2 partial } -> LayerNorm2.
#include <string.h>
#include <algorithm>

#include "tip_internal.h"
#include "tip_attention.h"
#include "tip_layernorm.h"
#include "tip_pgemm.h"

namespace tip {

typedef unsigned int u32x4_f2 __attribute__((ext_vector_type(4)));

namespace f2 {
constexpr int D = 256, F = 1024, R = 512, T = 40, ROWS = 80, RB = 5, KIN = 224;
constexpr int LDX = D + 8;           // 264: row stride = 8 (mod 64) dwords: conflict-free ds_read_b128 A fragments (tip_fused.hip)
constexpr int LDQ = 64 + 4;          // 68: Q / K plane of one quad (4 heads x 16 channels)
constexpr int PROWS = 96;            // plane rows: window w occupies rows 48w .. 48w+39 (+8 pad)
constexpr int LDV = PROWS + 4;       // 100: V^T [64 channels][96 keys]
constexpr int LDU = KIN + 8;         // 232
constexpr int LDH = 128 + 8;         // 136: FFN hidden chunk
constexpr int X_FLOATS = ROWS * LDX;                       // 21120
constexpr int C_FLOATS = 2 * PROWS * LDQ + 64 * LDV;       // 19456
constexpr int LDS_BYTES = (X_FLOATS + C_FLOATS) * 4;       // 162304
constexpr int THREADS = 512;
// packed fused section (identical to tip_fused.hip)
constexpr size_t IN_W = 0, IN_B = IN_W + (size_t)D * KIN, LAYER0 = IN_B + D;
constexpr size_t QKV_W = 0, QKV_B = QKV_W + (size_t)3 * D * D, WO_W = QKV_B + 3 * D, WO_B = WO_W + (size_t)D * D;
constexpr size_t W1_W = WO_B + D, W1_B = W1_W + (size_t)F * D, W2_W = W1_B + F, W2_B = W2_W + (size_t)D * F;
constexpr size_t G1 = W2_B + D, BE1 = G1 + D, G2 = BE1 + D, BE2 = G2 + D, LAYER_FLOATS = BE2 + D;
static_assert(C_FLOATS >= ROWS * LDU && C_FLOATS >= ROWS * LDH, "chunk region must hold U and Hc");
}  // namespace f2

// acc[r][n] += A(rows of block r) * Wblock(n, kb), kb in [0, KB).  lds + aoff[r]: this lane's LDS address of
// (its row of row-block r, k-offset 4*(lane>>4)) — row blocks need not be equally spaced (remapped planes).
// Ring semantics as in tip_fused.hip: k-blocks 0,1 on entry; the last pair of prefetches is redirected to the next
// phase (nsoff, nnstride_b).
template <int NRB, int NBW, int KB, bool SWAP = false>   // SWAP: transposed accumulator tiles (mfma_block2)
__device__ __forceinline__ void gemm_phase2(f32x4 (&acc)[NRB][NBW], const float* lds, const int (&aoff)[NRB],
                                            __amdgpu_buffer_rsrc_t rsrc, int voff, int soff, int nstride_b, WRing2<NBW>& g,
                                            int nsoff, int nnstride_b) {
    static_assert(KB % 2 == 0, "k-blocks are processed in pairs");
    float4 a0[NRB], a1[NRB];
#pragma unroll
    for (int r = 0; r < NRB; ++r) a0[r] = *reinterpret_cast<const float4*>(lds + aoff[r]);
#pragma unroll 1
    for (int kb = 0; kb < KB; kb += 2) {
        const bool last = kb + 2 >= KB;
        const int o = last ? nsoff : soff + (kb + 2) * 1024;
        const int st = last ? nnstride_b : nstride_b;
#pragma unroll
        for (int r = 0; r < NRB; ++r) a1[r] = *reinterpret_cast<const float4*>(lds + aoff[r] + (kb + 1) * 16);
        mfma_block2<NRB, NBW, SWAP>(acc, a0, g.w0);
#pragma unroll
        for (int n = 0; n < NBW; ++n) g.w0[n] = ldfrag2(rsrc, voff, o + n * st);
        // pin the refill here: left alone, the scheduler sinks it behind the second MFMA block, next to its use at the top of the
        // next trip — the two-k-block prefetch then covers two MFMAs instead of forty
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < NRB; ++r) a0[r] = *reinterpret_cast<const float4*>(lds + aoff[r] + (kb + 2) * 16);
        mfma_block2<NRB, NBW, SWAP>(acc, a1, g.w1);
#pragma unroll
        for (int n = 0; n < NBW; ++n) g.w1[n] = ldfrag2(rsrc, voff, o + n * st + 1024);
        __builtin_amdgcn_sched_barrier(0);
    }
}


// A wave-uniform pointer the compiler has lost track of (here: derived from the pair loop's induction variable), forced back into
// SGPRs: a buffer descriptor built from VGPRs makes every buffer instruction a readfirstlane "waterfall" loop.
__device__ __forceinline__ float* uniform_ptr(float* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<float*>(((unsigned long long)hi << 32) | lo);
}

// plane row of X row r: window 1 (rows 40..79) starts at plane row 48
__device__ __forceinline__ int prow(int r) { return r + (r >= f2::T ? 8 : 0); }

// measurement only (TIP_FUSED2_TRACE=1): s_memtime stamps of workgroup 0 / thread 0 at the phase boundaries of layer 1
__device__ unsigned long long g_f2_trace[64];
#define F2_STAMP(slot) do { if (TRACE && blockIdx.x == 0 && tid == 0 && pair == (int)blockIdx.x && layer == 1) g_f2_trace[slot] = __builtin_amdgcn_s_memtime(); } while (0)

// REUSE (exact streaming reuse; tip_forward_reuse): a frame's model inputs never change once recorded
// (real_time_runner_minimal.py:74,85,137: the history buffers are append-only), and its in_linear row and layer-0 Q / K / V rows are
// functions of that one input row alone — the same numbers in each of the 40 windows the frame appears in.  With the encoder's
// dropout and the past-state dropout off they are computed ONCE, when the frame enters (reuse_update_kernel below: the same MFMA
// instruction, packed fragments, k order and bias add as this kernel, so the same bits — every row of this kernel goes through
// v_mfma_f32_16x16x4_f32, whose per-element sum does not depend on the element's position in the tile), and kept per stream in a
// ring of 40 slots [x 256 | q 256 | k 256 | v 256].  This instantiation then runs layer 0 without prologue, in_linear and QKV
// projection: the residual rows and the quads' Q / K / V^T planes come from the ring.  Slot of window row p at frame c: (c + 1 + p)
// mod 40.  Each slot carries the frame index it was written for (the ring's header); a window whose 40 tags are not c - 39 .. c —
// a skipped frame, a ring that was never primed — gets NaN rows, never another frame's numbers.
struct ReuseArgs {
    const float* cache = nullptr;    // ring header (64 floats: 40 frame tags) + [B][40][1024]
    const int* frame_ctr = nullptr;  // device counter holding c (what a captured HIP graph needs), or null: frame_idx
    int frame_idx = 0;
};
namespace ru {
constexpr int SLOT = 1024, HDR = 64, TILE = 16, THREADS = 256, PARTS = 4;
}

// L2 warm-up for ring reads ("touch": one dword per 128-byte line, the value is never used).  The ring is 4 KiB per stream and frame,
// read once per call: every first read is an HBM / Infinity-Cache miss.  Inside a pair the reuse form hides that by requesting a
// quad's planes one attention ahead; the FIRST quad of a workgroup's next pair has no attention in front of it, so its lines are
// touched under the current pair's RNN input projection and the real 16-byte reads then hit this XCD's L2.
// ring: buffer over the pair's rings; slot of row r as in the kernel.
__device__ __forceinline__ unsigned ring_touch_planes(__amdgpu_buffer_rsrc_t ring, int ru_base, int q, int tid) {
    const int i = tid < f2::ROWS * 6 ? tid : f2::ROWS * 6 - 1;      // 80 rows x (q | k | v) x two lines of the quad's 64 channels
    const int r = i / 6, seg = i - r * 6;
    const int w = r >= f2::T ? 1 : 0;
    int sl = ru_base + r - w * f2::T;
    sl -= sl >= f2::T ? f2::T : 0;
    return __builtin_amdgcn_raw_buffer_load_b32(ring, ((w * f2::T + sl) * ru::SLOT + f2::D + (seg >> 1) * f2::D + q * 64 + (seg & 1) * 32) * 4, 0, 0);
}
__device__ __forceinline__ void touch_sink(unsigned v) { asm volatile("" ::"v"(v)); }   // (the wait for a touch sits HERE, long after it)

template <bool TRACE, bool REUSE = false>
__global__ __launch_bounds__(f2::THREADS) void fused_encoder2_kernel(
    const float* __restrict__ wts, const float* __restrict__ x_imu, const float* __restrict__ x_s,
    const float* __restrict__ keep_mask, float keep_scale, float* __restrict__ ih_out, unsigned* __restrict__ hall_sentinel,
    int B, int NI, int S, int L, int wbytes, int ih_off_b, ReuseArgs rua) {
    using namespace f2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* X = smem;
    float* C = smem + X_FLOATS;
    float* Qp = C;
    float* Kp = C + PROWS * LDQ;
    float* Vt = C + 2 * PROWS * LDQ;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wts), 0, wbytes, 0x00020000);
    const int voff = lane * 16;
    const int npairs = (B + 1) / 2;

    // this lane's A-fragment offsets (floats from smem) for an operand with leading dimension ld at region offset base
    auto rows_off = [&](int (&off)[RB], int base, int ld) {
#pragma unroll
        for (int r = 0; r < RB; ++r) off[r] = base + (r * 16 + l15) * ld + lg * 4;
    };
    for (int i = tid; i < C_FLOATS; i += THREADS) C[i] = 0.f;   // pad rows of the planes must never hold NaN patterns
    __syncthreads();

    // REUSE: the frame index c, the slot of window row 0 and whether the ring holds frames c - 39 .. c (wave-uniform, once per launch)
    int ru_base = 0;
    bool ru_ok = true;
    if (REUSE) {
        const int c = __builtin_amdgcn_readfirstlane(rua.frame_ctr ? *rua.frame_ctr : rua.frame_idx);
        ru_base = (c % T + T + 1) % T;
        const int p = lane < T ? lane : 0;
        int sl = ru_base + p;
        sl -= sl >= T ? T : 0;
        const bool good = c >= T - 1 && reinterpret_cast<const int*>(rua.cache)[sl] == c - (T - 1) + p;
        ru_ok = __builtin_amdgcn_ballot_w64(!good) == 0ull;
    }

    for (int pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
        const int win0 = pair * 2;
        const int nwin = (win0 + 1 < B) ? 2 : 1;
        WRing2<1> g_q;   // Q|K projection ring of the next quad
        if (!REUSE) {   // (layer 0 of the reuse form brings its own rows in: the residual stream arrives under its last quad's attention)
        const int in_soff = (int)(IN_W * 4) + (wave * 2) * (KIN / 16) * 1024;
        WRing2<2> g_in;
        ring2_prefetch<2>(g_in, rsrc, voff, in_soff, (KIN / 16) * 1024);
        // ---- prologue (:63-78) for both windows: U rows 40w + t ------------------------------------------------------
        // (an OPAQUE copy of the thread index: with the plain one the compiler hoists every per-thread row / column split of
        // the staging loops and of the IH store epilogue out of the pair loop and carries them — 27 VGPRs — across the whole
        // kernel, in scratch; recomputing them here costs a few dozen VALU instructions per pair)
        float* U = C;
        for (int i = tid; i < ROWS * LDU / 4; i += THREADS) reinterpret_cast<float4*>(U)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
        {
            // Round 3 (as in fused_encoder_h_kernel): rows wave, wave + 8, ... of the 80 (ten per wave), a lane per column — unit-stride
            // loads without an index division, ALL of them in flight before the first is used, unconditional (clamped addresses),
            // LDS stores through one base + compile-time offsets.  The element loops this replaces were 36 dependent HBM round trips
            // per window pair.
            constexpr int NR = (ROWS + 7) / 8;     // 10
            const int nrows = nwin * T;
            float vi[NR][2], vs[NR][3], vk[NR][3];
            const float* kmb = keep_mask ? keep_mask : x_s;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int r = wave + 8 * j;              // row of the pair: window r / T, frame r % T (the two windows are consecutive)
                const int rc = r < nrows ? r : nrows - 1;
                const float* xi = x_imu + ((size_t)win0 * T + rc) * NI;
                const float* xs = x_s + ((size_t)win0 * T + rc) * S;
                const float* km = kmb + ((size_t)win0 * T + rc) * S;
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
                if (wave + 8 * j < nrows) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        if (lane + 64 * q < NI) pu_i[8 * j * LDU + 64 * q] = vi[j][q];
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        if (lane + 64 * q < S) {
                            float v = vs[j][q];
                            if (v != v) v = 0.f;                               // :65
                            if (keep_mask) v = v * vk[j][q] * keep_scale;     // :77
                            pu_s[8 * j * LDU + 64 * q] = v;
                        }
                }
            }
        }
        __syncthreads();
        // ---- in_linear (:79) + channel shuffle (folded) -----------------------------------------------------------------
        {
            f32x4 acc[RB][2];
            zero_acc2<RB, 2>(acc);
            int au[RB];
            rows_off(au, X_FLOATS, LDU);
            gemm_phase2<RB, 2, KIN / 16>(acc, smem, au, rsrc, voff, in_soff, (KIN / 16) * 1024, g_in, in_soff, (KIN / 16) * 1024);
            // layer 0, quad 0: wave w < 4 -> Q block of head w, w >= 4 -> K block of head w-4
            ring2_prefetch<1>(g_q, rsrc, voff,
                              (int)(LAYER0 * 4) + (int)(QKV_W * 4) + ((wave >> 2) * 16 + (wave & 3)) * 16 * 1024, 0);
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
        }   // !REUSE
        __syncthreads();

#pragma unroll 1
        for (int layer = 0; layer < L; ++layer) {
            const float* LW = wts + LAYER0 + (size_t)layer * LAYER_FLOATS;
            const int lbase = (int)((LAYER0 + (size_t)layer * LAYER_FLOATS) * 4);
            f32x4 acc_o[RB][2];
            zero_acc2<RB, 2>(acc_o);
            WRing2<2> g_o, g_f2r;
            WRing2<1> g_v, g_f1;
            if (REUSE && layer == 0) {
                // ---- layer 0 of the reuse form.  No projection: a quad's Q / K planes and V^T (80 rows x 64 channels each) come from the
                //      ring, and with no GEMM in front of them their read latency would sit in front of every quad's attention.  The
                //      residual stream is not needed before this block's end, so the X region serves as a SECOND set of planes: quads
                //      0 / 2 live there, quads 1 / 3 in C; the requests of quad q + 1 (nine 16-byte loads per thread) go out in front of
                //      attention(q) and are stored behind it; the 80 in_linear rows themselves take that slot of quad 3 and land in X
                //      when quad 2 is done with it.  Barriers as in the projection form: one behind the attention, one behind the
                //      out-projection (whose O operand the next stores would overwrite). ---------------------------------------------
                const __amdgpu_buffer_rsrc_t ring = __builtin_amdgcn_make_buffer_rsrc(
                    uniform_ptr(const_cast<float*>(rua.cache) + ru::HDR + (size_t)win0 * T * ru::SLOT), 0,
                    __builtin_amdgcn_readfirstlane(nwin * T * ru::SLOT * 4), 0x00020000);
                int to = tid;     // (opaque: keeps the row / slot / address splits below out of the pair loop's live registers)
                asm volatile("" : "+v"(to));
                f32x4 rv[10];
                auto request_planes = [&](int q) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        int i = to + j * THREADS;
                        i = i < ROWS * 16 ? i : ROWS * 16 - 1;
                        const int r = i >> 4, c4 = i & 15;
                        const int w = r >= T ? 1 : 0;
                        int sl = ru_base + r - w * T;
                        sl -= sl >= T ? T : 0;
                        const int off = ((w * T + sl) * ru::SLOT + D + q * 64 + c4 * 4) * 4;
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            rv[pl * 3 + j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ring, off, pl * D * 4, 0));
                    }
                };
                auto store_planes = [&](float* base, bool pads) {
                    float* Qb = base;
                    float* Kb = base + PROWS * LDQ;
                    float* Vb = base + 2 * PROWS * LDQ;
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const int i = to + j * THREADS;
                        if (i < ROWS * 16) {
                            const int r = i >> 4, c4 = i & 15, pr = prow(r);
                            *reinterpret_cast<f32x4*>(Qb + pr * LDQ + c4 * 4) = rv[j];
                            *reinterpret_cast<f32x4*>(Kb + pr * LDQ + c4 * 4) = rv[3 + j];
#pragma unroll
                            for (int e = 0; e < 4; ++e) Vb[(c4 * 4 + e) * LDV + pr] = rv[6 + j][e];
                        }
                    }
                    for (int i = to; i < 64 * 16; i += THREADS) {          // the pad keys of V^T: multiplied by P = 0, must be finite
                        const int ch = i >> 4, k = i & 15;
                        Vb[ch * LDV + (k < 8 ? T + k : 48 + T + (k - 8))] = 0.f;
                    }
                    if (pads)                                              // the X region held anything: pad rows 40-47 / 88-95 of Q and K
                        for (int i = to; i < 2 * 16 * (LDQ / 4); i += THREADS) {
                            const int pl = i / (16 * (LDQ / 4)), k = i - pl * (16 * (LDQ / 4));
                            const int r = k / (LDQ / 4), c4 = k - r * (LDQ / 4);
                            *reinterpret_cast<f32x4*>(base + pl * PROWS * LDQ + (r < 8 ? T + r : 48 + T + (r - 8)) * LDQ + c4 * 4) = (f32x4){0.f, 0.f, 0.f, 0.f};
                        }
                };
                request_planes(0);
                store_planes(smem, true);
                __syncthreads();
#pragma unroll 1
                for (int q = 0; q < 4; ++q) {
                    const int hl = wave & 3;
                    float* cur = (q & 1) ? C : smem;
                    float* nxt = (q & 1) ? smem : C;
                    ring2_prefetch<2>(g_o, rsrc, voff, lbase + (int)(WO_W * 4) + ((wave * 2) * 16 + q * 4) * 1024, 16 * 1024);
                    if (q < 3) {
                        request_planes(q + 1);
                    } else {
#pragma unroll
                        for (int j = 0; j < 10; ++j) {
                            const int i = to + j * THREADS, r = i >> 6, c4 = i & 63;
                            const int w = r >= T ? 1 : 0;
                            int sl = ru_base + r - w * T;
                            sl -= sl >= T ? T : 0;
                            rv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ring, ((w * T + sl) * ru::SLOT + c4 * 4) * 4, 0, 0));
                        }
                    }
                    {
                        const int w = wave >> 2;
                        if (w < nwin)
                            attention_head_mfma<LDQ, LDV>(cur + w * 48 * LDQ, cur + PROWS * LDQ + w * 48 * LDQ, cur + 2 * PROWS * LDQ + w * 48, hl * 16, lane, T);
                    }
                    if (q < 3) {
                        store_planes(nxt, (q & 1) != 0);
                    } else {
                        // (an unprimed ring — a window whose 40 tags are not frames c - 39 .. c — poisons the residual stream: NaN rows out)
#pragma unroll
                        for (int j = 0; j < 10; ++j) {
                            const int i = to + j * THREADS, r = i >> 6, c4 = i & 63;
                            if (!ru_ok) rv[j] = (f32x4){__builtin_nanf(""), __builtin_nanf(""), __builtin_nanf(""), __builtin_nanf("")};
                            *reinterpret_cast<f32x4*>(X + r * LDX + c4 * 4) = rv[j];
                        }
                    }
                    __syncthreads();
                    if (q == 3) ring2_prefetch<1>(g_f1, rsrc, voff, lbase + (int)(W1_W * 4) + wave * 16 * 1024, 0);
                    {
                        const int osoff = lbase + (int)(WO_W * 4) + ((wave * 2) * 16 + q * 4) * 1024;
                        const int cb = (int)(cur - smem);
                        int ao[RB];   // O rows live in the (remapped) Q plane of the quad's buffer
#pragma unroll
                        for (int r = 0; r < RB; ++r) ao[r] = cb + prow(r * 16 + l15) * LDQ + lg * 4;
                        gemm_phase2<RB, 2, 4>(acc_o, smem, ao, rsrc, voff, osoff, 16 * 1024, g_o, osoff, 16 * 1024);
                    }
                    __syncthreads();
                }
            } else
#pragma unroll 1
            for (int q = 0; q < 4; ++q) {
                const int hl = wave & 3;              // head inside the quad
                const int head = q * 4 + hl;
                // ---- Q | K projection: one 16-column block per wave, all 5 row blocks -------------------------------
                {
                    const int isk = wave >> 2;        // 0: Q, 1: K
                    f32x4 acc[RB][1];
                    zero_acc2<RB, 1>(acc);
                    const int soff = lbase + (int)(QKV_W * 4) + (isk * 16 + head) * 16 * 1024;
                    int ax[RB];
                    rows_off(ax, 0, LDX);
                    gemm_phase2<RB, 1, 16>(acc, smem, ax, rsrc, voff, soff, 0, g_q, soff, 0);
                    // V fragments of this quad fly during the epilogue
                    const int vsoff = lbase + (int)(QKV_W * 4) + (32 + head) * 16 * 1024;
                    ring2_prefetch<1>(g_v, rsrc, voff, vsoff, 0);
                    const float bv = LW[QKV_B + isk * D + head * 16 + l15];
                    float* plane = isk ? Kp : Qp;
#pragma unroll
                    for (int r = 0; r < RB; ++r)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            plane[prow(r * 16 + lg * 4 + e) * LDQ + hl * 16 + l15] = acc[r][0][e] + bv;
                    // ---- V projection: head hl; waves 0-3 rows 0-47, waves 4-7 rows 48-79 ------------------------------
                    const float bvv = LW[QKV_B + 2 * D + head * 16 + l15];
                    if (wave < 4) {
                        f32x4 av[3][1];
                        zero_acc2<3, 1>(av);
                        const int ar[3] = {ax[0], ax[1], ax[2]};
                        gemm_phase2<3, 1, 16>(av, smem, ar, rsrc, voff, vsoff, 0, g_v, vsoff, 0);
#pragma unroll
                        for (int r = 0; r < 3; ++r)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                Vt[(hl * 16 + l15) * LDV + prow(r * 16 + lg * 4 + e)] = av[r][0][e] + bvv;
                    } else {
                        f32x4 av[2][1];
                        zero_acc2<2, 1>(av);
                        const int ar[2] = {ax[3], ax[4]};
                        gemm_phase2<2, 1, 16>(av, smem, ar, rsrc, voff, vsoff, 0, g_v, vsoff, 0);
#pragma unroll
                        for (int r = 0; r < 2; ++r)
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                Vt[(hl * 16 + l15) * LDV + prow((r + 3) * 16 + lg * 4 + e)] = av[r][0][e] + bvv;
                    }
                    // the 8 pad keys behind each window are multiplied by P = 0: they must be finite whatever the region held
                    for (int i = tid; i < 64 * 16; i += THREADS) {
                        const int ch = i >> 4, k = i & 15;
                        Vt[ch * LDV + (k < 8 ? T + k : 48 + T + (k - 8))] = 0.f;
                    }
                    // out-projection fragments of this quad fly during the barrier and the attention
                    ring2_prefetch<2>(g_o, rsrc, voff, lbase + (int)(WO_W * 4) + ((wave * 2) * 16 + q * 4) * 1024, 16 * 1024);
                }
                F2_STAMP(1 + 5 * q);
                __syncthreads();
                F2_STAMP(2 + 5 * q);
                // ---- attention: wave = (head hl, window wave>>2); window w lives at plane rows 48w .. -------------------
                {
                    const int w = wave >> 2;
                    if (w < nwin)
                        attention_head_mfma<LDQ, LDV>(Qp + w * 48 * LDQ, Kp + w * 48 * LDQ, Vt + w * 48, hl * 16, lane, T);
                }
                F2_STAMP(3 + 5 * q);
                __syncthreads();
                F2_STAMP(4 + 5 * q);
                // next Q|K ring (next quad of this layer) goes out before the out-projection MFMAs
                if (q < 3)
                    ring2_prefetch<1>(g_q, rsrc, voff,
                                      lbase + (int)(QKV_W * 4) + ((wave >> 2) * 16 + (q + 1) * 4 + (wave & 3)) * 16 * 1024, 0);
                else
                    ring2_prefetch<1>(g_f1, rsrc, voff, lbase + (int)(W1_W * 4) + wave * 16 * 1024, 0);
                // ---- out-projection partial: acc_o += O_quad [80 x 64] * Wo[:, 64q .. 64q+63]^T ---------------------------
                {
                    const int osoff = lbase + (int)(WO_W * 4) + ((wave * 2) * 16 + q * 4) * 1024;
                    int ao[RB];   // O rows live in the (remapped) Q plane
#pragma unroll
                    for (int r = 0; r < RB; ++r) ao[r] = X_FLOATS + prow(r * 16 + l15) * LDQ + lg * 4;
                    gemm_phase2<RB, 2, 4>(acc_o, smem, ao, rsrc, voff, osoff, 16 * 1024, g_o, osoff, 16 * 1024);
                }
                __syncthreads();
                F2_STAMP(5 + 5 * q);
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                // (a column block's residual values are READ before the first is written back: `X[i] += v` per element compiles to a
                //  chain of ds_read -> wait -> ds_write round trips)
                const int col = (wave * 2 + n) * 16 + l15;
                const float bv = LW[WO_B + col];
                float xr[RB][4];
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xr[r][e] = X[(r * 16 + lg * 4 + e) * LDX + col];
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) X[(r * 16 + lg * 4 + e) * LDX + col] = xr[r][e] + (acc_o[r][n][e] + bv);
            }
            __syncthreads();
            F2_STAMP(21);
            layernorm_rows16<f2::ROWS, f2::LDX>(X, LW + G1, LW + BE1, wave, lane);
            __syncthreads();
            F2_STAMP(22);
            // ---- feed-forward: 8 hidden chunks of 128; linear2 accumulates in registers ------------------------------------
            float* Hc = C;
            f32x4 acc_f[RB][2];
            zero_acc2<RB, 2>(acc_f);
#pragma unroll 1
            for (int f = 0; f < 8; ++f) {
                {
                    f32x4 acc[RB][1];
                    zero_acc2<RB, 1>(acc);
                    const int w1off = lbase + (int)(W1_W * 4) + (f * 8 + wave) * 16 * 1024;
                    int ax[RB];
                    rows_off(ax, 0, LDX);
                    gemm_phase2<RB, 1, 16>(acc, smem, ax, rsrc, voff, w1off, 0, g_f1, w1off, 0);
                    ring2_prefetch<2>(g_f2r, rsrc, voff, lbase + (int)(W2_W * 4) + ((wave * 2) * 64 + f * 8) * 1024, 64 * 1024);
                    const int col = wave * 16 + l15;
                    const float bv = LW[W1_B + f * 128 + col];
#pragma unroll
                    for (int r = 0; r < RB; ++r)
#pragma unroll
                        for (int e = 0; e < 4; ++e) Hc[(r * 16 + lg * 4 + e) * LDH + col] = fmaxf(acc[r][0][e] + bv, 0.f);
                }
                __syncthreads();
                F2_STAMP(23 + 2 * f);
                {
                    if (f < 7) ring2_prefetch<1>(g_f1, rsrc, voff, lbase + (int)(W1_W * 4) + ((f + 1) * 8 + wave) * 16 * 1024, 0);
                    const int w2off = lbase + (int)(W2_W * 4) + ((wave * 2) * 64 + f * 8) * 1024;
                    int ah[RB];
                    rows_off(ah, X_FLOATS, LDH);
                    gemm_phase2<RB, 2, 8>(acc_f, smem, ah, rsrc, voff, w2off, 64 * 1024, g_f2r, w2off, 64 * 1024);
                }
                __syncthreads();
                F2_STAMP(24 + 2 * f);
            }
            if (layer + 1 < L)
                ring2_prefetch<1>(g_q, rsrc, voff,
                                  lbase + (int)(LAYER_FLOATS * 4) + (int)(QKV_W * 4) + ((wave >> 2) * 16 + (wave & 3)) * 16 * 1024, 0);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                // (a column block's residual values are READ before the first is written back: `X[i] += v` per element compiles to a
                //  chain of ds_read -> wait -> ds_write round trips)
                const int col = (wave * 2 + n) * 16 + l15;
                const float bv = LW[W2_B + col];
                float xr[RB][4];
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xr[r][e] = X[(r * 16 + lg * 4 + e) * LDX + col];
#pragma unroll
                for (int r = 0; r < RB; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) X[(r * 16 + lg * 4 + e) * LDX + col] = xr[r][e] + (acc_f[r][n][e] + bv);
            }
            __syncthreads();
            F2_STAMP(40);
            layernorm_rows16<f2::ROWS, f2::LDX>(X, LW + G2, LW + BE2, wave, lane);
            __syncthreads();
            F2_STAMP(41);
        }
        // ---- RNN input projection: IH = X W_ih^T + (b_ih + b_hh) for both windows -> HBM -------------------------------------
        {
            f32x4 acc[RB][4];
            zero_acc2<RB, 4>(acc);
            const int isoff = ih_off_b + (wave * 4) * 16 * 1024;
            WRing2<4> g_ih;
            ring2_prefetch<4>(g_ih, rsrc, voff, isoff, 16 * 1024);
            unsigned tn = 0;
            if (REUSE && pair + (int)gridDim.x < npairs) {
                // the workgroup's NEXT pair: quad 0's planes on their way into L2 under this projection
                const int nw0 = (pair + (int)gridDim.x) * 2, nn = (nw0 + 1 < B) ? 2 : 1;
                const __amdgpu_buffer_rsrc_t nring = __builtin_amdgcn_make_buffer_rsrc(
                    uniform_ptr(const_cast<float*>(rua.cache) + ru::HDR + (size_t)nw0 * T * ru::SLOT), 0,
                    __builtin_amdgcn_readfirstlane(nn * T * ru::SLOT * 4), 0x00020000);
                int to = tid;
                asm volatile("" : "+v"(to));
                tn = ring_touch_planes(nring, ru_base, 0, to);
            }
            int ax[RB];
            rows_off(ax, 0, LDX);
            // Swapped operands: transposed accumulators, lane (l15, lg) = row 16 r + l15, columns 16 n + 4 lg ..: ONE 16-byte store per
            // tile instead of four scattered 4-byte ones, through a buffer whose extent is the pair's real rows (the second window's
            // pad rows, and an absent second window, are dropped by the range check instead of by a branch around every store), and
            // the four bias vectors requested before the first store (a load cannot be hoisted over a store that may alias).
            // Was 80 guarded 4-byte stores per lane, each behind its own 64-bit address computation, and a bias round trip per block.
            gemm_phase2<RB, 4, 16, true>(acc, smem, ax, rsrc, voff, isoff, 16 * 1024, g_ih, isoff, 16 * 1024);
            if (REUSE) touch_sink(tn);
            const int nrows = nwin * T;                  // the two windows are consecutive: rows 0..79 map 1:1
            const __amdgpu_buffer_rsrc_t io_rs = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(ih_out + (size_t)win0 * T * R), 0,
                                                                                    __builtin_amdgcn_readfirstlane(nrows * R * 4), 0x00020000);
            int le = lane;                               // opaque lane index: store offsets are formed here, not in the prologue
            asm volatile("" : "+v"(le));
            const int l15e = le & 15, lge = le >> 4;
            f32x4 bv4[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) bv4[n] = *reinterpret_cast<const f32x4*>(wts + ih_off_b / 4 + R * D + (wave * 4 + n) * 16 + lge * 4);
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int r = 0; r < RB; ++r)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_f2, acc[r][n] + bv4[n]), io_rs,
                                                           ((r * 16 + l15e) * R + (wave * 4 + n) * 16 + lge * 4) * 4, 0, 0);
        }
        if (hall_sentinel) {
            uint4* hp = reinterpret_cast<uint4*>(hall_sentinel + (size_t)win0 * T * R);
            int ts = tid;
            asm volatile("" : "+v"(ts));
            for (int i = ts; i < nwin * T * (R / 4); i += THREADS) hp[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        }
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------------
#undef F2_STAMP


// Pair-split plan ("fused2s"): the same two-window workgroup, but a window PAIR is shared by TWO workgroups on two CUs of
// one XCD, split by COLUMNS.  Why: with one window per CU (B = #CUs) the 40 rows pad to 3 MFMA row blocks, and no
// assignment of whole row blocks to CUs can do better (10 240 rows = 640 blocks on 256 CUs is 2.5 each).  Here every CU
// carries 80 rows (5 blocks, no padding) of HALF the columns: 2.5 block-columns instead of 3.
//   half h owns heads 8h .. 8h+7 (Q|K|V projection, attention), the K-half of the out-projection that those heads feed,
//   FFN hidden units 512h .. 512h+511 (linear1 chunks, and the K-half of linear2 they feed), columns 256h .. of the
//   RNN input projection.  in_linear, the residual stream and both LayerNorms are carried in full by both (identical bits).
// Two hand-offs per layer: the [80 x 256] partial sums of the out-projection and of linear2 are exchanged as register
// images ([wave][tile][lane] float4: the partner has the same thread -> element mapping) through a 2-slot buffer in
// HBM/L2, and both sides add them in the same order (half 0 + half 1).  Protocol per hand-off k: image stores ->
// s_waitcnt vmcnt(0) -> flag[h] = k + 1 (agent scope); poll flag[1 - h] >= k + 1 (bounded) -> sc1 (L1-bypassing) loads.
// Slot k & 1 is free again when the partner's flag k has been seen (it read slot k - 1's predecessor before producing k).
// The partners' XCC ids are compared at run time: same XCD -> plain stores (shared L2); otherwise agent-scope stores.
// All 2 * ceil(B / 2) workgroups must be co-resident (the launcher checks grid <= #CUs; one workgroup per CU by LDS size).
// ---------------------------------------------------------------------------------------------------------------------
__device__ unsigned g_spin_timeouts_fused2;
__device__ unsigned g_f2s_cross_xcd;
__device__ unsigned long long g_f2s_trace[32];   // measurement: s_memtime stamps of workgroup 0 around its hand-offs
__device__ __forceinline__ void note_spin_timeout(unsigned* err) {
    atomicAdd(&g_spin_timeouts_fused2, 1u);
    guard_report(err);
}
hipError_t read_spin_timeouts_fused2(unsigned* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_spin_timeouts_fused2), sizeof(unsigned));
}

namespace f2 {
constexpr int IMG_FLOATS = ROWS * D;                 // one partial-sum image: 8 waves x 10 tiles x 64 lanes x 4
constexpr int PAIR_IMG_FLOATS = 4 * IMG_FLOATS;      // per pair: [slot 2][half 2] images
constexpr int PAIR_FLAG_WORDS = 32;                  // per pair, behind ALL images: [0..1] hand-off counters, [16..17] XCC id + 1
}
// (sized for either form: window pairs, or — B <= 128 — single windows with 48-row images)
size_t fused2s_xchg_floats(int B) {
    const size_t pair = (size_t)((B + 1) / 2) * (f2::PAIR_IMG_FLOATS + f2::PAIR_FLAG_WORDS);
    const size_t one = B <= 128 ? (size_t)B * (4 * 48 * f2::D + f2::PAIR_FLAG_WORDS) : 0;
    const size_t four = B <= 64 ? (size_t)B * (8 * 48 * f2::D + f2::PAIR_FLAG_WORDS) : 0;   // one window on four workgroups
    return std::max(pair, std::max(one, four));
}

// NWIN = 2: a window PAIR on two workgroups (80 rows = 5 row blocks, no padding).  NWIN = 1 (round 4): ONE window on two workgroups
// (48 rows = 3 row blocks, 8 of them pad): the same column split and hand-offs, 3/5 of the matrix work per workgroup — for batches
// that leave at least half of the CUs idle (64 < B <= #CUs / 2), where a window's latency, not the round count, is what a step costs.
// NPART = 4 (round 4, NWIN = 1 only): ONE window on FOUR workgroups of one XCD — part p owns heads 4 p .. 4 p + 3 (one quad), hidden
// units 256 p .., columns 128 p .. of the RNN input projection; the partial sums of out-proj / linear2 meet as p0 + p1 + p2 + p3 in
// every partner (same order: bit-identical rows whichever partner stores them).  For batches that leave three quarters of the CUs idle.
template <int NWIN, int NPART = 2>
__global__ __launch_bounds__(f2::THREADS) void fused_encoder2s_kernel(
    const float* __restrict__ wts, const float* __restrict__ x_imu, const float* __restrict__ x_s,
    const float* __restrict__ keep_mask, float keep_scale, float* __restrict__ ih_out, unsigned* __restrict__ hall_sentinel,
    float* __restrict__ xchg, int xchg_bytes, int B, int NI, int S, int L, int wbytes, int ih_off_b, Guard gd) {
    using namespace f2;
    constexpr int RBK = NWIN == 2 ? 5 : 3;                     // 16-row blocks a workgroup carries
    constexpr int IMG = RBK * 16 * D;                          // one partial-sum image: 8 waves x 2 RBK tiles x 64 lanes x 4
    constexpr int PAIR_IMG = 2 * NPART * IMG;                  // per pair: [slot 2][part NPART] images
    static_assert(NPART == 2 || (NPART == 4 && NWIN == 1), "two partners, or four for a single window");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int s_same_xcd;
    __shared__ int s_poison;   // a hand-off wait gave up: the partner's partial sums count as NaN from here on (and no wait spins again)
    float* X = smem;
    float* C = smem + X_FLOATS;
    float* Qp = C;
    float* Kp = C + PROWS * LDQ;
    float* Vt = C + 2 * PROWS * LDQ;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wts), 0, wbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(xchg, 0, xchg_bytes, 0x00020000);
    const int voff = lane * 16;
    const int npairs = (B + NWIN - 1) / NWIN;                  // "pair" = the NWIN windows two partner workgroups share
    // workgroup -> (pair, part): the parts of a pair (called `half` below: two of them, or four) are 8 workgroup ids apart, i.e. on the same XCD when ids go
    // round-robin over the 8 XCDs (checked below, never assumed)
    const int xslot = blockIdx.x & 7, jj = blockIdx.x >> 3;
    const int half = jj % NPART;                           // this workgroup's part (0 .. NPART-1; "half" when NPART = 2)
    const int pair = (jj / NPART) * 8 + xslot;
    if (pair >= npairs) return;                           // (all parts of a surplus pair leave together)
    if ((gd.fault & 1) && pair == 0 && half == 1) return; // TIP_OPT_FAULT_INJECT: this partner never arrives
    float* px = xchg + (size_t)pair * PAIR_IMG;
    unsigned* pflag = reinterpret_cast<unsigned*>(xchg + (size_t)npairs * PAIR_IMG) + pair * PAIR_FLAG_WORDS;
    const int px_off_b = (int)((size_t)pair * PAIR_IMG * 4);

    auto rows_off = [&](int (&off)[RBK], int base, int ld) {
#pragma unroll
        for (int r = 0; r < RBK; ++r) off[r] = base + (r * 16 + l15) * ld + lg * 4;
    };
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_store(pflag + 16 + half, (xcc & 0xf) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_same_xcd = -1;                                  // decided at the first hand-off
        s_poison = 0;
    }
    for (int i = tid; i < C_FLOATS; i += THREADS) C[i] = 0.f;
    __syncthreads();

    const int win0 = pair * NWIN;
    const int nwin = (NWIN == 2 && win0 + 1 < B) ? 2 : 1;
    int handoff = 0;                                      // hand-offs done so far (uniform)

    // X[:, all columns] += (partial of part 0 + part 1 (+ part 2 + part 3)) + bias: acc holds THIS part's partial.
    // (A wave-to-wave variant — every wave publishing its own 10 tiles under its own counter and waiting only for its twin,
    // no workgroup barrier inside the hand-off — measured SLOWER: 0.787 vs 0.768 ms per step.)
    auto exchange_add = [&](f32x4 (&acc)[RBK][2], const float* bias) {
        if (tid == 0 && s_same_xcd < 0) {                  // first hand-off: are the partners on one XCD?
            unsigned mine = 0;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(mine));
            mine = (mine & 0xf) + 1u;
            const unsigned lim0 = guard_spin_limit(gd.fault, 1u << 22);
            bool same = true;
            for (int pp = 0; pp < NPART; ++pp) {
                if (pp == half) continue;
                unsigned theirs = 0;
                for (unsigned spins = 0; spins < lim0; ++spins) {
                    theirs = __hip_atomic_load(pflag + 16 + pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (theirs) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                if (!theirs) {                             // a partner is not there at all: poison now, never wait again
                    if (!s_poison) note_spin_timeout(gd.err);
                    s_poison = 1;
                }
                if (theirs != mine) atomicAdd(&g_f2s_cross_xcd, 1u);   // (measurement: pairs that straddle XCDs)
                same &= theirs == mine;
            }
            s_same_xcd = (same && !(gd.fault & 8)) ? 1 : 0;   // (fault bit 3: take the cross-XCD path wherever the partners sit)
        }
        __syncthreads();
        if (blockIdx.x == 0 && tid == 0 && handoff < 8) g_f2s_trace[handoff * 4 + 0] = __builtin_amdgcn_s_memtime();
        const bool same = s_same_xcd == 1;
        const int slot = handoff & 1;
        float* mine_img = px + (size_t)(slot * NPART + half) * IMG;
        // image: [wave][r * 2 + n][lane] float4
#pragma unroll
        for (int r = 0; r < RBK; ++r)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                float* dst = mine_img + ((wave * (RBK * 2) + r * 2 + n) * 64 + lane) * 4;
                if (same) {
                    *reinterpret_cast<f32x4*>(dst) = acc[r][n];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) __hip_atomic_store(dst + e, acc[r][n][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's image stores are acknowledged ...
        __syncthreads();                                   // ... before the one counter store that publishes them
        if (blockIdx.x == 0 && tid == 0 && handoff < 8) g_f2s_trace[handoff * 4 + 1] = __builtin_amdgcn_s_memtime();
        if (tid == 0) {
            __hip_atomic_store(pflag + half, (unsigned)(handoff + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!s_poison) {
                const unsigned lim = guard_spin_limit(gd.fault, 1u << 22);
                for (int pp = 0; pp < NPART && !s_poison; ++pp) {
                    if (pp == half) continue;
                    unsigned spins = 0;
                    for (; spins < lim; ++spins) {
                        if (__hip_atomic_load(pflag + pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)(handoff + 1)) break;
                        if (!same) __builtin_amdgcn_s_sleep(1);
                    }
                    if (spins == lim) {
                        note_spin_timeout(gd.err);
                        s_poison = 1;
                    }
                }
            }
        }
        __syncthreads();
        const bool poisoned = s_poison != 0;
        if (blockIdx.x == 0 && tid == 0 && handoff < 8) g_f2s_trace[handoff * 4 + 2] = __builtin_amdgcn_s_memtime();
        // the partners' images (sc1: not from L1), summed in part order p0 + p1 (+ p2 + p3) — the same order in every partner
        f32x4 sum[RBK][2];
#pragma unroll
        for (int r = 0; r < RBK; ++r)
#pragma unroll
            for (int n = 0; n < 2; ++n) sum[r][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int pp = 0; pp < NPART; ++pp) {
            const int off_b = px_off_b + (slot * NPART + pp) * IMG * 4;
#pragma unroll
            for (int r = 0; r < RBK; ++r)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    f32x4 v = acc[r][n];
                    if (pp != half) {
                        v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, lane * 16, off_b + (wave * (RBK * 2) + r * 2 + n) * 1024, 16));
                        if (poisoned) v = (f32x4){__uint_as_float(kPoisonBits), __uint_as_float(kPoisonBits), __uint_as_float(kPoisonBits), __uint_as_float(kPoisonBits)};
                    }
                    sum[r][n] = pp == 0 ? v : sum[r][n] + v;
                }
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = (wave * 2 + n) * 16 + l15;
            const float bv = bias[col];
#pragma unroll
            for (int r = 0; r < RBK; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) X[(r * 16 + lg * 4 + e) * LDX + col] += sum[r][n][e] + bv;
        }
        if (blockIdx.x == 0 && tid == 0 && handoff < 8) g_f2s_trace[handoff * 4 + 3] = __builtin_amdgcn_s_memtime();
        ++handoff;
    };

    {
        const int in_soff = (int)(IN_W * 4) + (wave * 2) * (KIN / 16) * 1024;
        WRing2<2> g_in;
        ring2_prefetch<2>(g_in, rsrc, voff, in_soff, (KIN / 16) * 1024);
        // ---- prologue (:63-78) for the pair's windows (carried by every part) --------------------------------------------
        float* U = C;
        for (int i = tid; i < ROWS * LDU; i += THREADS) U[i] = 0.f;
        __syncthreads();
        for (int w = 0; w < nwin; ++w) {
            const float* xi = x_imu + (size_t)(win0 + w) * T * NI;
            for (int i = tid; i < T * NI; i += THREADS) {
                const int r = i / NI, c = i - r * NI;
                U[(w * T + r) * LDU + c] = xi[i];
            }
            const float* xs = x_s + (size_t)(win0 + w) * T * S;
            const float* km = keep_mask ? keep_mask + (size_t)(win0 + w) * T * S : nullptr;
            for (int i = tid; i < T * S; i += THREADS) {
                const int r = i / S, c = i - r * S;
                float v = xs[i];
                if (v != v) v = 0.f;                  // :65
                if (km) v = v * km[i] * keep_scale;   // :77
                U[(w * T + r) * LDU + NI + c] = v;
            }
        }
        __syncthreads();
        // ---- in_linear (:79) + channel shuffle (folded), all 256 columns on every part -----------------------------------
        f32x4 acc[RBK][2];
        zero_acc2<RBK, 2>(acc);
        int au[RBK];
        rows_off(au, X_FLOATS, LDU);
        gemm_phase2<RBK, 2, KIN / 16>(acc, smem, au, rsrc, voff, in_soff, (KIN / 16) * 1024, g_in, in_soff, (KIN / 16) * 1024);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int col = (wave * 2 + n) * 16 + l15;
            const float bv = wts[IN_B + col];
#pragma unroll
            for (int r = 0; r < RBK; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) X[(r * 16 + lg * 4 + e) * LDX + col] = acc[r][n][e] + bv;
        }
    }
    __syncthreads();

    WRing2<1> g_q;   // Q|K projection ring of the next quad: primed one phase ahead (here for layer 0, quad 2 * half)
    ring2_prefetch<1>(g_q, rsrc, voff, (int)(LAYER0 * 4) + (int)(QKV_W * 4) + ((wave >> 2) * 16 + half * (16 / NPART) + (wave & 3)) * 16 * 1024, 0);
#pragma unroll 1
    for (int layer = 0; layer < L; ++layer) {
        const float* LW = wts + LAYER0 + (size_t)layer * LAYER_FLOATS;
        const int lbase = (int)((LAYER0 + (size_t)layer * LAYER_FLOATS) * 4);
        f32x4 acc_o[RBK][2];
        zero_acc2<RBK, 2>(acc_o);
        WRing2<2> g_o, g_f2r;
        WRing2<1> g_v, g_f1;
        constexpr int NQ = 4 / NPART;                  // quads of four heads per part
#pragma unroll 1
        for (int qq = 0; qq < NQ; ++qq) {
            const int q = half * NQ + qq;             // this part's quads
            const int hl = wave & 3;
            const int head = q * 4 + hl;
            {
                const int isk = wave >> 2;            // 0: Q, 1: K
                f32x4 acc[RBK][1];
                zero_acc2<RBK, 1>(acc);
                const int soff = lbase + (int)(QKV_W * 4) + (isk * 16 + head) * 16 * 1024;
                int ax[RBK];
                rows_off(ax, 0, LDX);
                gemm_phase2<RBK, 1, 16>(acc, smem, ax, rsrc, voff, soff, 0, g_q, soff, 0);
                const int vsoff = lbase + (int)(QKV_W * 4) + (32 + head) * 16 * 1024;
                ring2_prefetch<1>(g_v, rsrc, voff, vsoff, 0);
                const float bv = LW[QKV_B + isk * D + head * 16 + l15];
                float* plane = isk ? Kp : Qp;
#pragma unroll
                for (int r = 0; r < RBK; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        plane[prow(r * 16 + lg * 4 + e) * LDQ + hl * 16 + l15] = acc[r][0][e] + bv;
                const float bvv = LW[QKV_B + 2 * D + head * 16 + l15];
                // V of head hl: waves hl and 4 + hl share the row blocks (0-2 | 3-4 for a pair, 0-1 | 2 for one window)
                constexpr int VA = NWIN == 2 ? 3 : 2, VB = RBK - VA;
                if (wave < 4) {
                    f32x4 av[VA][1];
                    zero_acc2<VA, 1>(av);
                    int ar[VA];
#pragma unroll
                    for (int r = 0; r < VA; ++r) ar[r] = ax[r];
                    gemm_phase2<VA, 1, 16>(av, smem, ar, rsrc, voff, vsoff, 0, g_v, vsoff, 0);
#pragma unroll
                    for (int r = 0; r < VA; ++r)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            Vt[(hl * 16 + l15) * LDV + prow(r * 16 + lg * 4 + e)] = av[r][0][e] + bvv;
                } else {
                    f32x4 av[VB][1];
                    zero_acc2<VB, 1>(av);
                    int ar[VB];
#pragma unroll
                    for (int r = 0; r < VB; ++r) ar[r] = ax[VA + r];
                    gemm_phase2<VB, 1, 16>(av, smem, ar, rsrc, voff, vsoff, 0, g_v, vsoff, 0);
#pragma unroll
                    for (int r = 0; r < VB; ++r)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            Vt[(hl * 16 + l15) * LDV + prow((r + VA) * 16 + lg * 4 + e)] = av[r][0][e] + bvv;
                }
                for (int i = tid; i < 64 * 16; i += THREADS) {
                    const int ch = i >> 4, k = i & 15;
                    Vt[ch * LDV + (k < 8 ? T + k : 48 + T + (k - 8))] = 0.f;
                }
                ring2_prefetch<2>(g_o, rsrc, voff, lbase + (int)(WO_W * 4) + ((wave * 2) * 16 + q * 4) * 1024, 16 * 1024);
            }
            __syncthreads();
            {
                const int w = wave >> 2;
                if (w < nwin)
                    attention_head_mfma<LDQ, LDV>(Qp + w * 48 * LDQ, Kp + w * 48 * LDQ, Vt + w * 48, hl * 16, lane, T);
            }
            __syncthreads();
            if (qq + 1 < NQ)
                ring2_prefetch<1>(g_q, rsrc, voff,
                                  lbase + (int)(QKV_W * 4) + ((wave >> 2) * 16 + (q + 1) * 4 + (wave & 3)) * 16 * 1024, 0);
            else
                ring2_prefetch<1>(g_f1, rsrc, voff, lbase + (int)(W1_W * 4) + (half * (64 / NPART) + wave) * 16 * 1024, 0);
            {
                const int osoff = lbase + (int)(WO_W * 4) + ((wave * 2) * 16 + q * 4) * 1024;
                int ao[RBK];
#pragma unroll
                for (int r = 0; r < RBK; ++r) ao[r] = X_FLOATS + prow(r * 16 + l15) * LDQ + lg * 4;
                gemm_phase2<RBK, 2, 4>(acc_o, smem, ao, rsrc, voff, osoff, 16 * 1024, g_o, osoff, 16 * 1024);
            }
            __syncthreads();
        }
        exchange_add(acc_o, LW + WO_B);               // X += out-projection (every part's heads) + bias
        __syncthreads();
        layernorm_rows16<RBK * 16, f2::LDX>(X, LW + G1, LW + BE1, wave, lane);
        __syncthreads();
        // ---- feed-forward: this part's 8 / NPART hidden chunks of 128; its K-slice of linear2 accumulates in registers ------
        float* Hc = C;
        f32x4 acc_f[RBK][2];
        zero_acc2<RBK, 2>(acc_f);
        constexpr int NF = 8 / NPART;                  // hidden chunks of 128 per part
#pragma unroll 1
        for (int ff = 0; ff < NF; ++ff) {
            const int f = half * NF + ff;
            {
                f32x4 acc[RBK][1];
                zero_acc2<RBK, 1>(acc);
                const int w1off = lbase + (int)(W1_W * 4) + (f * 8 + wave) * 16 * 1024;
                int ax[RBK];
                rows_off(ax, 0, LDX);
                gemm_phase2<RBK, 1, 16>(acc, smem, ax, rsrc, voff, w1off, 0, g_f1, w1off, 0);
                ring2_prefetch<2>(g_f2r, rsrc, voff, lbase + (int)(W2_W * 4) + ((wave * 2) * 64 + f * 8) * 1024, 64 * 1024);
                const int col = wave * 16 + l15;
                const float bv = LW[W1_B + f * 128 + col];
#pragma unroll
                for (int r = 0; r < RBK; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) Hc[(r * 16 + lg * 4 + e) * LDH + col] = fmaxf(acc[r][0][e] + bv, 0.f);
            }
            __syncthreads();
            {
                if (ff + 1 < NF) ring2_prefetch<1>(g_f1, rsrc, voff, lbase + (int)(W1_W * 4) + ((f + 1) * 8 + wave) * 16 * 1024, 0);
                const int w2off = lbase + (int)(W2_W * 4) + ((wave * 2) * 64 + f * 8) * 1024;
                int ah[RBK];
                rows_off(ah, X_FLOATS, LDH);
                gemm_phase2<RBK, 2, 8>(acc_f, smem, ah, rsrc, voff, w2off, 64 * 1024, g_f2r, w2off, 64 * 1024);
            }
            __syncthreads();
        }
        if (layer + 1 < L)                              // next layer's first Q|K ring flies during the hand-off and LayerNorm2
            ring2_prefetch<1>(g_q, rsrc, voff, lbase + (int)(LAYER_FLOATS * 4) + (int)(QKV_W * 4) +
                                                   ((wave >> 2) * 16 + half * (16 / NPART) + (wave & 3)) * 16 * 1024, 0);
        exchange_add(acc_f, LW + W2_B);               // X += linear2 (every part's hidden units) + bias
        __syncthreads();
        layernorm_rows16<RBK * 16, f2::LDX>(X, LW + G2, LW + BE2, wave, lane);
        __syncthreads();
    }
    // ---- RNN input projection, columns (512 / NPART) * part ..: IH = X W_ih^T + (b_ih + b_hh) -> HBM -----------------------------
    {
        constexpr int NBI = 4 / NPART;                 // 16-column blocks per wave: 2 (halves) or 1 (quarters)
        f32x4 acc[RBK][NBI];
        zero_acc2<RBK, NBI>(acc);
        const int nb0 = half * (32 / NPART) + wave * NBI;
        const int isoff = ih_off_b + nb0 * 16 * 1024;
        WRing2<NBI> g_ih;
        ring2_prefetch<NBI>(g_ih, rsrc, voff, isoff, 16 * 1024);
        int ax[RBK];
        rows_off(ax, 0, LDX);
        // (transposed tiles, 16-byte buffer stores bounded to the pair's real rows, biases first: as in fused_encoder2_kernel)
        gemm_phase2<RBK, NBI, 16, true>(acc, smem, ax, rsrc, voff, isoff, 16 * 1024, g_ih, isoff, 16 * 1024);
        const int nrows = nwin * T;
        const __amdgpu_buffer_rsrc_t io_rs = __builtin_amdgcn_make_buffer_rsrc(uniform_ptr(ih_out + (size_t)win0 * T * R), 0,
                                                                                __builtin_amdgcn_readfirstlane(nrows * R * 4), 0x00020000);
        f32x4 bv4[NBI];
#pragma unroll
        for (int n = 0; n < NBI; ++n) bv4[n] = *reinterpret_cast<const f32x4*>(wts + ih_off_b / 4 + R * D + (nb0 + n) * 16 + lg * 4);
#pragma unroll
        for (int n = 0; n < NBI; ++n)
#pragma unroll
            for (int r = 0; r < RBK; ++r)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_f2, acc[r][n] + bv4[n]), io_rs,
                                                       ((r * 16 + l15) * R + (nb0 + n) * 16 + lg * 4) * 4, 0, 0);
    }
    if (hall_sentinel) {   // each part arms its share of the pair's HALL rows
        uint4* hp = reinterpret_cast<uint4*>(hall_sentinel + (size_t)win0 * T * R);
        const int n4 = nwin * T * (R / 4), h0 = half * (n4 / NPART), h1 = half == NPART - 1 ? n4 : (half + 1) * (n4 / NPART);
        for (int i = h0 + tid; i < h1; i += THREADS) hp[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
    }
}

bool fused2_supported(const Dims& d, int T) { return fused_supported(d, T) && fused_has_rnn_ih(d) && T == f2::T; }

// ---------------------------------------------------------------------------------------------------------------------
// The ring's writer: the NEWEST row (window row T - 1) of every stream -> its in_linear row and layer-0 Q | K | V row, into slot
// c mod 40 of the stream's ring; the tag of that slot <- c.  16 streams form one 16-row MFMA block; four 256-thread workgroups
// share a tile — each computes the tile's in_linear rows (the QKV projection's A operand, 224 x 256) and a quarter of the 768 QKV
// columns — so that 1024 streams are 256 workgroups of ~6 us.  Arithmetic per output element exactly as fused_encoder2_kernel's
// (and reuse's contract with it): gemm_phase2 -> mfma_block2 in the plain orientation over the same packed fragments, k-blocks in
// ascending order from a zero accumulator, `+ bias` last; the prologue's NaN scrub (:65) on the way in.  No keep mask, no dropout:
// tip_forward_reuse refuses them.
// ---------------------------------------------------------------------------------------------------------------------
// acc[0][n] += A[16 x 16 KB] (LDS) * Wblock(n, kb), kb ascending from 0 — gemm_phase2's sums for ONE row block, with a weight ring DEPTH
// k-blocks deep.  16 NBW MFMAs per k-block (512 / 384 cycles at NBW = 4 / 3) do not cover a first-touch L2 miss (~900 cycles) two
// k-blocks ahead, and this kernel runs one wave per SIMD on weights the previous frame's encoder pushed out of every L2: measured
// (s_memtime stamps, 1024 streams) the two GEMMs of the writer at 1.2x their MFMA time only after 416 KB of warm-up loads per
// workgroup that cost 4.9 us up front.  Fully unrolled: the ring slot of a k-block is a compile-time register set.
template <int NBW, int DEPTH>
struct DeepRing {
    float4 w[DEPTH][NBW];
};
template <int NBW, int DEPTH>
__device__ __forceinline__ void deep_prefetch(DeepRing<NBW, DEPTH>& g, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff, int nstride_b) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int n = 0; n < NBW; ++n) g.w[d][n] = ldfrag2(rsrc, voff, soff + n * nstride_b + d * 1024);
}
template <int NBW, int KB, int DEPTH>   // the ring holds k-blocks 0 .. DEPTH-1 on entry (deep_prefetch, called a phase early)
__device__ __forceinline__ void gemm_rows16_deep(f32x4 (&acc)[1][NBW], const float* As, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff,
                                                 int nstride_b, DeepRing<NBW, DEPTH>& g) {
    // A ROLLED loop over groups of DEPTH k-blocks (+ a tail of KB % DEPTH): this kernel's code is fetched by all 128 instruction caches of
    // the chip at the same moment and executed once — fully unrolled (10 KB) the fetch alone was worth more than the MFMAs
    // (stream_consume_kernel: 2.4 us of rotation math cost 13-25 us the first time through, profiles/r06/stream_glue_probe.txt).
    float4 a[2][1];
    a[0][0] = *reinterpret_cast<const float4*>(As);
    constexpr int FULL = KB / DEPTH * DEPTH;
#pragma unroll 1
    for (int kb0 = 0; kb0 < FULL; kb0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int kb = kb0 + d;
            const int ka = kb + 1 < KB ? kb + 1 : KB - 1, kw = kb + DEPTH < KB ? kb + DEPTH : KB - 1;   // (past the end: harmless reloads)
            a[(d + 1) & 1][0] = *reinterpret_cast<const float4*>(As + ka * 16);
            mfma_block2<1, NBW>(acc, a[d & 1], g.w[d]);
#pragma unroll
            for (int n = 0; n < NBW; ++n) g.w[d][n] = ldfrag2(rsrc, voff, soff + n * nstride_b + kw * 1024);
            __builtin_amdgcn_sched_barrier(0);   // (keeps every refill behind the block that freed its registers, DEPTH k-blocks ahead of its use)
        }
    }
    static_assert(DEPTH % 2 == 0, "the A double buffer's parity restarts with every group");
#pragma unroll
    for (int d = 0; d < KB - FULL; ++d) {
        if (FULL + d + 1 < KB) a[(d + 1) & 1][0] = *reinterpret_cast<const float4*>(As + (FULL + d + 1) * 16);
        mfma_block2<1, NBW>(acc, a[d & 1], g.w[d]);
    }
}

__global__ __launch_bounds__(ru::THREADS) void reuse_update_kernel(const float* __restrict__ wts, const float* __restrict__ x_imu,
                                                                  const float* __restrict__ x_s, float* __restrict__ cache,
                                                                  const int* __restrict__ frame_ctr, int frame_idx, int B, int T, int NI,
                                                                  int S, int wbytes) {
    using namespace f2;
    __shared__ __attribute__((aligned(16))) float sm[ru::TILE * LDU + ru::TILE * LDX + 64];
    float* U = sm;
    float* Xs = sm + ru::TILE * LDU;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int tile = blockIdx.x / ru::PARTS, part = blockIdx.x % ru::PARTS;
    const int s0 = tile * ru::TILE;
    const int c = __builtin_amdgcn_readfirstlane(frame_ctr ? *frame_ctr : frame_idx);
    const int slot = (c % f2::T + f2::T) % f2::T;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wts), 0, wbytes, 0x00020000);
    const int voff = lane * 16;
    const int in_soff = (int)(IN_W * 4) + (wave * 4) * (KIN / 16) * 1024;
    const int nb0 = part * (48 / ru::PARTS) + wave * 3;                // this wave's three of the 48 QKV column blocks
    const int q_soff = (int)(LAYER0 * 4) + (int)(QKV_W * 4) + nb0 * 16 * 1024;
    DeepRing<4, 4> g_in;
    deep_prefetch<4, 4>(g_in, rsrc, voff, in_soff, (KIN / 16) * 1024);
    // (the biases of both epilogues are requested here: a load behind a GEMM phase is a round trip with the matrix pipe idle)
    float b_in[4], b_qkv[3];
#pragma unroll
    for (int n = 0; n < 4; ++n) b_in[n] = wts[IN_B + (wave * 4 + n) * 16 + l15];
#pragma unroll
    for (int n = 0; n < 3; ++n) b_qkv[n] = wts[LAYER0 + QKV_B + (nb0 + n) * 16 + l15];
    for (int i = tid; i < (ru::TILE * LDU + ru::TILE * LDX + 64) / 4; i += ru::THREADS) reinterpret_cast<float4*>(sm)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();
    {
        float vi[4][2], vs[4][3];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int st = s0 + wave + 4 * j < B ? s0 + wave + 4 * j : B - 1;
            const float* xi = x_imu + ((size_t)st * T + (T - 1)) * NI;
            const float* xs = x_s + ((size_t)st * T + (T - 1)) * S;
#pragma unroll
            for (int q = 0; q < 2; ++q) vi[j][q] = xi[lane + 64 * q < NI ? lane + 64 * q : NI - 1];
#pragma unroll
            for (int q = 0; q < 3; ++q) vs[j][q] = xs[lane + 64 * q < S ? lane + 64 * q : S - 1];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float* pu = U + (wave + 4 * j) * LDU + lane;
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (lane + 64 * q < NI) pu[64 * q] = vi[j][q];
#pragma unroll
            for (int q = 0; q < 3; ++q)
                if (lane + 64 * q < S) {
                    float v = vs[j][q];
                    if (v != v) v = 0.f;                               // :65
                    pu[NI + 64 * q] = v;
                }
        }
    }
    __syncthreads();
    float* slot_rows = cache + ru::HDR + (size_t)slot * ru::SLOT;      // + stream * 40 * SLOT
    {
        f32x4 acc[1][4];
        zero_acc2<1, 4>(acc);
        const int au[1] = {l15 * LDU + lg * 4};
        gemm_rows16_deep<4, KIN / 16, 4>(acc, sm + au[0], rsrc, voff, in_soff, (KIN / 16) * 1024, g_in);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int col = (wave * 4 + n) * 16 + l15;
            const float bv = b_in[n];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = acc[0][n][e] + bv;
                Xs[(lg * 4 + e) * LDX + col] = v;
                if (part == 0 && s0 + lg * 4 + e < B) slot_rows[(size_t)(s0 + lg * 4 + e) * f2::T * ru::SLOT + col] = v;
            }
        }
    }
    DeepRing<3, 4> g_qkv;   // (in flight across the barrier)
    deep_prefetch<3, 4>(g_qkv, rsrc, voff, q_soff, 16 * 1024);
    __syncthreads();
    {
        f32x4 acc[1][3];
        zero_acc2<1, 3>(acc);
        const int ax[1] = {ru::TILE * LDU + l15 * LDX + lg * 4};
        gemm_rows16_deep<3, 16, 4>(acc, sm + ax[0], rsrc, voff, q_soff, 16 * 1024, g_qkv);
#pragma unroll
        for (int n = 0; n < 3; ++n) {
            const int col = (nb0 + n) * 16 + l15;                      // 0 .. 767: q | k | v, the slot's floats 256 ..
            const float bv = b_qkv[n];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (s0 + lg * 4 + e < B) slot_rows[(size_t)(s0 + lg * 4 + e) * f2::T * ru::SLOT + D + col] = acc[0][n][e] + bv;
        }
    }
    if (blockIdx.x == 0 && tid == 0) reinterpret_cast<int*>(cache)[slot] = c;
}

size_t reuse_cache_floats(int B) { return ru::HDR + (size_t)B * f2::T * ru::SLOT; }

hipError_t launch_reuse_update(const Dims& d, const float* fused_w, const float* x_imu, const float* x_s, float* cache, const int* frame_ctr,
                               int frame_idx, int B, int T, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    const int tiles = (B + ru::TILE - 1) / ru::TILE;
    hipLaunchKernelGGL(reuse_update_kernel, dim3(tiles * ru::PARTS), dim3(ru::THREADS), 0, s, fused_w, x_imu, x_s, cache, frame_ctr, frame_idx,
                       B, T, d.n_imu_total, d.S, (int)(fused_packed_floats(d) * 4));
    return hipGetLastError();
}

hipError_t launch_fused_encoder2(const Dims& d, const float* fused_w, const float* x_imu, const float* x_s,
                                 const float* keep_mask, float keep_scale, float* ih_out, float* hall_sentinel, int B,
                                 int num_cus, hipStream_t s, const float* reuse_cache, const int* frame_ctr, int frame_idx) {
    if (B <= 0) return hipSuccess;
    static PerDeviceFlag attr_flag; bool& attr_set = attr_flag.cur();
    if (!attr_set) {
        for (const void* f : {reinterpret_cast<const void*>(fused_encoder2_kernel<false>), reinterpret_cast<const void*>(fused_encoder2_kernel<true>),
                              reinterpret_cast<const void*>(fused_encoder2_kernel<false, true>)}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, f2::LDS_BYTES);
            if (e != hipSuccess) return e;
        }
        attr_set = true;
    }
    static int trace = -1;
    if (trace < 0) trace = tip_env("TIP_FUSED2_TRACE") ? 1 : 0;
    const int npairs = (B + 1) / 2;
    const int grid = npairs < num_cus ? npairs : num_cus;
    const size_t ih_off = f2::LAYER0 + (size_t)d.L * f2::LAYER_FLOATS;
    ReuseArgs rua;
    rua.cache = reuse_cache; rua.frame_ctr = frame_ctr; rua.frame_idx = frame_idx;
    if (reuse_cache)
        hipLaunchKernelGGL((fused_encoder2_kernel<false, true>), dim3(grid), dim3(f2::THREADS), f2::LDS_BYTES, s, fused_w, x_imu, x_s, keep_mask,
                           keep_scale, ih_out, reinterpret_cast<unsigned*>(hall_sentinel), B, d.n_imu_total, d.S, d.L,
                           (int)(fused_packed_floats(d) * 4), (int)(ih_off * 4), rua);
    else if (trace)
        hipLaunchKernelGGL(fused_encoder2_kernel<true>, dim3(grid), dim3(f2::THREADS), f2::LDS_BYTES, s, fused_w, x_imu, x_s, keep_mask,
                           keep_scale, ih_out, reinterpret_cast<unsigned*>(hall_sentinel), B, d.n_imu_total, d.S, d.L,
                           (int)(fused_packed_floats(d) * 4), (int)(ih_off * 4), rua);
    else
        hipLaunchKernelGGL(fused_encoder2_kernel<false>, dim3(grid), dim3(f2::THREADS), f2::LDS_BYTES, s, fused_w, x_imu, x_s, keep_mask,
                           keep_scale, ih_out, reinterpret_cast<unsigned*>(hall_sentinel), B, d.n_imu_total, d.S, d.L,
                           (int)(fused_packed_floats(d) * 4), (int)(ih_off * 4), rua);
    return hipGetLastError();
}

bool fused1s_fits(int B, int num_cus) { return B >= 1 && 2 * B <= num_cus && B <= 128; }
// one window on FOUR workgroups (the fused1s plan takes this form when it fits; TIP_OPT_F1S_PARTS = 2 keeps two)
bool fused1s_quad_fits(int B, int num_cus) { return B >= 1 && 4 * B <= num_cus && B <= 64; }

// ONE window per two (or four) workgroups (TIP_PLAN_FUSED1S).  Only NWIN = 1 is instantiated: the window-PAIR form (NWIN = 2, the pair-split
// plan of round 1) was retired in round 6 — the kernel template keeps the parameter, its 80-row paths are dead code the compiler drops.
template <int NWIN, int NPART = 2>
static hipError_t launch_fused_encoder_split(const Dims& d, const float* fused_w, const float* x_imu, const float* x_s,
                                             const float* keep_mask, float keep_scale, float* ih_out, float* hall_sentinel, float* xchg,
                                             int B, int num_cus, const Guard& gd, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    static_assert(NWIN == 1, "the window-pair form was retired (round 6)");
    if (!(NPART == 4 ? fused1s_quad_fits(B, num_cus) : fused1s_fits(B, num_cus))) return hipErrorInvalidValue;   // every workgroup must be resident: partners wait for each other
    static PerDeviceFlag attr_flag; bool& attr_set = attr_flag.cur();
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fused_encoder2s_kernel<NWIN, NPART>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, f2::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    constexpr int RBK = NWIN == 2 ? 5 : 3;
    constexpr size_t PAIR_IMG = (size_t)2 * NPART * RBK * 16 * f2::D;
    const int npairs = (B + NWIN - 1) / NWIN;
    const size_t xbytes = (size_t)npairs * (PAIR_IMG + f2::PAIR_FLAG_WORDS) * sizeof(float);
    if (xbytes > 0x7fffffffull || xbytes > fused2s_xchg_floats(B) * sizeof(float)) return hipErrorInvalidValue;
    // hand-off counters and XCC words (behind all images) start from zero every forward
    hipError_t e = hipMemsetAsync(xchg + (size_t)npairs * PAIR_IMG, 0, (size_t)npairs * f2::PAIR_FLAG_WORDS * 4, s);
    if (e != hipSuccess) return e;
    constexpr int BLK = 8 * NPART;
    const int grid = (NPART * npairs + BLK - 1) / BLK * BLK;           // whole (xcd, j) blocks of the id -> (pair, part) map
    {
        // the runtime's own answer to "how many of these workgroups fit on a CU" (what a cooperative launch would check): the
        // 2 * npairs working ones must all be resident (surplus ids leave at once)
        static PerDeviceInt occ_dev; int& occ = occ_dev.cur();
        hipError_t ce = check_coresident(fused_encoder2s_kernel<NWIN, NPART>, f2::THREADS, (size_t)f2::LDS_BYTES, NPART * npairs, num_cus, &occ);
        if (ce != hipSuccess) return ce;
    }
    const size_t ih_off = f2::LAYER0 + (size_t)d.L * f2::LAYER_FLOATS;
    hipLaunchKernelGGL((fused_encoder2s_kernel<NWIN, NPART>), dim3(grid), dim3(f2::THREADS), f2::LDS_BYTES, s, fused_w, x_imu, x_s, keep_mask,
                       keep_scale, ih_out, reinterpret_cast<unsigned*>(hall_sentinel), xchg, (int)xbytes, B, d.n_imu_total, d.S, d.L,
                       (int)(fused_packed_floats(d) * 4), (int)(ih_off * 4), gd);
    return hipGetLastError();
}

hipError_t launch_fused_encoder1s(const Dims& d, const float* fused_w, const float* x_imu, const float* x_s,
                                  const float* keep_mask, float keep_scale, float* ih_out, float* hall_sentinel, float* xchg,
                                  int B, int num_cus, int parts, const Guard& gd, hipStream_t s) {
    if (parts == 4 && !fused1s_quad_fits(B, num_cus)) return hipErrorInvalidValue;
    if (parts != 2 && fused1s_quad_fits(B, num_cus))
        return launch_fused_encoder_split<1, 4>(d, fused_w, x_imu, x_s, keep_mask, keep_scale, ih_out, hall_sentinel, xchg, B, num_cus, gd, s);
    return launch_fused_encoder_split<1>(d, fused_w, x_imu, x_s, keep_mask, keep_scale, ih_out, hall_sentinel, xchg, B, num_cus, gd, s);
}

// general plan: bias (+ residual) (+ ReLU)
template <int FLAGS>   // 1 = relu, 2 = residual
struct PgEpi {
    const float* bias;
    const float* res;
    float* C;
    int ldres, ldc;
    typedef f32x4 Col;
    typedef f32x4 Aux;
    __device__ __forceinline__ Col begin(int col) const { return *reinterpret_cast<const f32x4*>(bias + col); }
    __device__ __forceinline__ Aux load(int row, int col) const {
        if (FLAGS & 2) return *reinterpret_cast<const f32x4*>(res + (size_t)row * ldres + col);
        return (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ void store(int row, int col, f32x4 v, const Col& b, const Aux& r) const {
        v = v + b;
        if (FLAGS & 2) v = v + r;
        if (FLAGS & 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
        }
        *reinterpret_cast<f32x4*>(C + (size_t)row * ldc + col) = v;
    }
};

template <int FLAGS>
__global__ __launch_bounds__(pg::THREADS) void pgemm_kernel(const float* __restrict__ A, int lda, const float* __restrict__ wfrag,
                                                            int wbytes, const float* __restrict__ bias, const float* __restrict__ res,
                                                            int ldres, float* __restrict__ C, int ldc, int M, int N, int K) {
    const PgEpi<FLAGS> epi{bias, res, C, ldres, ldc};
    pgemm_body(A, lda, wfrag, wbytes, M, N, K, epi);
}

bool pgemm_shape_ok(int M, int N, int K) { return pgemm_ok(M, N, K); }

static unsigned long long g_pgemm_launches = 0;   // host-side count (tests prove which GEMM kernel a configuration took)

hipError_t launch_pgemm(const float* A, int lda, const float* wfrag, size_t wfrag_floats, const float* bias, const float* res, int ldres,
                        float* C, int ldc, int M, int N, int K, int flags, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    if (!pgemm_ok(M, N, K) || wfrag_floats * 4 > 0x7fffffffULL) return hipErrorInvalidValue;
    ++g_pgemm_launches;
    static PerDeviceFlag attr_flag; bool& attr_set = attr_flag.cur();
    if (!attr_set) {
        for (const void* f : {reinterpret_cast<const void*>(pgemm_kernel<0>), reinterpret_cast<const void*>(pgemm_kernel<1>),
                              reinterpret_cast<const void*>(pgemm_kernel<2>), reinterpret_cast<const void*>(pgemm_kernel<3>)}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, pg::LDS_BYTES);
            if (e != hipSuccess) return e;
        }
        attr_set = true;
    }
    const dim3 grid(N / pg::COLS, (M + pg::ROWS - 1) / pg::ROWS), block(pg::THREADS);
    const int wb = (int)(wfrag_floats * 4);
    switch (flags & 3) {
        case 0: hipLaunchKernelGGL(pgemm_kernel<0>, grid, block, pg::LDS_BYTES, s, A, lda, wfrag, wb, bias, res, ldres, C, ldc, M, N, K); break;
        case 1: hipLaunchKernelGGL(pgemm_kernel<1>, grid, block, pg::LDS_BYTES, s, A, lda, wfrag, wb, bias, res, ldres, C, ldc, M, N, K); break;
        case 2: hipLaunchKernelGGL(pgemm_kernel<2>, grid, block, pg::LDS_BYTES, s, A, lda, wfrag, wb, bias, res, ldres, C, ldc, M, N, K); break;
        default: hipLaunchKernelGGL(pgemm_kernel<3>, grid, block, pg::LDS_BYTES, s, A, lda, wfrag, wb, bias, res, ldres, C, ldc, M, N, K); break;
    }
    return hipGetLastError();
}

}  // namespace tip

extern "C" int tip_debug_pgemm_launches(unsigned long long* out) {
    if (!out) return -1;
    *out = tip::g_pgemm_launches;
    return 0;
}

extern "C" int tip_debug_read_f2s_cross_xcd(unsigned* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tip::g_f2s_cross_xcd), sizeof(unsigned)) == hipSuccess ? 0 : -5;
}

extern "C" int tip_debug_read_f2_trace(unsigned long long* out, int n) {
    if (!out || n < 0 || n > 64) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tip::g_f2_trace), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -5;
}

extern "C" int tip_debug_read_f2s_trace(unsigned long long* out, int n) {
    if (!out || n < 0 || n > 32) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tip::g_f2s_trace), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -5;
}

// tip_latency.hip — the few-stream ("latency") execution plan, B <= 64 windows, paper configuration.
//
// The fused plan gives one window to one CU (0.53 ms): fine for throughput, 10x too slow for a single 60-Hz stream.  Here ONE
// window is spread over 16-64 CUs per stage: every workgroup owns ONE 16-column block of a stage's output, its 4 waves split K,
// partial accumulators meet in LDS.  Stages are cut at every all-to-all seam:
//
//   prologue   :63-78 (NaN scrub, root-velocity columns, keep mask, concatenation)                   8 workgroups per window  (one-launch form)
//   in         in_linear (:79, shuffle folded); the chain's lat_in_kernel gathers the prologue itself 16
//   per layer: qkv+attn  [LN2 of the previous layer] + Q/K/V projection of ONE head + its attention   16 (one per head)
//              out       out-projection + residual (pre-LN1 sum)                                      16
//              ffn1      LN1 + linear1 + ReLU                                                         64
//              ffn2      linear2 + residual (pre-LN2 sum), K = 1024                                   16
//   ih         LN2 of the last layer + RNN input projection                                           32
//   rnn        tanh recurrence (:98-99) as a VALU GEMV (M = 1 per stream: MFMA would idle 15/16 rows): 8 (chain) / 16 (role)
//              workgroups per stream keep W_hh in VGPRs and exchange the 512-float hidden vector through 8-byte {step tag, value}
//              granules (the data is the flag; no fence, no counter)
//   head       output projection (:102)                                                               9
//
// Two forms of the SAME stage bodies (`template <bool FLOW>`: same arithmetic, same summation orders, same bits):
//   * one launch (lat_flow_kernel, round 6, B <= 24): every stage is a ROLE of one launch, all workgroups of a window on the window's
//     XCD, stage-to-stage hand-off through that XCD's L2 (0.86 us per hop; a kernel boundary costs 2.6-3.4 us on this part and a
//     hand-off across XCDs 2.6-2.9: tools/probes/l2_probe.hip, dataflow_probe.hip) — see the comment above lat_flow_kernel;
//   * a chain of 20 launches over the whole device (24 < B <= 64, CU-masked streams, fault bit 3).
//
// LayerNorm is never a stage of its own: producers store the pre-norm sum, every consumer re-normalises the rows it stages
// (40 x 256, ~1 us) and the first workgroup publishes (mean, rstd) per row for the stages that only need the normalised residual
// of their own 16 columns.
// Weights: the fused plan's fragment-ordered image (tip_fused.hip) — nothing is packed twice.
#include "tip_internal.h"
#include "tip_attention.h"
#include "tip_layernorm.h"

namespace tip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kLatMaxBatch = 64;        // streams the few-stream plan serves
namespace lz {
constexpr int D = 256, DH = 16, F = 1024, R = 512, RP = 48, RB = 3, KIN = 224;
constexpr int LDX = D + 4, LDU = KIN + 4;
}  // namespace lz
constexpr int kRnnGemvMaxMembers = 8;   // workgroups per stream of the GEMV recurrence (XCC words per stream in the workspace)

__device__ __forceinline__ float wsum64(float v) { return wave64_sum(v); }

__device__ __forceinline__ f32x4 ldfrag(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}

// The training step's dropout for the latency plan (tip_forward_dropout): the four sites of an encoder layer draw from the same
// counter-based hash, with the same element indices, as the fused training forward (fused_encoder_h_kernel<., TR>, tip_fused.hip) —
// the keep decisions do not depend on which plan computes the forward.  thresh 0 = off.
struct LatDrop {
    unsigned long long seed = 0;
    unsigned thresh = 0;
    float scale = 1.f;
    unsigned site = 0;      // 4 * layer + {0: attention probabilities, 1: out-projection output, 2: hidden after ReLU, 3: linear2 output}
};

// ---- dataflow form (lat_flow_kernel below): the stages of a forward as ROLES of one launch ------------------------------------
// A stage body is written once, for both forms.  FLOW = false: a kernel of its own (the launch chain: stream order is the
// dependency, plain loads).  FLOW = true: a role inside lat_flow_kernel — its producers ran DURING this launch on other CUs of
// the same XCD, so (i) it waits for their completion flags, (ii) it reads what they wrote with L1-bypassing loads (the L2 of the
// XCD is the coherence point: plain stores acknowledged by the L2 are visible to every CU's sc1 loads), (iii) it publishes its
// own flag behind `s_waitcnt vmcnt(0)` + a workgroup barrier.  Weights never change during a forward: plain loads in both forms.
typedef unsigned long long u64;
__device__ unsigned g_spin_timeouts_latency;   // see tip_spin_timeouts()
struct FlowCtx {
    u64* wflags = nullptr;     // this window's completion flags [stage][64]
    u64 want = 0;              // what a finished workgroup of THIS launch on THIS XCD stores: (epoch << 8) | (xcc id + 1)
    u64 publish = 0;           // what THIS workgroup stores (= want, unless TIP_OPT_FAULT_INJECT bit 4 makes it pose as another XCD's)
    unsigned spin_lim = 0;
    unsigned* err = nullptr;
    bool poisoned = false;     // a wait gave up (or a producer sits on another XCD): everything this workgroup stores is NaN from here on
};
// the activation workspace of a launch as one buffer resource (FLOW loads are buffer loads with the sc1 bit: offsets from `base`)
struct Act {
    __amdgpu_buffer_rsrc_t rs;
    const float* base;
};
template <bool FLOW>
__device__ __forceinline__ float4 ld_act4(const Act& a, const float* p) {
    if (FLOW)
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(a.rs, (int)(reinterpret_cast<const char*>(p) - reinterpret_cast<const char*>(a.base)), 0, 16));
    return *reinterpret_cast<const float4*>(p);
}
template <bool FLOW>
__device__ __forceinline__ float ld_act1(const Act& a, const float* p) {
    if (FLOW)
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(a.rs, (int)(reinterpret_cast<const char*>(p) - reinterpret_cast<const char*>(a.base)), 0, 16));
    return *p;
}
// all G workgroups of `stage` (this window) done?  One poll = one 8-byte load per lane of wave 0.  `s_ok`: one LDS word.
__device__ __forceinline__ void flow_wait(FlowCtx& fc, int stage, int G, int* s_ok) {
    const int tid = threadIdx.x;
    if (tid < 64) {
        const u64* f = fc.wflags + (size_t)stage * 64;
        bool ok = false, foreign = false;
        const unsigned lim = fc.poisoned ? 1u : fc.spin_lim;
        for (unsigned spin = 0; spin < lim; ++spin) {
            const u64 v = tid < G ? __hip_atomic_load(f + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : fc.want;
            if (__builtin_amdgcn_ballot_w64(v != fc.want) == 0) { ok = true; break; }
            // this launch's stamp from ANOTHER XCD: the placement rule (id % 8 = XCD) does not hold here — no point in waiting
            if (__builtin_amdgcn_ballot_w64(v != fc.want && (v >> 8) == (fc.want >> 8)) != 0) { foreign = true; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (tid == 0) {
            *s_ok = ok ? 1 : 0;
            if (!ok && !fc.poisoned) {
                atomicAdd(&g_spin_timeouts_latency, 1u);
                // word 1 of the handle's error block: "only the one-launch form's placement" (hosts then fall back to the launch chain,
                // TIP_OPT_NO_FLOW, instead of giving up every cooperating kernel); a producer that never arrives is word 0 like any other
                guard_report(fc.err ? fc.err + (foreign ? 1 : 0) : nullptr);
            }
        }
    }
    __syncthreads();
    if (!*s_ok) fc.poisoned = true;
    __syncthreads();            // (s_ok is reused by the next wait)
}
__device__ __forceinline__ void flow_done(const FlowCtx& fc, int stage, int nb) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's stores are acknowledged by the L2 ...
    __syncthreads();                                   // ... before the one flag store that publishes them
    if (threadIdx.x == 0) fc.wflags[(size_t)stage * 64 + nb] = fc.publish;
}
__device__ __forceinline__ float poison_if(bool p, float v) { return p ? __uint_as_float(kPoisonBits) : v; }

// Stage a [T x 256] activation into LDS [48][260] (rows >= T zero) and optionally LayerNorm it on the way
// (eps 1e-5, biased variance).  Row statistics go to `stats_out` ([48][2] = mean, rstd) when non-null.
// Sixteen lanes per row (a wave takes four rows per pass, each lane 16 columns): the two statistics are a local sum plus four
// DPP adds each (tip_layernorm.h) instead of six ds_bpermute round trips per statistic and row.
template <int THREADS, bool FLOW = false>
__device__ __forceinline__ void stage_rows_ln(float* Xs, const float* __restrict__ src, int T, const float* __restrict__ g,
                                              const float* __restrict__ be, float* __restrict__ stats_out, const Act& act = Act{}) {
    using namespace lz;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NW = THREADS / 64, NPASS = (RP + 4 * NW - 1) / (4 * NW);
    const int q = lane & 15, sub = lane >> 4;
    // all of this wave's rows are requested before the first reduction: one L2 round trip, not NPASS of them
    float4 v[NPASS][4];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int row = (p * NW + wave) * 4 + sub;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[p][j] = row < T ? ld_act4<FLOW>(act, src + (size_t)row * D + (q + 16 * j) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (g) {
        float4 gg[4], bb[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            gg[j] = *reinterpret_cast<const float4*>(g + (q + 16 * j) * 4);
            bb[j] = *reinterpret_cast<const float4*>(be + (q + 16 * j) * 4);
        }
#pragma unroll
        for (int p = 0; p < NPASS; ++p) {
            const int row = (p * NW + wave) * 4 + sub;
            float s = ((v[p][0].x + v[p][0].y) + (v[p][0].z + v[p][0].w)) + ((v[p][1].x + v[p][1].y) + (v[p][1].z + v[p][1].w));
            s += ((v[p][2].x + v[p][2].y) + (v[p][2].z + v[p][2].w)) + ((v[p][3].x + v[p][3].y) + (v[p][3].z + v[p][3].w));
            const float mean = row16_sum(s) * (1.f / D);
            float qs[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[p][j].x -= mean; v[p][j].y -= mean; v[p][j].z -= mean; v[p][j].w -= mean;
                qs[j] = (v[p][j].x * v[p][j].x + v[p][j].y * v[p][j].y) + (v[p][j].z * v[p][j].z + v[p][j].w * v[p][j].w);
            }
            const float var = row16_sum((qs[0] + qs[1]) + (qs[2] + qs[3])) * (1.f / D);
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[p][j].x = v[p][j].x * rstd * gg[j].x + bb[j].x; v[p][j].y = v[p][j].y * rstd * gg[j].y + bb[j].y;
                v[p][j].z = v[p][j].z * rstd * gg[j].z + bb[j].z; v[p][j].w = v[p][j].w * rstd * gg[j].w + bb[j].w;
                if (row >= T) v[p][j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (stats_out && q == 0 && row < T) { stats_out[row * 2] = mean; stats_out[row * 2 + 1] = rstd; }
        }
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int row = (p * NW + wave) * 4 + sub;
        if (row < RP)
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<float4*>(Xs + row * LDX + (q + 16 * j) * 4) = v[p][j];
    }
}

// This wave's K slice (k-blocks kb0 .. kb0+nkb) of ONE 16-column block: weights first (they do not depend on the
// activations, so callers request them BEFORE staging / normalising the rows), MFMAs later.
template <int KBW>
__device__ __forceinline__ void load_kslice(f32x4 (&w)[KBW], __amdgpu_buffer_rsrc_t rsrc, int voff, int soff, int nkb) {
#pragma unroll
    for (int k = 0; k < KBW; ++k) w[k] = k < nkb ? ldfrag(rsrc, voff, soff + k * 1024) : (f32x4){0.f, 0.f, 0.f, 0.f};
}

template <int KBW>
__device__ __forceinline__ void mma_kslice(f32x4 (&acc)[lz::RB], const float* As /* + l15*lda + lg*4 + kb0*16 */, int lda,
                                           const f32x4 (&w)[KBW], int nkb) {
#pragma unroll
    for (int k = 0; k < KBW; ++k) {
        if (k < nkb) {
            float4 a[lz::RB];
#pragma unroll
            for (int r = 0; r < lz::RB; ++r) a[r] = *reinterpret_cast<const float4*>(As + r * 16 * lda + k * 16);
#pragma unroll
            for (int r = 0; r < lz::RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].x, w[k].x, acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < lz::RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].y, w[k].y, acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < lz::RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].z, w[k].z, acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < lz::RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r].w, w[k].w, acc[r], 0, 0, 0);
        }
    }
}

// Cross-wave reduction of the K-split partials: wave r (< 3) ends up with the full row-block r of the 48x16 tile
// in the MFMA C layout (row = 16 r + 4 (lane>>4) + e, col = lane & 15).  red: [NW][3][64][4] floats.
template <int NW>
__device__ __forceinline__ f32x4 reduce_partials(float* red, const f32x4 (&acc)[lz::RB], int wave, int lane) {
#pragma unroll
    for (int r = 0; r < lz::RB; ++r) *reinterpret_cast<f32x4*>(red + ((wave * 3 + r) * 64 + lane) * 4) = acc[r];
    __syncthreads();
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (wave < lz::RB) {
#pragma unroll
        for (int w = 0; w < NW; ++w) s += *reinterpret_cast<const f32x4*>(red + ((w * 3 + wave) * 64 + lane) * 4);
    }
    return s;
}

// ---- lat_in: prologue + in_linear --------------------------------------------------------------------------
// LDS (floats): U [48][228] + red [4][3][256]
constexpr int kLatInSmem = lz::RP * lz::LDU + 4 * 3 * 256;
struct LatInArgs {
    const float* wts; int wbytes;
    const float* x_imu; const float* x_s; const float* keep_mask; float keep_scale;
    float* xpre; int T, NI, S, in_w_off_b, in_b_off;
    unsigned long long* gran; unsigned* xcc_words; unsigned mkey, mthresh;
    const float* ubuf;       // one-launch form: the window rows [T][224] as the prologue role left them (:63-78 applied); null: gather here
};
// The value the prologue (:63-78) puts at column c of row `row` of window `win`: x_imu | NaN-scrubbed, kept, scaled x_s | zero pad.
// ONE expression for the launch chain's gather (lat_in_body) and the one-launch form's prologue role: same bits.
__device__ __forceinline__ float lat_prologue_value(const LatInArgs& a, int win, int row, int c, bool skip_hash) {
    const int T = a.T, NI = a.NI, S = a.S;
    const float* xi = a.x_imu + (size_t)win * T * NI;
    const float* xs = a.x_s + (size_t)win * T * S;
    const float* km = a.keep_mask ? a.keep_mask + (size_t)win * T * S : nullptr;
    // One unconditional load per element from a SELECTED (clamped, always valid) address, the result masked by selects: with
    // `if (row < T) { if (c < NI) .. else if (c < NI + S) .. }` around each of the 48 loads the compiler emitted 160 exec-mask
    // branches and waited for each load inside its own region.
    const int rc = row < T ? row : T - 1;
    const bool imu = c < NI;
    const int cs = c - NI < 0 ? 0 : (c - NI < S ? c - NI : S - 1);
    const float* pa = imu ? xi + (size_t)rc * NI + c : xs + (size_t)rc * S + cs;
    const float x = *pa;
    float kv = 1.f;
    if (km) kv = *(imu ? pa : km + (size_t)rc * S + cs);                  // :77 (km: wave-uniform; IMU lanes read a dummy)
    // ... or the keep decision drawn here (tip_forward_dropout with a state seed): element index of x_s [B][T][S], the
    // same decisions tip_draw_keep_mask writes out
    // (32-bit index arithmetic: the hash takes the index mod 2^32 anyway; chunks that hold IMU columns only — NI >= 64 — skip it)
    if (a.mthresh && !skip_hash)
        kv = tip_drop_hash_k(a.mkey, ((unsigned)win * (unsigned)T + (unsigned)rc) * (unsigned)S + (unsigned)cs) >= a.mthresh ? 1.f : 0.f;
    const float xs_v = (x != x ? 0.f : x) * kv * ((km || a.mthresh) ? a.keep_scale : 1.f); // :65, then (x * mask) * scale as before
    return (row < T && c < NI + S) ? (imu ? x : xs_v) : 0.f;
}

// ---- prologue role of the one-launch form: 8 workgroups write the window's rows [T][224] ONCE (the chain's lat_in gathers them in
// each of its 16 workgroups: 48 scalar loads + up to 36 hashes per lane — 4.5 us of the in_linear stage, 9.8 us with the keep mask drawn) ----
__device__ __forceinline__ void lat_pre_role(const LatInArgs& a, float* ubuf, int p, int win) {
    using namespace lz;
    const int c = threadIdx.x;
    if (c < KIN) {
        float v[(RP + 7) / 8];
#pragma unroll
        for (int i = 0; i < (RP + 7) / 8; ++i) v[i] = lat_prologue_value(a, win, p + 8 * i, c, false);
#pragma unroll
        for (int i = 0; i < (RP + 7) / 8; ++i)
            if (p + 8 * i < a.T) ubuf[((size_t)win * a.T + p + 8 * i) * KIN + c] = v[i];
    }
}

struct NoWait { __device__ void operator()() const {} };
template <bool FLOW, typename WAIT = NoWait>
__device__ __forceinline__ void lat_in_body(const LatInArgs& a, int nb, int win, float* smem, const FlowCtx* fcp, unsigned long long* tr = nullptr,
                                            const Act* act = nullptr, WAIT&& wait = WAIT()) {
    using namespace lz;
    float* U = smem;
    float* red = smem + RP * LDU;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const float* __restrict__ wts = a.wts;
    const int T = a.T, NI = a.NI;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wts), 0, a.wbytes, 0x00020000);
    constexpr int KB = KIN / 16;  // 14 k-blocks: waves take 4,4,4,2
    const int kb0 = wave * 4;
    const int nkb = KB - kb0 < 4 ? KB - kb0 : 4;
    f32x4 w[4];
    load_kslice<4>(w, rsrc, lane * 16, a.in_w_off_b + (nb * KB + kb0) * 1024, nkb);
    const float bv_in = wts[a.in_b_off + nb * 16 + l15];
    // the first column block also clears this stream's RNN hand-off granules (tags must start at 0 every launch)
    if (nb == 0) {
        unsigned long long* gq = a.gran + (size_t)win * 2 * R;
        for (int i = tid; i < 2 * R; i += 256) gq[i] = 0ull;
        if (tid < kRnnGemvMaxMembers) a.xcc_words[win * kRnnGemvMaxMembers + tid] = 0u;
    }
    if (FLOW) {
        wait();                                                            // (weights and bias are on their way)
        // the rows as the prologue role wrote them: 16-byte loads, not from L1 (another CU of this XCD produced them during this launch)
        constexpr int F4 = KIN / 4;                                        // 56 float4 per row
        const float* ub = a.ubuf + (size_t)win * T * KIN;
        float4 v[(RP * F4 + 255) / 256];
#pragma unroll
        for (int i = 0; i < (RP * F4 + 255) / 256; ++i) {
            const int f = tid + i * 256, row = f / F4, c4 = f - row * F4;
            v[i] = (f < RP * F4 && row < T) ? ld_act4<true>(*act, ub + (size_t)row * KIN + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < (RP * F4 + 255) / 256; ++i) {
            const int f = tid + i * 256, row = f / F4, c4 = f - row * F4;
            if (f < RP * F4) *reinterpret_cast<float4*>(U + row * LDU + c4 * 4) = v[i];
        }
        if (tid < RP) *reinterpret_cast<float4*>(U + tid * LDU + KIN) = make_float4(0.f, 0.f, 0.f, 0.f);   // (the pad columns 224..227)
    } else {
    // window inputs -> U[row][0:NI | NI:NI+S | zero pad]; wave w stages rows w, w+4, ...; lanes walk the columns.
    // All global loads of a wave are requested before its first LDS store.
        constexpr int RPW = RP / 4, NCH = (KIN + 4 + 63) / 64;   // 12 rows per wave, 4 column chunks of 64
        float v[RPW][NCH];
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) v[i][ch] = lat_prologue_value(a, win, wave + i * 4, ch * 64 + lane, (ch + 1) * 64 <= NI);
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int c = ch * 64 + lane;
                if (c < LDU) U[(wave + i * 4) * LDU + c] = v[i][ch];
            }
    }
    __syncthreads();
    if (tr && tid == 0) tr[0] = __builtin_amdgcn_s_memtime();
    f32x4 acc[RB] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    mma_kslice<4>(acc, U + l15 * LDU + lg * 4 + kb0 * 16, LDU, w, nkb);
    if (tr && tid == 0) tr[1] = __builtin_amdgcn_s_memtime();
    const f32x4 s = reduce_partials<4>(red, acc, wave, lane);
    if (tr && tid == 0) tr[2] = __builtin_amdgcn_s_memtime();
    if (wave < RB) {
        const int col = nb * 16 + l15;
        const float bv = bv_in;
        float* o = a.xpre + (size_t)win * T * D;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = wave * 16 + lg * 4 + e;
            if (row < T) o[(size_t)row * D + col] = FLOW ? poison_if(fcp->poisoned, s[e] + bv) : s[e] + bv;
        }
    }
}
__global__ __launch_bounds__(256) void lat_in_kernel(LatInArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[kLatInSmem];
    lat_in_body<false>(a, blockIdx.x, blockIdx.y, smem, nullptr);
}

// ---- generic "LN(optional) -> one 16-column block of X W^T" stage: QKV, FFN1 (+ReLU), RNN-ih ----------------
//   xpre [B][T][256] pre-norm activations; g/be: LayerNorm applied while staging (null: none);
//   out [B][T][ldo]; stats [B][48][2] written by column-block 0 (null: skip)
// LDS (floats): Xs [48][260] + red [4][3][256]
constexpr int kLatLnSmem = lz::RP * lz::LDX + 4 * 3 * 256;
struct LatLnArgs {
    const float* wts; int wbytes; const float* xpre; const float* g; const float* be;
    int w_off_b, b_off; float* out; int ldo; float* stats; int T; LatDrop dr;
};
// WAIT: called (FLOW only) after the weight loads are requested and before the first activation load
// NCB: 16-column blocks per workgroup.  (Measured in the one-launch form, FFN1 on 32 workgroups with two blocks each so that FFN2's
// workgroups become resident early: the stage's body 4.2 -> 5.1-7.3 us, forward 153.5 -> 157.6 us on the same box: not used.)
template <bool RELU, bool BLOCKED_OUT, bool DROP, bool FLOW, int NCB = 1, typename WAIT>
__device__ __forceinline__ void lat_ln_gemm_body(const LatLnArgs& a, int nb0, int win, float* smem, const Act& act, FlowCtx& fc, WAIT&& wait) {
    using namespace lz;
    float* Xs = smem;
    float* red = smem + RP * LDX;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const float* __restrict__ wts = a.wts;
    const int T = a.T, ldo = a.ldo;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wts), 0, a.wbytes, 0x00020000);
    const int kb0 = wave * 4;  // K = 256: 16 k-blocks, 4 per wave
    f32x4 w[NCB][4];
    float bv[NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        load_kslice<4>(w[cb], rsrc, lane * 16, a.w_off_b + ((nb0 * NCB + cb) * 16 + kb0) * 1024, 4);
        bv[cb] = wts[a.b_off + (nb0 * NCB + cb) * 16 + l15];   // requested with the operands, not behind the reduction's barrier
    }
    if (FLOW) wait();
    stage_rows_ln<256, FLOW>(Xs, a.xpre + (size_t)win * T * D, T, a.g, a.be,
                             (a.stats && nb0 == 0) ? a.stats + (size_t)win * RP * 2 : nullptr, act);
    __syncthreads();
    const LatDrop dr = a.dr;
    const unsigned dkey = DROP ? tip_drop_key_s(dr.seed, dr.site) : 0u;
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
        const int nb = nb0 * NCB + cb;
        f32x4 acc[RB] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        mma_kslice<4>(acc, Xs + l15 * LDX + lg * 4 + kb0 * 16, LDX, w[cb], 4);
        if (cb > 0) __syncthreads();                   // (the previous block's partials have been read)
        const f32x4 s = reduce_partials<4>(red, acc, wave, lane);
        if (wave < RB) {
            const int col = nb * 16 + l15;
            float* o = a.out + (size_t)win * T * ldo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = wave * 16 + lg * 4 + e;
                float v = s[e] + bv[cb];
                if (RELU) v = fmaxf(v, 0.f);
                if (DROP && dr.thresh)   // element (global row, column) of the [M][ldo] activation, as the fused training forward counts it
                    v = tip_drop_hash_k(dkey, (unsigned)(((size_t)win * T + row) * ldo + col)) >= dr.thresh ? v * dr.scale : 0.f;
                if (FLOW) v = poison_if(fc.poisoned, v);
                if (row < T) {
                    if (BLOCKED_OUT) o[((size_t)nb * T + row) * 16 + l15] = v;   // [N/16][T][16] for the fragment-shaped consumer
                    else o[(size_t)row * ldo + col] = v;
                }
            }
        }
    }
}
__device__ __forceinline__ Act null_act() { return Act{__builtin_amdgcn_make_buffer_rsrc(static_cast<float*>(nullptr), 0, 0, 0x00020000), nullptr}; }
template <bool RELU, bool BLOCKED_OUT, bool DROP = false>
__global__ __launch_bounds__(256) void lat_ln_gemm_kernel(LatLnArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[kLatLnSmem];
    FlowCtx fc;
    lat_ln_gemm_body<RELU, BLOCKED_OUT, DROP, false>(a, blockIdx.x, blockIdx.y, smem, null_act(), fc, [] {});
}

// ---- lat_qkv_attn: [pending LN] + Q/K/V projection of ONE head + its causal attention ---------------------------
//   grid (16 heads, B).  The head's three 16-column blocks are computed with K split over the 4 waves, reduced
//   into LDS planes, then the 4 waves split the keys (lane = query) and the partial softmax states are merged.
//   Output O in K-blocked layout [16 heads][T][16] (head = k-block of the out-projection).
// LDS (floats): Xs [48][260] + red [4][3][256] + Qs, Ks [48][20] + Vts [16][52]
constexpr int kLatQkvSmem = lz::RP * lz::LDX + 4 * 3 * 256 + 2 * lz::RP * (lz::DH + 4) + lz::DH * (lz::RP + 4);
struct LatQkvArgs {
    const float* wts; int wbytes; const float* xpre; const float* g; const float* be;
    int w_off_b, b_off; float* o_out; float* stats; int T; LatDrop dr;
};
template <bool DROP, bool FLOW, typename WAIT>
__device__ __forceinline__ void lat_qkv_attn_body(const LatQkvArgs& a, int head, int win, float* smem, const Act& act, FlowCtx& fc, WAIT&& wait) {
    using namespace lz;
    float* Xs = smem;
    float* red = Xs + RP * LDX;            // K-split partials
    float* Qs = red + 4 * 3 * 256;
    float* Ks = Qs + RP * (DH + 4);
    float* Vts = Ks + RP * (DH + 4);       // V transposed
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const float* __restrict__ wts = a.wts;
    const int T = a.T;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wts), 0, a.wbytes, 0x00020000);
    const int kb0 = wave * 4;
    f32x4 w[3][4];   // Q, K, V column block of this head: nb = which*16 + head
#pragma unroll
    for (int which = 0; which < 3; ++which)
        load_kslice<4>(w[which], rsrc, lane * 16, a.w_off_b + ((which * 16 + head) * 16 + kb0) * 1024, 4);
    float bqkv[3];                                     // requested with the operands, not behind each reduction's barrier
#pragma unroll
    for (int which = 0; which < 3; ++which) bqkv[which] = wts[a.b_off + (which * 16 + head) * 16 + l15];
    if (FLOW) wait();
    stage_rows_ln<256, FLOW>(Xs, a.xpre + (size_t)win * T * D, T, a.g, a.be,
                             (a.stats && head == 0) ? a.stats + (size_t)win * RP * 2 : nullptr, act);
    __syncthreads();
#pragma unroll
    for (int which = 0; which < 3; ++which) {
        f32x4 acc[RB] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        mma_kslice<4>(acc, Xs + l15 * LDX + lg * 4 + kb0 * 16, LDX, w[which], 4);
        const f32x4 sres = reduce_partials<4>(red, acc, wave, lane);
        if (wave < RB) {
            const float bv = bqkv[which];
            if (which < 2) {
                float* dst = which == 0 ? Qs : Ks;
#pragma unroll
                for (int e = 0; e < 4; ++e) dst[(wave * 16 + lg * 4 + e) * (DH + 4) + l15] = sres[e] + bv;
            } else {
                *reinterpret_cast<f32x4*>(Vts + l15 * (RP + 4) + wave * 16 + lg * 4) = sres + bv;
            }
        }
        __syncthreads();
    }
    // attention on the matrix cores, ONE QUERY ROW BLOCK PER WAVE (waves 0-2: 1 / 2 / 3 key blocks; one wave for all three took ~2 us
    // of the stage's 6.8).  A wave reads its own Q rows and the K / V rows at or below them and writes O over its own Q rows only.
    {
        const LatDrop dr = a.dr;
        const unsigned dk = DROP ? tip_drop_key_s(dr.seed, dr.site) : 0u;
        const unsigned long long bh = (unsigned long long)win * 16 + head;
        if (wave == 0) attention_head_mfma<DH + 4, RP + 4, DROP, 0, 1>(Qs, Ks, Vts, 0, lane, 48, dk, dr.thresh, dr.scale, bh, T);
        else if (wave == 1) attention_head_mfma<DH + 4, RP + 4, DROP, 1, 2>(Qs, Ks, Vts, 0, lane, 48, dk, dr.thresh, dr.scale, bh, T);
        else if (wave == 2) attention_head_mfma<DH + 4, RP + 4, DROP, 2, 3>(Qs, Ks, Vts, 0, lane, 48, dk, dr.thresh, dr.scale, bh, T);
    }
    __syncthreads();
    // O (in the Q plane) -> K-blocked global layout [16 heads][T][16]
    {
        const int row = tid >> 2, c0 = (tid & 3) * 4;
        if (row < T) {
            float4 ov = *reinterpret_cast<const float4*>(Qs + row * (DH + 4) + c0);
            if (FLOW && fc.poisoned) ov = make_float4(__uint_as_float(kPoisonBits), __uint_as_float(kPoisonBits), __uint_as_float(kPoisonBits), __uint_as_float(kPoisonBits));
            *reinterpret_cast<float4*>(a.o_out + (size_t)win * T * D + ((size_t)head * T + row) * 16 + c0) = ov;
        }
    }
}
template <bool DROP = false>
__global__ __launch_bounds__(256) void lat_qkv_attn_kernel(LatQkvArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[kLatQkvSmem];
    FlowCtx fc;
    lat_qkv_attn_body<DROP, false>(a, blockIdx.x, blockIdx.y, smem, null_act(), fc, [] {});
}

// ---- residual GEMMs: out[row][col] = resid(xpre)[row][col] + (A W^T + b)[row][col] ---------------------------
//   resid = LayerNorm(xpre) via the published row statistics when g != null, else xpre itself.
//   NW waves split K (= 16*KBT k-blocks); A is read straight from global in fragment shape, from a K-BLOCKED
//   layout [K/16][T][16] written by the producer kernel: the 16 rows x 64 B a wave touches per load are one
//   contiguous KiB (a row-major [T][K] image would put the 16 rows 1-4 KB apart on the same channel).
//   PARTS > 1 (dataflow form, 256-thread workgroups): wave w carries the K slices of the chain form's waves w * PARTS .. + PARTS - 1,
//   one after the other, in accumulators of their own; the partials meet in the chain form's order — bit-identical results.
// LDS (floats): red [NW * PARTS][3][256]
struct LatResArgs {
    const float* wts; int wbytes; const float* A; int lda; const float* xpre; const float* stats; const float* g; const float* be;
    int w_off_b, b_off; float* out; int T; LatDrop dr;
};
template <int NW, int KBT, int PARTS, bool DROP, bool FLOW, typename WAIT>
__device__ __forceinline__ void lat_res_gemm_body(const LatResArgs& a, int nb, int win, float* red, const Act& act, FlowCtx& fc, WAIT&& wait) {
    using namespace lz;
    constexpr int KBW = KBT / (NW * PARTS);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const float* __restrict__ wts = a.wts;
    const int T = a.T;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wts), 0, a.wbytes, 0x00020000);
    f32x4 w[PARTS][KBW];
#pragma unroll
    for (int p = 0; p < PARTS; ++p)
#pragma unroll
        for (int k = 0; k < KBW; ++k) w[p][k] = ldfrag(rsrc, lane * 16, a.w_off_b + (nb * KBT + (wave * PARTS + p) * KBW + k) * 1024);
    // The epilogue's inputs (bias, LayerNorm parameters / statistics, residual values) are requested HERE, with the operands: loaded
    // where they are used — behind the reduction's barrier — they are one more exposed L2 round trip in each of the forward's 18 kernels.
    const int ecol = nb * 16 + l15;
    const int ewave = wave < RB ? wave : 0;               // (waves >= RB load a dummy row block: keeps the loads unconditional)
    const float bv = wts[a.b_off + ecol];
    const float* g = a.g;
    const float gc = g ? g[ecol] : 1.f, bc = g ? a.be[ecol] : 0.f;
    if (FLOW) wait();
    float xres[4], smean[4], srstd[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = ewave * 16 + lg * 4 + e, rc = row < T ? row : T - 1;
        xres[e] = ld_act1<FLOW>(act, a.xpre + (size_t)win * T * D + (size_t)rc * D + ecol);
        smean[e] = g ? ld_act1<FLOW>(act, a.stats + (size_t)win * RP * 2 + rc * 2) : 0.f;
        srstd[e] = g ? ld_act1<FLOW>(act, a.stats + (size_t)win * RP * 2 + rc * 2 + 1) : 1.f;
    }
    const float* Ab = a.A + (size_t)win * T * a.lda + lg * 4;   // lda = K: a window's blocked image has T*K floats
#pragma unroll
    for (int p = 0; p < PARTS; ++p) {
        const int kb0 = (wave * PARTS + p) * KBW;
        float4 af[RB][KBW];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int row = r * 16 + l15;
#pragma unroll
            for (int k = 0; k < KBW; ++k)
                af[r][k] = row < T ? ld_act4<FLOW>(act, Ab + ((size_t)(kb0 + k) * T + row) * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        f32x4 acc[RB] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int k = 0; k < KBW; ++k) {
#pragma unroll
            for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r][k].x, w[p][k].x, acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r][k].y, w[p][k].y, acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r][k].z, w[p][k].z, acc[r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[r][k].w, w[p][k].w, acc[r], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) *reinterpret_cast<f32x4*>(red + (((wave * PARTS + p) * 3 + r) * 64 + lane) * 4) = acc[r];
    }
    __syncthreads();
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (wave < RB) {
#pragma unroll
        for (int wv = 0; wv < NW * PARTS; ++wv) s += *reinterpret_cast<const f32x4*>(red + ((wv * 3 + wave) * 64 + lane) * 4);
    }
    const LatDrop dr = a.dr;
    const unsigned dkey = DROP ? tip_drop_key_s(dr.seed, dr.site) : 0u;
    if (wave < RB) {
        float* o = a.out + (size_t)win * T * D;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = wave * 16 + lg * 4 + e;
            float x = xres[e];
            if (g) x = (x - smean[e]) * srstd[e] * gc + bc;
            float v = s[e] + bv;
            if (DROP && dr.thresh)
                v = tip_drop_hash_k(dkey, (unsigned)(((size_t)win * T + row) * D + ecol)) >= dr.thresh ? v * dr.scale : 0.f;
            if (row < T) o[(size_t)row * D + ecol] = FLOW ? poison_if(fc.poisoned, x + v) : x + v;
        }
    }
}
template <int NW, int KBT, bool DROP = false>
__global__ __launch_bounds__(NW * 64) void lat_res_gemm_kernel(LatResArgs a) {
    __shared__ __attribute__((aligned(16))) float red[NW * 3 * 256];
    FlowCtx fc;
    lat_res_gemm_body<NW, KBT, 1, DROP, false>(a, blockIdx.x, blockIdx.y, red, null_act(), fc, [] {});
}

// ---- lat_head: output projection (:102) for few rows: one 16-column block x 48 rows per workgroup, K split --------
// LDS (floats): red [4][3][256]
template <bool FLOW, typename WAIT>
__device__ __forceinline__ void lat_head_body(const float* __restrict__ A, long long lda, const float* __restrict__ wfrag,
                                              const float* __restrict__ bias, float* __restrict__ Y, int ldy, int M, int N, int nb, int m0,
                                              float* red, const Act& act, FlowCtx& fc, WAIT&& wait) {
    using namespace lz;
    constexpr int KBT = R / 16, KBW = KBT / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wfrag), 0, ((N + 15) / 16) * R * 64, 0x00020000);
    const int kb0 = wave * KBW;
    f32x4 w[KBW];
#pragma unroll
    for (int k = 0; k < KBW; ++k) w[k] = ldfrag(rsrc, lane * 16, (nb * KBT + kb0 + k) * 1024);
    const int col = nb * 16 + l15;
    const float bv = col < N ? bias[col] : 0.f;
    if (FLOW) wait();
    float4 a[RB][KBW];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int row = m0 + r * 16 + l15;
        const float* ap = A + (size_t)(row < M ? row : 0) * lda + kb0 * 16 + lg * 4;
#pragma unroll
        for (int k = 0; k < KBW; ++k)
            a[r][k] = row < M ? ld_act4<FLOW>(act, ap + k * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f32x4 acc[RB] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int k = 0; k < KBW; ++k) {
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][k].x, w[k].x, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][k].y, w[k].y, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][k].z, w[k].z, acc[r], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r][k].w, w[k].w, acc[r], 0, 0, 0);
    }
    const f32x4 sres = reduce_partials<4>(red, acc, wave, lane);
    if (wave < RB) {
        if (col < N) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + wave * 16 + lg * 4 + e;
                if (row < M) Y[(size_t)row * ldy + col] = FLOW ? poison_if(fc.poisoned, sres[e] + bv) : sres[e] + bv;
            }
        }
    }
}
__global__ __launch_bounds__(256) void lat_head_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ wfrag,
                                                       const float* __restrict__ bias, float* __restrict__ Y, int ldy, int M,
                                                       int N) {
    __shared__ __attribute__((aligned(16))) float red[4 * 3 * 256];
    FlowCtx fc;
    lat_head_body<false>(A, lda, wfrag, bias, Y, ldy, M, N, blockIdx.x, blockIdx.y * lz::RP, red, null_act(), fc, [] {});
}

hipError_t launch_latency_head(const float* A, long long lda, const float* wfrag, const float* bias, float* Y, int ldy,
                               int M, int N, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(lat_head_kernel, dim3((N + 15) / 16, (M + lz::RP - 1) / lz::RP), dim3(256), 0, s, A, lda, wfrag, bias, Y,
                       ldy, M, N);
    return hipGetLastError();
}

// ---- rnn_gemv: the recurrence for one stream on a cluster of NM = 4 or 8 workgroups of one XCD -------------------------------
//   A workgroup (8 waves) owns 512 / NM hidden units: wave w the RPW = 64 / NM ... rows [wg * 512/NM + w * RPW, + RPW); a row's 512-term
//   dot product is split over KS = 64 / RPW lanes (lane = ks * RPW + row-in-wave), KL = 512 / KS terms each, W_hh in VGPRs (KL per lane).
//   Why more members: per step every lane reads its KL values of h_{t-1} from LDS (b128 reads; the 16 / 8 lanes of a k-slice read the
//   same words, which costs the LDS the same 8 cycles per instruction): 8 waves x 32 reads x 8 cycles = 2 048 cycles per step at
//   NM = 4 — the step (1.16 us = 2 780 cycles) was LDS-issue bound, not hop bound.  NM = 8 halves the reads per wave.
//   Summation order (the same for every NM, so a stream's bits do not depend on how many streams share the launch): the 512 terms
//   form 16 sub-chains of 32 (k = 32 c .. 32 c + 31; four fused-multiply-add chains over the float4 components, combined
//   (x + y) + (z + w)), the sub-chains a binary tree over c: c ^ 1, c ^ 2, c ^ 4, c ^ 8 — in-lane first, then lanes ^RPW, ... ^32.
//   W_hh is read once from the MFMA-fragment image: fragment (nb, kb) lane' = 16*lg' + l15 holds W[nb*16 + l15][kb*16 + 4 lg' .. +3].
//   granules: hb[win][parity][512] of {tag = step + 1, value}; zeroed before every launch (lat_in).
template <int NM>
__global__ __launch_bounds__(512) void rnn_gemv_kernel(const float* __restrict__ ih, const float* __restrict__ whh_frag,
                                                       float* __restrict__ hall, u64* __restrict__ hb,
                                                       unsigned* __restrict__ xcc_words, int B, int T, Guard gd) {
    using namespace lz;
    constexpr int RW = R / NM, RPW = RW / 8, KS = 64 / RPW, KL = R / KS, NJ = KL / 4, SC = KL / 32;
    static_assert(NM == 4 || NM == 8, "members per stream");
    __shared__ __attribute__((aligned(16))) float hs2[2][KS * (KL + 4)];   // KS k-slices, padded: distinct banks per slice; two buffers
                                                                           // alternate, so one barrier per step suffices
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rl = lane & (RPW - 1), ks = lane / RPW;
    // the NM members of a stream's cluster are taken 8 blocks apart: observed to share an XCD (verified below)
    const int wg = (blockIdx.x >> 3) % NM, win = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * NM));
    if (win >= B) return;
    if ((gd.fault & 4) && win == 0 && wg == 1) return;   // TIP_OPT_FAULT_INJECT: this member never arrives
    const unsigned spin_lim = guard_spin_limit(gd.fault, 1u << 22);
    bool poisoned = false;   // a granule wait gave up: this thread's h words are NaN from here on and it never spins again
    const int row = wg * RW + wave * RPW + rl;
    // run-time check that all members really sit on one XCD: only then may granules be published with plain 8-byte
    // stores (they stay in the shared L2, where the peers' L1-bypassing loads see them after ~0.4 us); otherwise the
    // write-through agent-scope stores that are correct for any placement are used.
    __shared__ int s_same_xcd;
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xf;
        __hip_atomic_store(xcc_words + win * kRnnGemvMaxMembers + wg, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool same = true;
        for (int m = 0; m < NM; ++m) {
            unsigned v = 0;
            for (unsigned spins = 0; spins < spin_lim; ++spins) {
                v = __hip_atomic_load(xcc_words + win * kRnnGemvMaxMembers + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (!v) {
                atomicAdd(&g_spin_timeouts_latency, 1u);
                guard_report(gd.err);
            }
            same &= (v == xcc + 1u);
        }
        s_same_xcd = (same && !(gd.fault & 8)) ? 1 : 0;
    }
    __syncthreads();
    const bool same_xcd = s_same_xcd != 0;
    float4 w[NJ];
    {
        const float4* wf = reinterpret_cast<const float4*>(whh_frag);
        const int nb = row >> 4, l15 = row & 15;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int k = ks * KL + 4 * j;
            w[j] = wf[(size_t)(nb * 32 + (k >> 4)) * 64 + ((k & 15) >> 2) * 16 + l15];
        }
    }
    u64* hbw = hb + (size_t)win * 2 * R;
    const float* ihw = ih + (size_t)win * T * R;
    float* hw = hall + (size_t)win * T * R;
    float ih_next = ihw[row];
    for (int t = 0; t < T; ++t) {
        const float ihv = ih_next;
        float acc = 0.f;
        if (t == 0 && T > 1) ih_next = ihw[(size_t)R + row];
        if (t > 0) {
            // gather h_{t-1}: thread i polls granule i until its tag says "step t" (bounded: never hang the GPU)
            const u64* gp = hbw + (size_t)((t - 1) & 1) * R + tid;
            u64 v = 0;
            const unsigned lim = poisoned ? 1u : spin_lim;
            for (unsigned spins = 0; spins < lim; ++spins) {
                v = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(v >> 32) == (unsigned)t) break;
            }
            if ((unsigned)(v >> 32) != (unsigned)t) {
                if (!poisoned) {
                    atomicAdd(&g_spin_timeouts_latency, 1u);
                    guard_report(gd.err);
                }
                if (!poisoned && ks == 0)   // row t-1 is incomplete (the missing slice holds stale memory): no finite y row may come of it
                    hw[(size_t)(t - 1) * R + row] = __uint_as_float(kPoisonBits);
                poisoned = true;
                v = (u64)kPoisonBits;   // the word that never came is NaN: every hidden unit that reads it follows
            }
            float* hs = hs2[t & 1];
            hs[(tid / KL) * (KL + 4) + (tid % KL)] = __uint_as_float((unsigned)v);
            __syncthreads();
            // the next step's input term is requested AFTER this step's polls (vector memory returns in order: issued in front
            // of them it would hold every poll back by its own round trip) and flies during the dot products
            if (t + 1 < T) ih_next = ihw[(size_t)(t + 1) * R + row];
            const float* hq = hs + ks * (KL + 4);
            // a sub-chain = 32 terms: four chains (one per float4 component), written as two 2-wide FMAs on the ADJACENT halves of the
            // operands (the compiler's own pairing of the scalar form — (x, z) and (y, w) — cost three register moves per v_pk_fma_f32)
            typedef float f2 __attribute__((ext_vector_type(2)));
            float sub[SC];
            {
                // the sub-chains ADVANCE TOGETHER: step j of all SC chains, then step j + 1 — SC independent reads and 2 SC independent
                // FMAs per step instead of one dependent chain at a time (written chain by chain the compiler waits for every LDS read
                // on its own at NM = 4: 181 against 163 us per B = 1 forward, same box)
                f2 a01[SC], a23[SC];
#pragma unroll
                for (int c = 0; c < SC; ++c) { a01[c] = (f2){0.f, 0.f}; a23[c] = (f2){0.f, 0.f}; }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float4 hv[SC];
#pragma unroll
                    for (int c = 0; c < SC; ++c) hv[c] = *reinterpret_cast<const float4*>(hq + (c * 8 + j) * 4);
#pragma unroll
                    for (int c = 0; c < SC; ++c) {
                        a01[c] = __builtin_elementwise_fma((f2){w[c * 8 + j].x, w[c * 8 + j].y}, (f2){hv[c].x, hv[c].y}, a01[c]);
                        a23[c] = __builtin_elementwise_fma((f2){w[c * 8 + j].z, w[c * 8 + j].w}, (f2){hv[c].z, hv[c].w}, a23[c]);
                    }
                }
#pragma unroll
                for (int c = 0; c < SC; ++c) sub[c] = (a01[c][0] + a01[c][1]) + (a23[c][0] + a23[c][1]);
            }
            // the tree over the 16 sub-chains: in-lane levels, then the lanes of the row (lane ^ RPW, ... ^ 32: DPP / permlane swaps,
            // no LDS round trips on the serial chain)
            if (SC == 4) acc = (sub[0] + sub[1]) + (sub[2] + sub[3]);
            else acc = sub[0] + sub[SC - 1];
            static_assert(SC == 4 || SC == 2, "sub-chains per lane");
            if (RPW == 8) acc += dpp_peer<0x128>(acc);   // row_ror:8 = lane ^ 8 inside a 16-lane row
            acc = lg4_sum(acc);                          // lane ^ 16, lane ^ 32
        }
        if (ks == 0) {
            const float hv = tip_tanh(acc + ihv);
            const u64 gran = ((u64)(unsigned)(t + 1) << 32) | (u64)__float_as_uint(hv);
            u64* gdst = hbw + (size_t)(t & 1) * R + row;
            if (same_xcd) *gdst = gran;   // one aligned 8-byte store: single-copy atomic, lands in the shared L2
            else __hip_atomic_store(gdst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hw[(size_t)t * R + row] = hv;   // (after the granule: the partners wait for that one)
        }
    }
}

// ---- the recurrence as a role of lat_flow_kernel: 16 workgroups of 4 waves per stream (the 8-member kernel's wave, twice as many
// workgroups: 32 hidden units each), W_hh in registers BEFORE the input terms exist, h_{t-1} through the same {tag, value} granules.
// Same summation order as rnn_gemv_kernel (16 sub-chains of 32, tree c ^ 1, c ^ 2, c ^ 4, c ^ 8): same bits.
constexpr int kFlowRnnMembers = 16;
template <typename WAIT>
__device__ __forceinline__ void rnn_flow_role(const float* __restrict__ ih, const float* __restrict__ whh_frag, float* __restrict__ hall,
                                              u64* __restrict__ hb, int win, int wg, int T, float* smem, const Act& act, FlowCtx& fc,
                                              const Guard& gd, WAIT&& wait, unsigned long long* trace = nullptr) {
    using namespace lz;
    constexpr int RW = R / kFlowRnnMembers, RPW = RW / 4, KS = 64 / RPW, KL = R / KS, NJ = KL / 4, SC = KL / 32;
    static_assert(RPW == 8 && SC == 2, "the 8-member kernel's wave shape");
    float* hs2 = smem;                                       // [2][KS * (KL + 4)]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rl = lane & (RPW - 1), ks = lane / RPW;
    const int row = wg * RW + wave * RPW + rl;
    float4 w[NJ];
    {
        const float4* wf = reinterpret_cast<const float4*>(whh_frag);
        const int nb = row >> 4, l15 = row & 15;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int k = ks * KL + 4 * j;
            w[j] = wf[(size_t)(nb * 32 + (k >> 4)) * 64 + ((k & 15) >> 2) * 16 + l15];
        }
    }
    wait();                                                  // the input terms of every step (RNN input projection: 32 workgroups)
    const unsigned spin_lim = fc.spin_lim;
    bool poisoned = fc.poisoned;
    u64* hbw = hb + (size_t)win * 2 * R;
    const float* ihw = ih + (size_t)win * T * R;
    float* hw = hall + (size_t)win * T * R;
    float ih_next = ld_act1<true>(act, ihw + row);
    for (int t = 0; t < T; ++t) {
        const float ihv = ih_next;
        float acc = 0.f;
        if (t == 0 && T > 1) ih_next = ld_act1<true>(act, ihw + (size_t)R + row);
        if (t > 0) {
            float* hs = hs2 + (t & 1) * (KS * (KL + 4));
            {                                                 // 512 granules, 256 threads: two per thread, polled TOGETHER
                const u64* gp = hbw + (size_t)((t - 1) & 1) * R + tid;
                u64 v0 = 0, v1 = 0;
                const unsigned lim = poisoned ? 1u : spin_lim;
                if (trace && t == 20 && tid == 0) trace[0] = __builtin_amdgcn_s_memtime();
                for (unsigned spins = 0; spins < lim; ++spins) {
                    v0 = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    v1 = __hip_atomic_load(gp + 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(v0 >> 32) == (unsigned)t && (unsigned)(v1 >> 32) == (unsigned)t) break;
                }
                if (trace && t == 20 && tid == 0) trace[1] = __builtin_amdgcn_s_memtime();
                if ((unsigned)(v0 >> 32) != (unsigned)t || (unsigned)(v1 >> 32) != (unsigned)t) {
                    if (!poisoned) {
                        atomicAdd(&g_spin_timeouts_latency, 1u);
                        guard_report(gd.err);
                    }
                    poisoned = true;
                    v0 = v1 = (u64)kPoisonBits;
                }
                hs[(tid / KL) * (KL + 4) + (tid % KL)] = __uint_as_float((unsigned)v0);
                hs[((tid + 256) / KL) * (KL + 4) + (tid % KL)] = __uint_as_float((unsigned)v1);
            }
            __syncthreads();
            if (trace && t == 20 && tid == 0) trace[2] = __builtin_amdgcn_s_memtime();
            if (t + 1 < T) ih_next = ld_act1<true>(act, ihw + (size_t)(t + 1) * R + row);
            const float* hq = hs + ks * (KL + 4);
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 a01[SC], a23[SC];
#pragma unroll
            for (int c = 0; c < SC; ++c) { a01[c] = (f2){0.f, 0.f}; a23[c] = (f2){0.f, 0.f}; }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float4 hv[SC];
#pragma unroll
                for (int c = 0; c < SC; ++c) hv[c] = *reinterpret_cast<const float4*>(hq + (c * 8 + j) * 4);
#pragma unroll
                for (int c = 0; c < SC; ++c) {
                    a01[c] = __builtin_elementwise_fma((f2){w[c * 8 + j].x, w[c * 8 + j].y}, (f2){hv[c].x, hv[c].y}, a01[c]);
                    a23[c] = __builtin_elementwise_fma((f2){w[c * 8 + j].z, w[c * 8 + j].w}, (f2){hv[c].z, hv[c].w}, a23[c]);
                }
            }
            acc = ((a01[0][0] + a01[0][1]) + (a23[0][0] + a23[0][1])) + ((a01[1][0] + a01[1][1]) + (a23[1][0] + a23[1][1]));
            acc += dpp_peer<0x128>(acc);   // row_ror:8 = lane ^ 8 inside a 16-lane row
            acc = lg4_sum(acc);            // lane ^ 16, lane ^ 32
            if (trace && t == 20 && tid == 0) trace[3] = __builtin_amdgcn_s_memtime();
        }
        if (ks == 0) {
            const float hv = poison_if(poisoned, tip_tanh(acc + ihv));
            hbw[(size_t)(t & 1) * R + row] = ((u64)(unsigned)(t + 1) << 32) | (u64)__float_as_uint(hv);   // one aligned 8-byte store into the XCD's L2
            hw[(size_t)t * R + row] = hv;
        }
        if (trace && (t == 20 || t == 21) && tid == 0) trace[4 + (t - 20)] = __builtin_amdgcn_s_memtime();
    }
    fc.poisoned = poisoned;
}

// ---- lat_flow: the stages IN .. output projection as ROLES of one launch -----------------------------------------------------
// Why: a kernel boundary costs ~3 us on this part (tools/probes/l2_probe.hip: an empty kernel 2.6 us, a 16-KiB dependent one 3.4) and
// the few-stream forward crosses 18 of them for ~0.5 us of matrix work each.  An in-launch hand-off ACROSS XCDs costs as much
// (write-through stores, memory-side flags: 2.6-2.9 us per stage, tools/probes/dataflow_probe.hip — what the persistent kernel of
// round 4 found), but inside ONE XCD the L2 is the coherence point and a stage-to-stage hop is 0.86 us.  So: every workgroup of a
// window runs on the window's XCD (workgroup ids go round-robin over the 8 XCDs: id % 8 = XCD, checked at run time through the
// flags), every stage has its OWN workgroups (weights in registers long before the inputs exist), and the ids are stage-major: a
// workgroup only ever waits for LOWER ids, which the dispatcher has placed before it — no co-residency requirement.
// Window w lives on XCD w % 8; the windows of one XCD share its stage-major list.
// Stage numbers: 0 prologue (8 workgroups) | 1 in_linear | 2 + 4 l: QKV + attention, 3 + 4 l: out-projection, 4 + 4 l: FFN1, 5 + 4 l: FFN2 |
// 2 + 4 L: RNN input projection | 3 + 4 L: recurrence (16 workgroups per stream) | 4 + 4 L: output projection (one workgroup per 16 output columns).
// What stamps a launch's flags: (per-handle nonce + launch counter) where the COUNTER LIVES IN THE WORKSPACE (behind the flags) and is
// advanced by the launch itself — a kernel argument would be frozen into a captured HIP graph and every replay would find the
// previous replay's flags "done".  One counter PER WINDOW (the XCD lanes dispatch their lists independently of each other: a counter
// for the whole grid could be advanced by one lane before another lane's workgroups have read it).  Every workgroup reads its
// window's counter at entry; the window's LAST workgroup in dispatch order (output projection, last column block: every other
// workgroup of the window has an id below its own on the same lane, i.e. was dispatched before it, and it finishes microseconds after
// it started) stores counter + 1 when it is done.  A workgroup that nevertheless read another value than its producers waits in
// vain: time-out, NaN rows, TIP_ERR_HANDOFF — never a wrong number.
constexpr int kFlowSmem = kLatQkvSmem;   // the largest role (floats)
static_assert(kLatInSmem <= kFlowSmem && kLatLnSmem <= kFlowSmem && 8 * 3 * 256 <= kFlowSmem && 2 * 8 * 68 <= kFlowSmem, "role LDS");
constexpr int kFlowMaxStages = 2 + 4 * 8;
// The flags and launch counters of the one-launch form, [64 windows][34][64] u64 = 1.06 MiB, live at the very FRONT of the caller's
// workspace (tip_abi.hip carve_workspace), at offsets that depend on nothing — not on B, not on T.  A counter must only ever be
// advanced by its own window's launches: behind the activations (where round 6 first put them) the words moved with every (B, T), a
// counter of one layout was overwritten by another layout's small counter values, and a window found the stamps of ITS OWN earlier
// launch "current" — finite-but-wrong rows, 6 times in 138 000 calls of tools/flow_soak.py.
size_t latency_flow_flag_floats() { return (size_t)kLatMaxBatch * kFlowMaxStages * 64 * 2; }
struct LatFlowArgs {
    const float* wts; int wbytes;
    const float* x_imu; const float* x_s; const float* keep_mask; float keep_scale;
    float* ws; int ws_bytes;               // the latency workspace (activation loads of the roles are offsets from it)
    float *xa, *xb, *o, *hid, *ihb, *st0, *st1, *ubuf;
    unsigned long long* gran; unsigned* xccw;
    u64* flags;                            // [B][kFlowMaxStages][64]; a window's slot kFlowMaxStages - 1 holds its launch counter
    u64 nonce;                             // per-handle constant mixed into the stamps
    const float* whh_frag; float* hall;    // recurrence role
    const float* out_frag; const float* out_bias; float* y; int ldy, N, last_only;   // output projection role
    int B, T, NI, S, L;
    unsigned mkey, mthresh;
    LatDrop dr;
    Guard gd;
    int trace;                             // measurement (TIP_FLOW_TRACE=1): stamp g_flow_trace
};
__device__ unsigned long long g_flow_trace[kFlowMaxStages * 4];   // measurement: s_memtime of (entry, inputs ready, results stored, flag published) per stage, window 0 / workgroup 0
#define FLOW_STAMP(k) do { if (a.trace && win == 0 && nb == 0 && threadIdx.x == 0) g_flow_trace[stage * 4 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
template <bool DROP>
__global__ __launch_bounds__(256) void lat_flow_kernel(LatFlowArgs a) {
    using namespace lz;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int s_ok;
    // ---- who am I: XCD lane x = id % 8, position p in that XCD's stage-major list ----
    const int x = blockIdx.x & 7;
    int p = blockIdx.x >> 3;
    const int nw = (a.B - x + 7) >> 3;                  // windows on this XCD: x, x + 8, ...
    if (nw <= 0) return;
    const int L = a.L, nenc = 3 + 4 * L, nstage = nenc + 2;
    const int nhead = (a.N + 15) >> 4;
    int stage = -1, j = 0, nb = 0;
    for (int st = 0; st < nstage; ++st) {
        const int G = st == 0 ? 8 : st == 1 ? 16 : st == nenc - 1 ? 32 : st == nenc ? kFlowRnnMembers : st == nenc + 1 ? nhead : (((st - 2) & 3) == 2 ? 64 : 16);
        const int cnt = G * nw;
        if (p < cnt) { stage = st; j = p / G; nb = p - j * G; break; }
        p -= cnt;
    }
    if (stage < 0) return;                               // (grid padding)
    const int win = x + 8 * j;
    if ((a.gd.fault & 1) && win == 0 && stage == 3 && nb == 1) return;   // TIP_OPT_FAULT_INJECT bit 0: this producer (out-projection of layer 0, column block 1) never arrives
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    FlowCtx fc;
    fc.wflags = a.flags + (size_t)win * kFlowMaxStages * 64;
    u64* const counter = fc.wflags + (size_t)(kFlowMaxStages - 1) * 64;   // this WINDOW's launch counter (the last flag slot: no stage uses it)
    const u64 launch = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    fc.want = (((a.nonce + launch) & 0x00ffffffffffffffull) << 8) | (u64)((xcc & 0xf) + 1u);
    fc.publish = fc.want;
    // TIP_OPT_FAULT_INJECT bit 4: this producer (out-projection of layer 0, column block 1) stamps its flag as another XCD's would
    if ((a.gd.fault & 16) && win == 0 && stage == 3 && nb == 1) fc.publish = (fc.want & ~0xffull) | (u64)(((xcc + 1u) & 0xf) + 1u);
    fc.spin_lim = guard_spin_limit(a.gd.fault, 1u << 22);
    fc.err = a.gd.err;
    const Act act{__builtin_amdgcn_make_buffer_rsrc(a.ws, 0, a.ws_bytes, 0x00020000), a.ws};
    const int T = a.T;
    FLOW_STAMP(0);
    // offsets inside the fused section (tip_fused.hip)
    constexpr size_t IN_W = 0, IN_B = (size_t)D * KIN, LAYER0 = IN_B + D;
    constexpr size_t QKV_W = 0, QKV_B = QKV_W + (size_t)3 * D * D, WO_W = QKV_B + 3 * D, WO_B = WO_W + (size_t)D * D;
    constexpr size_t W1_W = WO_B + D, W1_B = W1_W + (size_t)F * D, W2_W = W1_B + F, W2_B = W2_W + (size_t)D * F;
    constexpr size_t G1 = W2_B + D, BE1 = G1 + D, G2 = BE1 + D, BE2 = G2 + D, LAYER_FLOATS = BE2 + D;
    if (stage <= 1) {
        LatInArgs ia{a.wts, a.wbytes, a.x_imu, a.x_s, a.keep_mask, a.keep_scale, a.xa, T, a.NI, a.S, (int)(IN_W * 4), (int)IN_B,
                     a.gran, a.xccw, a.mkey, a.mthresh, a.ubuf};
        if (stage == 0) {                                 // prologue (:63-78): the window's rows, once
            FLOW_STAMP(1);
            lat_pre_role(ia, a.ubuf, nb, win);
        } else {                                          // in_linear (:79, shuffle folded); its weights are requested in front of the wait
            lat_in_body<true>(ia, nb, win, smem, &fc, (a.trace && win == 0 && nb == 0) ? g_flow_trace + (nenc + 2) * 4 + 8 : nullptr, &act,
                              [&] { flow_wait(fc, 0, 8, &s_ok); FLOW_STAMP(1); });
        }
        FLOW_STAMP(2);
        flow_done(fc, stage, nb);
        FLOW_STAMP(3);
        return;
    }
    if (stage == nenc) {                                  // the recurrence
        if ((a.gd.fault & 4) && win == 0 && nb == 1) return;   // TIP_OPT_FAULT_INJECT bit 2: this member never arrives
        rnn_flow_role(a.ihb, a.whh_frag, a.hall, a.gran, win, nb, T, smem, act, fc, a.gd, [&] { flow_wait(fc, nenc - 1, 32, &s_ok); FLOW_STAMP(1); },
                      (a.trace && win == 0 && nb == 0) ? g_flow_trace + (nenc + 2) * 4 : nullptr);
        FLOW_STAMP(2);
        flow_done(fc, stage, nb);
        FLOW_STAMP(3);
        return;
    }
    if (stage == nenc + 1) {                              // output projection: every row of the window, or row T - 1 (real_time_runner_minimal.py:150)
        const float* hwin = a.hall + (size_t)win * T * R;
        const Act hact{__builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hwin), 0, T * R * 4, 0x00020000), hwin};
        const float* hA = a.last_only ? hwin + (size_t)(T - 1) * R : hwin;
        float* yw = a.last_only ? a.y + (size_t)win * a.ldy : a.y + (size_t)win * T * a.ldy;
        lat_head_body<true>(hA, R, a.out_frag, a.out_bias, yw, a.ldy, a.last_only ? 1 : T, a.N, nb, 0, smem, hact, fc,
                            [&] { flow_wait(fc, nenc, kFlowRnnMembers, &s_ok); FLOW_STAMP(1); });
        FLOW_STAMP(2);
        // (the outputs leave the launch: no flag; the designated workgroup advances the launch counter for the next launch)
        if (nb == nhead - 1 && threadIdx.x == 0) __hip_atomic_store(counter, launch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    if (stage == nenc - 1) {                              // LN2 of the last layer + RNN input projection
        const size_t lo = LAYER0 + (size_t)(L - 1) * LAYER_FLOATS, ih_off = LAYER0 + (size_t)L * LAYER_FLOATS;
        LatLnArgs la{a.wts, a.wbytes, a.xa, a.wts + lo + G2, a.wts + lo + BE2, (int)(ih_off * 4), (int)(ih_off + (size_t)R * D),
                     a.ihb, R, nullptr, T, LatDrop{}};
        lat_ln_gemm_body<false, false, false, true>(la, nb, win, smem, act, fc, [&] { flow_wait(fc, stage - 1, 16, &s_ok); FLOW_STAMP(1); });
        FLOW_STAMP(2);
        flow_done(fc, stage, nb);
        FLOW_STAMP(3);
        return;
    }
    const int l = (stage - 2) >> 2, role = (stage - 2) & 3;
    const size_t lo = LAYER0 + (size_t)l * LAYER_FLOATS;
    const float* LW = a.wts + lo;
    const float* pg = l > 0 ? a.wts + lo - LAYER_FLOATS + G2 : nullptr;   // LayerNorm pending on the residual stream (norm2 of the previous layer)
    const float* pb = l > 0 ? a.wts + lo - LAYER_FLOATS + BE2 : nullptr;
    LatDrop dr = a.dr;
    dr.site = (unsigned)l * 4u + (unsigned)role;
    if (role == 0) {
        LatQkvArgs qa{a.wts, a.wbytes, a.xa, pg, pb, (int)((lo + QKV_W) * 4), (int)(lo + QKV_B), a.o, a.st0, T, dr};
        lat_qkv_attn_body<DROP, true>(qa, nb, win, smem, act, fc, [&] { flow_wait(fc, stage - 1, 16, &s_ok); FLOW_STAMP(1); });
    } else if (role == 1) {
        LatResArgs ra{a.wts, a.wbytes, a.o, D, a.xa, a.st0, pg, pb, (int)((lo + WO_W) * 4), (int)(lo + WO_B), a.xb, T, dr};
        lat_res_gemm_body<4, 16, 1, DROP, true>(ra, nb, win, smem, act, fc, [&] { flow_wait(fc, stage - 1, 16, &s_ok); FLOW_STAMP(1); });
    } else if (role == 2) {
        LatLnArgs la{a.wts, a.wbytes, a.xb, LW + G1, LW + BE1, (int)((lo + W1_W) * 4), (int)(lo + W1_B), a.hid, F, a.st1, T, dr};
        lat_ln_gemm_body<true, true, DROP, true>(la, nb, win, smem, act, fc, [&] { flow_wait(fc, stage - 1, 16, &s_ok); FLOW_STAMP(1); });
    } else {
        LatResArgs ra{a.wts, a.wbytes, a.hid, F, a.xb, a.st1, LW + G1, LW + BE1, (int)((lo + W2_W) * 4), (int)(lo + W2_B), a.xa, T, dr};
        lat_res_gemm_body<4, 64, 2, DROP, true>(ra, nb, win, smem, act, fc, [&] { flow_wait(fc, stage - 1, 64, &s_ok); FLOW_STAMP(1); });
    }
    FLOW_STAMP(2);
    flow_done(fc, stage, nb);
    FLOW_STAMP(3);
}
#undef FLOW_STAMP

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
hipError_t read_spin_timeouts_latency(unsigned* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_spin_timeouts_latency), sizeof(unsigned));
}

bool latency_supported(const Dims& d, int B, int T) {
    return fused_supported(d, T) && fused_has_rnn_ih(d) && B >= 1 && B <= kLatMaxBatch;
}

// workspace (floats): xa, xb [B][T][256]; qkv [B][T][768]; o [B][T][256]; hid [B][T][1024]; ih [B][T][512];
//                     stats [2][B][48][2]; granules [B][2][512] u64; 8 XCC words per stream.
size_t latency_workspace_floats(int B, int T) {
    const size_t bt = (size_t)B * T;
    return bt * (256 + 256 + 768 + 256 + 1024 + 512) + (size_t)2 * B * 48 * 2 + (size_t)B * 2 * 512 * 2 +
           (size_t)B * kRnnGemvMaxMembers + 1024 + 128;
}

// B up to which the stages run as roles of ONE launch (lat_flow_kernel): a window's roles share the 32 CUs of one XCD, so windows
// beyond 8 queue behind each other on their XCD; the launch chain spreads every stage over the whole device instead.
// Measured back to back on one box (tools/b1_chain.py; one launch / chain, us per forward): B = 1 147 / 158, 8 149 / 173, 12 180 / 191,
// 16 181 / 199, 24 241 / 262, 32 298 / 286.  Both forms give the same bits (same stage bodies, same summation orders).
constexpr int kFlowMaxBatch = 24;

hipError_t launch_latency_plan(const Dims& d, const float* fused_w, const float* whh_frag, const float* x_imu,
                               const float* x_s, const float* keep_mask, float keep_scale, float* ws, float* hall, int B,
                               int T, int num_cus, const Guard& gd, hipStream_t s, const TrainDropout* td, const LatencyHead* head) {
    using namespace lz;
    const size_t bt = (size_t)B * T;
    u64* flags = head ? reinterpret_cast<u64*>(head->flags) : nullptr;
    float* xa = ws;
    float* xb = xa + bt * 256;
    float* o = xb + bt * 256 + bt * 768;   // (the QKV slot of the workspace is unused since QKV + attention merged)
    float* hid = o + bt * 256;
    float* ihb = hid + bt * 1024;
    float* st0 = ihb + bt * 512;
    float* st1 = st0 + (size_t)B * 48 * 2;
    u64* gran = reinterpret_cast<u64*>(st1 + (size_t)B * 48 * 2);   // every term above is an even float count: 8-B aligned
    unsigned* xccw = reinterpret_cast<unsigned*>(gran + (size_t)B * 2 * R);   // kRnnGemvMaxMembers words per stream
    const int wbytes = (int)(fused_packed_floats(d) * 4);
    // offsets inside the fused section (tip_fused.hip)
    constexpr size_t IN_W = 0, IN_B = (size_t)D * KIN, LAYER0 = IN_B + D;
    constexpr size_t QKV_W = 0, QKV_B = QKV_W + (size_t)3 * D * D, WO_W = QKV_B + 3 * D, WO_B = WO_W + (size_t)D * D;
    constexpr size_t W1_W = WO_B + D, W1_B = W1_W + (size_t)F * D, W2_W = W1_B + F, W2_B = W2_W + (size_t)D * F;
    constexpr size_t G1 = W2_B + D, BE1 = G1 + D, G2 = BE1 + D, BE2 = G2 + D, LAYER_FLOATS = BE2 + D;
    const bool drop = td && td->thresh;
    // measurement: TIP_LAT_FLOW=0 keeps the launch chain at every batch size, =N moves the switch-over to N windows
    static const int flow_max = tip_env("TIP_LAT_FLOW") ? atoi(tip_env("TIP_LAT_FLOW")) : kFlowMaxBatch;
    const size_t ws_bytes = latency_workspace_floats(B, T) * sizeof(float);
    if (head && head->done) *head->done = false;
    // (fault bit 3 = "partners on different XCDs": the chain's territory)
    if (head && head->flags && B <= flow_max && !(gd.fault & 8) && num_cus % 8 == 0 && num_cus >= 64 && 6 + 4 * d.L <= kFlowMaxStages &&
        ws_bytes < 0x7fffffffull && (head->N + 15) / 16 <= 64 && (size_t)T * R * 4 < 0x7fffffffull) {
        static PerDeviceFlag attr_flag; bool& attr_set = attr_flag.cur();
        if (!attr_set) {
            for (const void* f : {reinterpret_cast<const void*>(lat_flow_kernel<false>), reinterpret_cast<const void*>(lat_flow_kernel<true>)}) {
                hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kFlowSmem * (int)sizeof(float));
                if (e != hipSuccess) return e;
            }
            attr_set = true;
        }
        LatFlowArgs fa;
        fa.wts = fused_w; fa.wbytes = wbytes;
        fa.x_imu = x_imu; fa.x_s = x_s; fa.keep_mask = keep_mask; fa.keep_scale = keep_scale;
        fa.ws = ws; fa.ws_bytes = (int)ws_bytes;
        fa.xa = xa; fa.xb = xb; fa.o = o; fa.hid = hid; fa.ihb = ihb; fa.st0 = st0; fa.st1 = st1;
        fa.ubuf = xb + bt * 256;                                 // the [B][T][768] slot nobody else uses: [B][T][224] prologue rows
        fa.gran = gran; fa.xccw = xccw; fa.flags = flags;
        fa.nonce = head->nonce;
        fa.whh_frag = whh_frag; fa.hall = hall;
        fa.out_frag = head->wfrag; fa.out_bias = head->bias; fa.y = head->y; fa.ldy = head->ldy; fa.N = head->N; fa.last_only = head->last_only ? 1 : 0;
        fa.B = B; fa.T = T; fa.NI = d.n_imu_total; fa.S = d.S; fa.L = d.L;
        fa.mkey = td ? td->mkey : 0u; fa.mthresh = td && !keep_mask ? td->mthresh : 0u;
        fa.dr = LatDrop{};
        if (td) { fa.dr.seed = td->seed; fa.dr.thresh = td->thresh; fa.dr.scale = td->scale; }
        fa.gd = gd;
        static const int trace_on = (tip_env("TIP_FLOW_TRACE") && tip_env("TIP_FLOW_TRACE")[0] == '1') ? 1 : 0;
        fa.trace = trace_on;
        const int per_win = 8 + 16 + d.L * 112 + 32 + kFlowRnnMembers + (head->N + 15) / 16;   // workgroups of one window
        const int grid = 8 * per_win * ((B + 7) / 8);
        if (drop) hipLaunchKernelGGL(lat_flow_kernel<true>, dim3(grid), dim3(256), kFlowSmem * sizeof(float), s, fa);
        else hipLaunchKernelGGL(lat_flow_kernel<false>, dim3(grid), dim3(256), kFlowSmem * sizeof(float), s, fa);
        if (head->done) *head->done = true;                      // recurrence and output projection were roles of that launch
        return hipGetLastError();
    } else {
        hipLaunchKernelGGL(lat_in_kernel, dim3(16, B), dim3(256), 0, s,
                           LatInArgs{fused_w, wbytes, x_imu, x_s, keep_mask, keep_scale, xa, T, d.n_imu_total, d.S, (int)(IN_W * 4), (int)IN_B,
                                     gran, xccw, td ? td->mkey : 0u, td && !keep_mask ? td->mthresh : 0u, nullptr});
        const float* pg = nullptr;   // LayerNorm pending on the residual stream (norm2 of the previous layer)
        const float* pb = nullptr;
        for (int l = 0; l < d.L; ++l) {
            const size_t lo = LAYER0 + (size_t)l * LAYER_FLOATS;
            const float* LW = fused_w + lo;
            // xa = pre-norm input of the layer; st0 = its row statistics (when pg != null)
            LatDrop dr;
            if (td) { dr.seed = td->seed; dr.thresh = td->thresh; dr.scale = td->scale; }
            auto site = [&](unsigned k) { LatDrop x = dr; x.site = (unsigned)l * 4u + k; return x; };
            const LatQkvArgs qa{fused_w, wbytes, xa, pg, pb, (int)((lo + QKV_W) * 4), (int)(lo + QKV_B), o, st0, T, site(0)};
            const LatResArgs oa{fused_w, wbytes, o, D, xa, st0, pg, pb, (int)((lo + WO_W) * 4), (int)(lo + WO_B), xb, T, site(1)};
            const LatLnArgs f1{fused_w, wbytes, xb, LW + G1, LW + BE1, (int)((lo + W1_W) * 4), (int)(lo + W1_B), hid, F, st1, T, site(2)};   // xb = pre-LN1 sum
            const LatResArgs f2{fused_w, wbytes, hid, F, xb, st1, LW + G1, LW + BE1, (int)((lo + W2_W) * 4), (int)(lo + W2_B), xa, T, site(3)};
            if (drop) {   // tip_forward_dropout: the same stages with the training step's four dropout sites live
                hipLaunchKernelGGL(lat_qkv_attn_kernel<true>, dim3(16, B), dim3(256), 0, s, qa);
                hipLaunchKernelGGL((lat_res_gemm_kernel<4, 16, true>), dim3(16, B), dim3(256), 0, s, oa);
                hipLaunchKernelGGL((lat_ln_gemm_kernel<true, true, true>), dim3(64, B), dim3(256), 0, s, f1);
                hipLaunchKernelGGL((lat_res_gemm_kernel<8, 64, true>), dim3(16, B), dim3(512), 0, s, f2);
            } else {
                hipLaunchKernelGGL(lat_qkv_attn_kernel<false>, dim3(16, B), dim3(256), 0, s, qa);
                hipLaunchKernelGGL((lat_res_gemm_kernel<4, 16>), dim3(16, B), dim3(256), 0, s, oa);
                hipLaunchKernelGGL((lat_ln_gemm_kernel<true, true>), dim3(64, B), dim3(256), 0, s, f1);
                hipLaunchKernelGGL((lat_res_gemm_kernel<8, 64>), dim3(16, B), dim3(512), 0, s, f2);
            }
            pg = LW + G2;
            pb = LW + BE2;
        }
        const size_t ih_off = LAYER0 + (size_t)d.L * LAYER_FLOATS;
        hipLaunchKernelGGL((lat_ln_gemm_kernel<false, false>), dim3(32, B), dim3(256), 0, s,
                           LatLnArgs{fused_w, wbytes, xa, pg, pb, (int)(ih_off * 4), (int)(ih_off + (size_t)R * D), ihb, R, nullptr, T, LatDrop{}});
    }
    // members per stream: 8 while they all fit (8 B <= #CUs), else 4 — same bits either way (one canonical summation order);
    // measurement: TIP_RNN_GEMV_NM=4 pins the round-5 form
    static const int nm_env = tip_env("TIP_RNN_GEMV_NM") ? atoi(tip_env("TIP_RNN_GEMV_NM")) : 0;
    const int nm = nm_env == 4 || nm_env == 8 ? nm_env : (8 * B <= num_cus ? 8 : 4);
    if (nm == 8) {
        static PerDeviceInt occ_dev; int& occ = occ_dev.cur();   // the workgroups of every stream's cluster must be resident together
        hipError_t ce = check_coresident(rnn_gemv_kernel<8>, 512, (size_t)0, 8 * B, num_cus, &occ);
        if (ce != hipSuccess) return ce;
        hipLaunchKernelGGL(rnn_gemv_kernel<8>, dim3(64 * ((B + 7) / 8)), dim3(512), 0, s, ihb, whh_frag, hall, gran, xccw, B, T, gd);
    } else {
        static PerDeviceInt occ_dev; int& occ = occ_dev.cur();
        hipError_t ce = check_coresident(rnn_gemv_kernel<4>, 512, (size_t)0, 4 * B, num_cus, &occ);
        if (ce != hipSuccess) return ce;
        hipLaunchKernelGGL(rnn_gemv_kernel<4>, dim3(32 * ((B + 7) / 8)), dim3(512), 0, s, ihb, whh_frag, hall, gran, xccw, B, T, gd);
    }
    return hipGetLastError();
}

// The past-state keep mask (:77) as a tensor: mask[i] = hash(key, i) >= thresh ? 1 : 0 over the elements of x_s — the decisions
// lat_in_kernel draws in place when tip_forward_dropout is given the seed instead of a mask.
__global__ void keep_mask_kernel(float* __restrict__ mask, size_t n, unsigned key, unsigned thresh) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        mask[i] = tip_drop_hash_k(key, (unsigned)i) >= thresh ? 1.f : 0.f;
}
hipError_t launch_keep_mask(float* mask, size_t n, unsigned key, unsigned thresh, hipStream_t s) {
    if (!n) return hipSuccess;
    const size_t blocks = (n + 255) / 256;
    hipLaunchKernelGGL(keep_mask_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, s, mask, n, key, thresh);
    return hipGetLastError();
}

}  // namespace tip

extern "C" int tip_debug_read_flow_trace(unsigned long long* out, int n) {
    if (!out || n < 0 || n > tip::kFlowMaxStages * 4) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tip::g_flow_trace), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -5;
}

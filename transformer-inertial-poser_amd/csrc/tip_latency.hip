// tip_latency.hip — "latency" execution plan for few concurrent streams (B <= 64), paper configuration.
//
// The fused plan gives one window to one CU (0.75 ms): fine for throughput, 13x too slow for a single 60-Hz
// stream.  Here ONE window is spread over up to 64 CUs per GEMM: every workgroup owns ONE 16-column block of the
// output, its 4 (or 8) waves split K, partial accumulators meet in LDS.  Stages are cut at every all-to-all seam
// (kernel boundary ~1.5 us is cheaper on this chip than an in-kernel grid exchange, cdna_hip_programming.md 5.6):
//
//   lat_in        prologue (:63-78) + in_linear (:79, shuffle folded)           grid (16, B)
//   per layer:    lat_qkv   [LN2 of previous layer] + QKV projection            grid (48, B)
//                 lat_attn  causal attention, keys split over 4 waves           grid (16, B)
//                 lat_out   out-proj + residual (pre-LN1 sum)                   grid (16, B)
//                 lat_ffn1  LN1 + linear1 + ReLU                                grid (64, B)
//                 lat_ffn2  linear2 + residual (pre-LN2 sum), 8 waves split K   grid (16, B)
//   lat_ih        LN2 of the last layer + RNN input projection                  grid (32, B)
//   rnn_gemv      tanh recurrence (:98-99) as a VALU GEMV (M = 1 per stream: MFMA would idle 15/16 rows);
//                 4 workgroups per stream keep W_hh in VGPRs (128 per lane) and exchange the 512-float hidden
//                 vector through 8-byte {step tag, value} granules (the data is the flag; no fence, no counter)
//   head_gemm     output projection (tip_general.hip)
//
// LayerNorm is never a kernel of its own: producers store the pre-norm sum, every consumer re-normalises the
// rows it stages (40 x 256, ~1 us) and the first workgroup publishes (mean, rstd) per row for the kernels that
// only need the normalised residual of their own 16 columns.
// Weights: the fused plan's fragment-ordered image (tip_fused.hip) — nothing is packed twice.
#include "tip_internal.h"
#include "tip_attention.h"
#include "tip_layernorm.h"

namespace tip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace lz {
constexpr int D = 256, DH = 16, F = 1024, R = 512, RP = 48, RB = 3, KIN = 224;
constexpr int LDX = D + 4, LDU = KIN + 4;
}  // namespace lz

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

// Stage a [T x 256] activation into LDS [48][260] (rows >= T zero) and optionally LayerNorm it on the way
// (eps 1e-5, biased variance).  Row statistics go to `stats_out` ([48][2] = mean, rstd) when non-null.
// Sixteen lanes per row (a wave takes four rows per pass, each lane 16 columns): the two statistics are a local sum plus four
// DPP adds each (tip_layernorm.h) instead of six ds_bpermute round trips per statistic and row.
template <int THREADS>
__device__ __forceinline__ void stage_rows_ln(float* Xs, const float* __restrict__ src, int T, const float* __restrict__ g,
                                              const float* __restrict__ be, float* __restrict__ stats_out) {
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
            v[p][j] = row < T ? *reinterpret_cast<const float4*>(src + (size_t)row * D + (q + 16 * j) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
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
__global__ __launch_bounds__(256) void lat_in_kernel(const float* __restrict__ wts, int wbytes, const float* __restrict__ x_imu,
                                                     const float* __restrict__ x_s, const float* __restrict__ keep_mask,
                                                     float keep_scale, float* __restrict__ xpre, int T, int NI, int S,
                                                     int in_w_off_b, int in_b_off, unsigned long long* __restrict__ gran,
                                                     unsigned* __restrict__ xcc_words, unsigned mkey, unsigned mthresh) {
    using namespace lz;
    __shared__ __attribute__((aligned(16))) float U[RP * LDU];
    __shared__ __attribute__((aligned(16))) float red[4 * 3 * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int nb = blockIdx.x, win = blockIdx.y;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wts), 0, wbytes, 0x00020000);
    constexpr int KB = KIN / 16;  // 14 k-blocks: waves take 4,4,4,2
    const int kb0 = wave * 4;
    const int nkb = KB - kb0 < 4 ? KB - kb0 : 4;
    f32x4 w[4];
    load_kslice<4>(w, rsrc, lane * 16, in_w_off_b + (nb * KB + kb0) * 1024, nkb);
    const float bv_in = wts[in_b_off + nb * 16 + l15];
    // the first column block also clears this stream's RNN hand-off granules (tags must start at 0 every launch)
    if (nb == 0) {
        unsigned long long* gq = gran + (size_t)win * 2 * R;
        for (int i = tid; i < 2 * R; i += 256) gq[i] = 0ull;
        if (tid < 4) xcc_words[win * 4 + tid] = 0u;
    }
    // window inputs -> U[row][0:NI | NI:NI+S | zero pad]; wave w stages rows w, w+4, ...; lanes walk the columns.
    // All global loads of a wave are requested before its first LDS store.
    {
        const float* xi = x_imu + (size_t)win * T * NI;
        const float* xs = x_s + (size_t)win * T * S;
        const float* km = keep_mask ? keep_mask + (size_t)win * T * S : nullptr;
        constexpr int RPW = RP / 4, NCH = (KIN + 4 + 63) / 64;   // 12 rows per wave, 4 column chunks of 64
        float v[RPW][NCH];
#pragma unroll
        for (int i = 0; i < RPW; ++i) {
            const int row = wave + i * 4;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                // One unconditional load per element from a SELECTED (clamped, always valid) address, the result masked by selects: with
                // `if (row < T) { if (c < NI) .. else if (c < NI + S) .. }` around each of the 48 loads the compiler emitted 160 exec-mask
                // branches and waited for each load inside its own region.
                const int c = ch * 64 + lane;
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
                if (mthresh && (ch + 1) * 64 > NI)
                    kv = tip_drop_hash_k(mkey, ((unsigned)win * (unsigned)T + (unsigned)rc) * (unsigned)S + (unsigned)cs) >= mthresh ? 1.f : 0.f;
                const float xs_v = (x != x ? 0.f : x) * kv * ((km || mthresh) ? keep_scale : 1.f); // :65, then (x * mask) * scale as before
                v[i][ch] = (row < T && c < NI + S) ? (imu ? x : xs_v) : 0.f;
            }
        }
#pragma unroll
        for (int i = 0; i < RPW; ++i)
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                const int c = ch * 64 + lane;
                if (c < LDU) U[(wave + i * 4) * LDU + c] = v[i][ch];
            }
    }
    __syncthreads();
    f32x4 acc[RB] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    mma_kslice<4>(acc, U + l15 * LDU + lg * 4 + kb0 * 16, LDU, w, nkb);
    const f32x4 s = reduce_partials<4>(red, acc, wave, lane);
    if (wave < RB) {
        const int col = nb * 16 + l15;
        const float bv = bv_in;
        float* o = xpre + (size_t)win * T * D;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = wave * 16 + lg * 4 + e;
            if (row < T) o[(size_t)row * D + col] = s[e] + bv;
        }
    }
}

// ---- generic "LN(optional) -> one 16-column block of X W^T" kernel: QKV, FFN1 (+ReLU), RNN-ih ----------------
//   xpre [B][T][256] pre-norm activations; g/be: LayerNorm applied while staging (null: none);
//   out [B][T][ldo]; stats [B][48][2] written by column-block 0 (null: skip)
template <bool RELU, bool BLOCKED_OUT, bool DROP = false>
__global__ __launch_bounds__(256) void lat_ln_gemm_kernel(const float* __restrict__ wts, int wbytes, const float* __restrict__ xpre,
                                                          const float* __restrict__ g, const float* __restrict__ be,
                                                          int w_off_b, int b_off, float* __restrict__ out, int ldo,
                                                          float* __restrict__ stats, int T, LatDrop dr) {
    using namespace lz;
    __shared__ __attribute__((aligned(16))) float Xs[RP * LDX];
    __shared__ __attribute__((aligned(16))) float red[4 * 3 * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int nb = blockIdx.x, win = blockIdx.y;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wts), 0, wbytes, 0x00020000);
    const int kb0 = wave * 4;  // K = 256: 16 k-blocks, 4 per wave
    f32x4 w[4];
    load_kslice<4>(w, rsrc, lane * 16, w_off_b + (nb * 16 + kb0) * 1024, 4);
    const float bv = wts[b_off + nb * 16 + l15];   // requested with the operands, not behind the reduction's barrier
    stage_rows_ln<256>(Xs, xpre + (size_t)win * T * D, T, g, be,
                       (stats && nb == 0) ? stats + (size_t)win * RP * 2 : nullptr);
    __syncthreads();
    f32x4 acc[RB] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    mma_kslice<4>(acc, Xs + l15 * LDX + lg * 4 + kb0 * 16, LDX, w, 4);
    const f32x4 s = reduce_partials<4>(red, acc, wave, lane);
    const unsigned dkey = DROP ? tip_drop_key_s(dr.seed, dr.site) : 0u;
    if (wave < RB) {
        const int col = nb * 16 + l15;
        float* o = out + (size_t)win * T * ldo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = wave * 16 + lg * 4 + e;
            float v = s[e] + bv;
            if (RELU) v = fmaxf(v, 0.f);
            if (DROP && dr.thresh)   // element (global row, column) of the [M][ldo] activation, as the fused training forward counts it
                v = tip_drop_hash_k(dkey, (unsigned)(((size_t)win * T + row) * ldo + col)) >= dr.thresh ? v * dr.scale : 0.f;
            if (row < T) {
                if (BLOCKED_OUT) o[((size_t)nb * T + row) * 16 + l15] = v;   // [N/16][T][16] for the fragment-shaped consumer
                else o[(size_t)row * ldo + col] = v;
            }
        }
    }
}

// ---- lat_qkv_attn: [pending LN] + Q/K/V projection of ONE head + its causal attention ---------------------------
//   grid (16 heads, B).  The head's three 16-column blocks are computed with K split over the 4 waves, reduced
//   into LDS planes, then the 4 waves split the keys (lane = query) and the partial softmax states are merged.
//   Output O in K-blocked layout [16 heads][T][16] (head = k-block of the out-projection).
template <bool DROP = false>
__global__ __launch_bounds__(256) void lat_qkv_attn_kernel(const float* __restrict__ wts, int wbytes, const float* __restrict__ xpre,
                                                           const float* __restrict__ g, const float* __restrict__ be,
                                                           int w_off_b, int b_off, float* __restrict__ o_out,
                                                           float* __restrict__ stats, int T, LatDrop dr) {
    using namespace lz;
    __shared__ __attribute__((aligned(16))) float Xs[RP * LDX];
    __shared__ __attribute__((aligned(16))) float red[4 * 3 * 256];       // K-split partials; later softmax partials
    __shared__ __attribute__((aligned(16))) float Qs[RP * (DH + 4)], Ks[RP * (DH + 4)], Vts[DH * (RP + 4)];   // V transposed
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int head = blockIdx.x, win = blockIdx.y;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wts), 0, wbytes, 0x00020000);
    const int kb0 = wave * 4;
    f32x4 w[3][4];   // Q, K, V column block of this head: nb = which*16 + head
#pragma unroll
    for (int which = 0; which < 3; ++which)
        load_kslice<4>(w[which], rsrc, lane * 16, w_off_b + ((which * 16 + head) * 16 + kb0) * 1024, 4);
    float bqkv[3];                                     // requested with the operands, not behind each reduction's barrier
#pragma unroll
    for (int which = 0; which < 3; ++which) bqkv[which] = wts[b_off + (which * 16 + head) * 16 + l15];
    stage_rows_ln<256>(Xs, xpre + (size_t)win * T * D, T, g, be,
                       (stats && head == 0) ? stats + (size_t)win * RP * 2 : nullptr);
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
    // attention on the matrix cores by wave 0 (48 MFMAs + a 16-lane-shuffle softmax: ~2 us; the other waves idle)
    if (wave == 0) {
        if (DROP)
            attention_head_mfma<DH + 4, RP + 4, true>(Qs, Ks, Vts, 0, lane, 48, tip_drop_key_s(dr.seed, dr.site), dr.thresh, dr.scale,
                                                      (unsigned long long)win * 16 + head, T);
        else
            attention_head_mfma<DH + 4, RP + 4>(Qs, Ks, Vts, 0, lane);
    }
    __syncthreads();
    // O (in the Q plane) -> K-blocked global layout [16 heads][T][16]
    {
        const int row = tid >> 2, c0 = (tid & 3) * 4;
        if (row < T)
            *reinterpret_cast<float4*>(o_out + (size_t)win * T * D + ((size_t)head * T + row) * 16 + c0) =
                *reinterpret_cast<const float4*>(Qs + row * (DH + 4) + c0);
    }
}

// ---- residual GEMMs: out[row][col] = resid(xpre)[row][col] + (A W^T + b)[row][col] ---------------------------
//   resid = LayerNorm(xpre) via the published row statistics when g != null, else xpre itself.
//   NW waves split K (= 16*KBT k-blocks); A is read straight from global in fragment shape, from a K-BLOCKED
//   layout [K/16][T][16] written by the producer kernel: the 16 rows x 64 B a wave touches per load are one
//   contiguous KiB (a row-major [T][K] image would put the 16 rows 1-4 KB apart on the same channel).
template <int NW, int KBT, bool DROP = false>
__global__ __launch_bounds__(NW * 64) void lat_res_gemm_kernel(const float* __restrict__ wts, int wbytes, const float* __restrict__ A,
                                                               int lda, const float* __restrict__ xpre, const float* __restrict__ stats,
                                                               const float* __restrict__ g, const float* __restrict__ be, int w_off_b,
                                                               int b_off, float* __restrict__ out, int T, LatDrop dr) {
    using namespace lz;
    constexpr int KBW = KBT / NW;
    __shared__ __attribute__((aligned(16))) float red[NW * 3 * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int nb = blockIdx.x, win = blockIdx.y;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wts), 0, wbytes, 0x00020000);
    const int kb0 = wave * KBW;
    f32x4 w[KBW];
#pragma unroll
    for (int k = 0; k < KBW; ++k) w[k] = ldfrag(rsrc, lane * 16, w_off_b + (nb * KBT + kb0 + k) * 1024);
    // The epilogue's inputs (bias, LayerNorm parameters / statistics, residual values) are requested HERE, with the operands: loaded
    // where they are used — behind the reduction's barrier — they are one more exposed L2 round trip in each of the forward's 18 kernels.
    const int ecol = nb * 16 + l15;
    const int ewave = wave < RB ? wave : 0;               // (waves >= RB load a dummy row block: keeps the loads unconditional)
    const float bv = wts[b_off + ecol];
    const float gc = g ? g[ecol] : 1.f, bc = g ? be[ecol] : 0.f;
    float xres[4], smean[4], srstd[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int row = ewave * 16 + lg * 4 + e, rc = row < T ? row : T - 1;
        xres[e] = xpre[(size_t)win * T * D + (size_t)rc * D + ecol];
        smean[e] = g ? stats[(size_t)win * RP * 2 + rc * 2] : 0.f;
        srstd[e] = g ? stats[(size_t)win * RP * 2 + rc * 2 + 1] : 1.f;
    }
    const float* Ab = A + (size_t)win * T * lda + lg * 4;   // lda = K: a window's blocked image has T*K floats
    float4 a[RB][KBW];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int row = r * 16 + l15;
#pragma unroll
        for (int k = 0; k < KBW; ++k)
            a[r][k] = row < T ? *reinterpret_cast<const float4*>(Ab + ((size_t)(kb0 + k) * T + row) * 16)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
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
    const f32x4 s = reduce_partials<NW>(red, acc, wave, lane);
    const unsigned dkey = DROP ? tip_drop_key_s(dr.seed, dr.site) : 0u;
    if (wave < RB) {
        float* o = out + (size_t)win * T * D;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = wave * 16 + lg * 4 + e;
            float x = xres[e];
            if (g) x = (x - smean[e]) * srstd[e] * gc + bc;
            float v = s[e] + bv;
            if (DROP && dr.thresh)
                v = tip_drop_hash_k(dkey, (unsigned)(((size_t)win * T + row) * D + ecol)) >= dr.thresh ? v * dr.scale : 0.f;
            if (row < T) o[(size_t)row * D + ecol] = x + v;
        }
    }
}

// ---- lat_head: output projection (:102) for few rows: one 16-column block x 48 rows per workgroup, K split --------
__global__ __launch_bounds__(256) void lat_head_kernel(const float* __restrict__ A, long long lda, const float* __restrict__ wfrag,
                                                       const float* __restrict__ bias, float* __restrict__ Y, int ldy, int M,
                                                       int N) {
    using namespace lz;
    constexpr int KBT = R / 16, KBW = KBT / 4;
    __shared__ __attribute__((aligned(16))) float red[4 * 3 * 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int nb = blockIdx.x, m0 = blockIdx.y * RP;
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wfrag), 0, ((N + 15) / 16) * R * 64, 0x00020000);
    const int kb0 = wave * KBW;
    f32x4 w[KBW];
#pragma unroll
    for (int k = 0; k < KBW; ++k) w[k] = ldfrag(rsrc, lane * 16, (nb * KBT + kb0 + k) * 1024);
    float4 a[RB][KBW];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int row = m0 + r * 16 + l15;
        const float* ap = A + (size_t)(row < M ? row : 0) * lda + kb0 * 16 + lg * 4;
#pragma unroll
        for (int k = 0; k < KBW; ++k)
            a[r][k] = row < M ? *reinterpret_cast<const float4*>(ap + k * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
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
        const int col = nb * 16 + l15;
        if (col < N) {
            const float bv = bias[col];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = m0 + wave * 16 + lg * 4 + e;
                if (row < M) Y[(size_t)row * ldy + col] = sres[e] + bv;
            }
        }
    }
}

hipError_t launch_latency_head(const float* A, long long lda, const float* wfrag, const float* bias, float* Y, int ldy,
                               int M, int N, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(lat_head_kernel, dim3((N + 15) / 16, (M + lz::RP - 1) / lz::RP), dim3(256), 0, s, A, lda, wfrag, bias, Y,
                       ldy, M, N);
    return hipGetLastError();
}

// ---- rnn_gemv: the recurrence for one stream on a cluster of 4 workgroups --------------------------------------
//   thread (wave, lane): hidden unit row = wg*128 + wave*16 + (lane & 15); K quarter = lane >> 4 (128 k each).
//   W_hh is read once from the MFMA-fragment image: fragment (nb, kb) lane' = 16*lg' + l15 holds
//   W[nb*16 + l15][kb*16 + 4 lg' .. +3].
//   granules: hb[win][parity][512] of {tag = step + 1, value}; zeroed before every launch.
typedef unsigned long long u64;
__device__ unsigned g_spin_timeouts_latency;   // see tip_spin_timeouts()
__global__ __launch_bounds__(512) void rnn_gemv_kernel(const float* __restrict__ ih, const float* __restrict__ whh_frag,
                                                       float* __restrict__ hall, u64* __restrict__ hb,
                                                       unsigned* __restrict__ xcc_words, int B, int T, Guard gd) {
    using namespace lz;
    __shared__ __attribute__((aligned(16))) float hs2[2][4 * 132];   // 4 K-quarters of 128, padded: distinct banks per quarter; two buffers
                                                                     // alternate, so one barrier per step suffices
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    // the 4 members of a stream's cluster are taken 8 blocks apart: observed to share an XCD (verified below)
    const int wg = (blockIdx.x >> 3) & 3, win = (blockIdx.x & 7) + 8 * (blockIdx.x >> 5);
    if (win >= B) return;
    if ((gd.fault & 4) && win == 0 && wg == 1) return;   // TIP_OPT_FAULT_INJECT: this member never arrives
    const unsigned spin_lim = guard_spin_limit(gd.fault, 1u << 22);
    bool poisoned = false;   // a granule wait gave up: this thread's h words are NaN from here on and it never spins again
    const int row = wg * 128 + wave * 16 + l15;
    // run-time check that all 4 members really sit on one XCD: only then may granules be published with plain 8-byte
    // stores (they stay in the shared L2, where the peers' L1-bypassing loads see them after ~0.4 us); otherwise the
    // write-through agent-scope stores that are correct for any placement are used.
    __shared__ int s_same_xcd;
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xf;
        __hip_atomic_store(xcc_words + win * 4 + wg, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bool same = true;
        for (int m = 0; m < 4; ++m) {
            unsigned v = 0;
            for (unsigned spins = 0; spins < spin_lim; ++spins) {
                v = __hip_atomic_load(xcc_words + win * 4 + m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
    float4 w[32];
    {
        const float4* wf = reinterpret_cast<const float4*>(whh_frag);
        const int nb = row >> 4;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const int kb = lg * 8 + (j >> 2), lgp = j & 3;   // k = lg*128 + 4 j .. +3
            w[j] = wf[(size_t)(nb * 32 + kb) * 64 + lgp * 16 + l15];
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
                if (!poisoned && lg == 0)   // row t-1 is incomplete (the missing slice holds stale memory): no finite y row may come of it
                    hw[(size_t)(t - 1) * R + row] = __uint_as_float(kPoisonBits);
                poisoned = true;
                v = (u64)kPoisonBits;   // the word that never came is NaN: every hidden unit that reads it follows
            }
            float* hs = hs2[t & 1];
            hs[(tid >> 7) * 132 + (tid & 127)] = __uint_as_float((unsigned)v);
            __syncthreads();
            // the next step's input term is requested AFTER this step's polls (vector memory returns in order: issued in front
            // of them it would hold every poll back by its own round trip) and flies during the dot products
            if (t + 1 < T) ih_next = ihw[(size_t)(t + 1) * R + row];
            const float* hq = hs + lg * 132;
            // four chains (one per float4 component), written as two 2-wide FMAs on the ADJACENT halves of the operands: the
            // compiler's own pairing of the scalar form — (x, z) and (y, w) — cost three register moves per v_pk_fma_f32
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const float4 hv = *reinterpret_cast<const float4*>(hq + j * 4);
                a01 = __builtin_elementwise_fma((f2){w[j].x, w[j].y}, (f2){hv.x, hv.y}, a01);
                a23 = __builtin_elementwise_fma((f2){w[j].z, w[j].w}, (f2){hv.z, hv.w}, a23);
            }
            acc = (a01[0] + a01[1]) + (a23[0] + a23[1]);
            acc = lg4_sum(acc);   // the four K quarters (lane ^ 16, lane ^ 32) on permlane swaps: no LDS round trips on the serial chain
        }
        if (lg == 0) {
            const float hv = tip_tanh(acc + ihv);
            const u64 gran = ((u64)(unsigned)(t + 1) << 32) | (u64)__float_as_uint(hv);
            u64* gdst = hbw + (size_t)(t & 1) * R + row;
            if (same_xcd) *gdst = gran;   // one aligned 8-byte store: single-copy atomic, lands in the shared L2
            else __hip_atomic_store(gdst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            hw[(size_t)t * R + row] = hv;   // (after the granule: the partners wait for that one)
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
hipError_t read_spin_timeouts_latency(unsigned* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_spin_timeouts_latency), sizeof(unsigned));
}

bool latency_supported(const Dims& d, int B, int T) {
    return fused_supported(d, T) && fused_has_rnn_ih(d) && B >= 1 && B <= 64;
}

// workspace (floats): xa, xb [B][T][256]; qkv [B][T][768]; o [B][T][256]; hid [B][T][1024]; ih [B][T][512];
//                     stats [2][B][48][2]; granules [B][2][512] u64
size_t latency_workspace_floats(int B, int T) {
    const size_t bt = (size_t)B * T;
    return bt * (256 + 256 + 768 + 256 + 1024 + 512) + (size_t)2 * B * 48 * 2 + (size_t)B * 2 * 512 * 2 + (size_t)B * 4 + 1024;
}

hipError_t launch_latency_plan(const Dims& d, const float* fused_w, const float* whh_frag, const float* x_imu,
                               const float* x_s, const float* keep_mask, float keep_scale, float* ws, float* hall, int B,
                               int T, int num_cus, const Guard& gd, hipStream_t s, const TrainDropout* td) {
    using namespace lz;
    const size_t bt = (size_t)B * T;
    float* xa = ws;
    float* xb = xa + bt * 256;
    float* o = xb + bt * 256 + bt * 768;   // (the QKV slot of the workspace is unused since QKV + attention merged)
    float* hid = o + bt * 256;
    float* ihb = hid + bt * 1024;
    float* st0 = ihb + bt * 512;
    float* st1 = st0 + (size_t)B * 48 * 2;
    u64* gran = reinterpret_cast<u64*>(st1 + (size_t)B * 48 * 2);   // every term above is an even float count: 8-B aligned
    unsigned* xccw = reinterpret_cast<unsigned*>(gran + (size_t)B * 2 * R);   // 4 words per stream
    const int wbytes = (int)(fused_packed_floats(d) * 4);
    // offsets inside the fused section (tip_fused.hip)
    constexpr size_t IN_W = 0, IN_B = (size_t)D * KIN, LAYER0 = IN_B + D;
    constexpr size_t QKV_W = 0, QKV_B = QKV_W + (size_t)3 * D * D, WO_W = QKV_B + 3 * D, WO_B = WO_W + (size_t)D * D;
    constexpr size_t W1_W = WO_B + D, W1_B = W1_W + (size_t)F * D, W2_W = W1_B + F, W2_B = W2_W + (size_t)D * F;
    constexpr size_t G1 = W2_B + D, BE1 = G1 + D, G2 = BE1 + D, BE2 = G2 + D, LAYER_FLOATS = BE2 + D;
    hipLaunchKernelGGL(lat_in_kernel, dim3(16, B), dim3(256), 0, s, fused_w, wbytes, x_imu, x_s, keep_mask, keep_scale, xa, T,
                       d.n_imu_total, d.S, (int)(IN_W * 4), (int)IN_B, gran, xccw, td ? td->mkey : 0u, td && !keep_mask ? td->mthresh : 0u);
    const float* pg = nullptr;   // LayerNorm pending on the residual stream (norm2 of the previous layer)
    const float* pb = nullptr;
    for (int l = 0; l < d.L; ++l) {
        const size_t lo = LAYER0 + (size_t)l * LAYER_FLOATS;
        const float* LW = fused_w + lo;
        // xa = pre-norm input of the layer; st0 = its row statistics (when pg != null)
        LatDrop dr;
        if (td) { dr.seed = td->seed; dr.thresh = td->thresh; dr.scale = td->scale; }
        auto site = [&](unsigned k) { LatDrop x = dr; x.site = (unsigned)l * 4u + k; return x; };
        if (td && td->thresh) {   // tip_forward_dropout: the same stages with the training step's four dropout sites live
            hipLaunchKernelGGL(lat_qkv_attn_kernel<true>, dim3(16, B), dim3(256), 0, s, fused_w, wbytes, xa, pg, pb,
                               (int)((lo + QKV_W) * 4), (int)(lo + QKV_B), o, st0, T, site(0));
            hipLaunchKernelGGL((lat_res_gemm_kernel<4, 16, true>), dim3(16, B), dim3(256), 0, s, fused_w, wbytes, o, D, xa, st0, pg, pb,
                               (int)((lo + WO_W) * 4), (int)(lo + WO_B), xb, T, site(1));
            hipLaunchKernelGGL((lat_ln_gemm_kernel<true, true, true>), dim3(64, B), dim3(256), 0, s, fused_w, wbytes, xb, LW + G1,
                               LW + BE1, (int)((lo + W1_W) * 4), (int)(lo + W1_B), hid, F, st1, T, site(2));
            hipLaunchKernelGGL((lat_res_gemm_kernel<8, 64, true>), dim3(16, B), dim3(512), 0, s, fused_w, wbytes, hid, F, xb, st1, LW + G1,
                               LW + BE1, (int)((lo + W2_W) * 4), (int)(lo + W2_B), xa, T, site(3));
        } else {
            hipLaunchKernelGGL(lat_qkv_attn_kernel<false>, dim3(16, B), dim3(256), 0, s, fused_w, wbytes, xa, pg, pb,
                               (int)((lo + QKV_W) * 4), (int)(lo + QKV_B), o, st0, T, dr);
            hipLaunchKernelGGL((lat_res_gemm_kernel<4, 16>), dim3(16, B), dim3(256), 0, s, fused_w, wbytes, o, D, xa, st0, pg, pb,
                               (int)((lo + WO_W) * 4), (int)(lo + WO_B), xb, T, dr);
            // xb = pre-LN1 sum
            hipLaunchKernelGGL((lat_ln_gemm_kernel<true, true>), dim3(64, B), dim3(256), 0, s, fused_w, wbytes, xb, LW + G1,
                               LW + BE1, (int)((lo + W1_W) * 4), (int)(lo + W1_B), hid, F, st1, T, dr);
            hipLaunchKernelGGL((lat_res_gemm_kernel<8, 64>), dim3(16, B), dim3(512), 0, s, fused_w, wbytes, hid, F, xb, st1, LW + G1,
                               LW + BE1, (int)((lo + W2_W) * 4), (int)(lo + W2_B), xa, T, dr);
        }
        pg = LW + G2;
        pb = LW + BE2;
    }
    const size_t ih_off = LAYER0 + (size_t)d.L * LAYER_FLOATS;
    hipLaunchKernelGGL((lat_ln_gemm_kernel<false, false>), dim3(32, B), dim3(256), 0, s, fused_w, wbytes, xa, pg, pb, (int)(ih_off * 4),
                       (int)(ih_off + (size_t)R * D), ihb, R, (float*)nullptr, T, LatDrop{});
    {
        static PerDeviceInt occ_dev; int& occ = occ_dev.cur();   // the 4 workgroups of every stream's cluster must be resident together
        hipError_t ce = check_coresident(rnn_gemv_kernel, 512, (size_t)0, 4 * B, num_cus, &occ);
        if (ce != hipSuccess) return ce;
    }
    hipLaunchKernelGGL(rnn_gemv_kernel, dim3(32 * ((B + 7) / 8)), dim3(512), 0, s, ihb, whh_frag, hall, gran, xccw, B, T, gd);
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

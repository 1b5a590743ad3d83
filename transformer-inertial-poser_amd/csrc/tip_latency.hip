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
constexpr int D = 256, DH = 16, F = 1024, R = 512, RP = 48, RB = 3, TMAX = 40, KIN = 224;
constexpr int LDX = D + 4, LDU = KIN + 4;
}  // namespace lz

__device__ __forceinline__ float wsum64(float v) { return wave64_sum(v); }

__device__ __forceinline__ f32x4 ldfrag(__amdgpu_buffer_rsrc_t rsrc, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0));
}

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
                                                     unsigned* __restrict__ xcc_words) {
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
                const float xs_v = (x != x ? 0.f : x) * kv * (km ? keep_scale : 1.f); // :65, then (x * mask) * scale as before
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
template <bool RELU, bool BLOCKED_OUT>
__global__ __launch_bounds__(256) void lat_ln_gemm_kernel(const float* __restrict__ wts, int wbytes, const float* __restrict__ xpre,
                                                          const float* __restrict__ g, const float* __restrict__ be,
                                                          int w_off_b, int b_off, float* __restrict__ out, int ldo,
                                                          float* __restrict__ stats, int T) {
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
    if (wave < RB) {
        const int col = nb * 16 + l15;
        float* o = out + (size_t)win * T * ldo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = wave * 16 + lg * 4 + e;
            float v = s[e] + bv;
            if (RELU) v = fmaxf(v, 0.f);
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
__global__ __launch_bounds__(256) void lat_qkv_attn_kernel(const float* __restrict__ wts, int wbytes, const float* __restrict__ xpre,
                                                           const float* __restrict__ g, const float* __restrict__ be,
                                                           int w_off_b, int b_off, float* __restrict__ o_out,
                                                           float* __restrict__ stats, int T) {
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
    if (wave == 0) attention_head_mfma<DH + 4, RP + 4>(Qs, Ks, Vts, 0, lane);
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
template <int NW, int KBT>
__global__ __launch_bounds__(NW * 64) void lat_res_gemm_kernel(const float* __restrict__ wts, int wbytes, const float* __restrict__ A,
                                                               int lda, const float* __restrict__ xpre, const float* __restrict__ stats,
                                                               const float* __restrict__ g, const float* __restrict__ be, int w_off_b,
                                                               int b_off, float* __restrict__ out, int T) {
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
    if (wave < RB) {
        float* o = out + (size_t)win * T * D;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int row = wave * 16 + lg * 4 + e;
            float x = xres[e];
            if (g) x = (x - smean[e]) * srstd[e] * gc + bc;
            if (row < T) o[(size_t)row * D + ecol] = x + s[e] + bv;
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
                               int T, int num_cus, const Guard& gd, hipStream_t s) {
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
                       d.n_imu_total, d.S, (int)(IN_W * 4), (int)IN_B, gran, xccw);
    const float* pg = nullptr;   // LayerNorm pending on the residual stream (norm2 of the previous layer)
    const float* pb = nullptr;
    for (int l = 0; l < d.L; ++l) {
        const size_t lo = LAYER0 + (size_t)l * LAYER_FLOATS;
        const float* LW = fused_w + lo;
        // xa = pre-norm input of the layer; st0 = its row statistics (when pg != null)
        hipLaunchKernelGGL(lat_qkv_attn_kernel, dim3(16, B), dim3(256), 0, s, fused_w, wbytes, xa, pg, pb,
                           (int)((lo + QKV_W) * 4), (int)(lo + QKV_B), o, st0, T);
        hipLaunchKernelGGL((lat_res_gemm_kernel<4, 16>), dim3(16, B), dim3(256), 0, s, fused_w, wbytes, o, D, xa, st0, pg, pb,
                           (int)((lo + WO_W) * 4), (int)(lo + WO_B), xb, T);
        // xb = pre-LN1 sum
        hipLaunchKernelGGL((lat_ln_gemm_kernel<true, true>), dim3(64, B), dim3(256), 0, s, fused_w, wbytes, xb, LW + G1,
                           LW + BE1, (int)((lo + W1_W) * 4), (int)(lo + W1_B), hid, F, st1, T);
        hipLaunchKernelGGL((lat_res_gemm_kernel<8, 64>), dim3(16, B), dim3(512), 0, s, fused_w, wbytes, hid, F, xb, st1, LW + G1,
                           LW + BE1, (int)((lo + W2_W) * 4), (int)(lo + W2_B), xa, T);
        pg = LW + G2;
        pb = LW + BE2;
    }
    const size_t ih_off = LAYER0 + (size_t)d.L * LAYER_FLOATS;
    hipLaunchKernelGGL((lat_ln_gemm_kernel<false, false>), dim3(32, B), dim3(256), 0, s, fused_w, wbytes, xa, pg, pb, (int)(ih_off * 4),
                       (int)(ih_off + (size_t)R * D), ihb, R, (float*)nullptr, T);
    {
        static PerDeviceInt occ_dev; int& occ = occ_dev.cur();   // the 4 workgroups of every stream's cluster must be resident together
        hipError_t ce = check_coresident(rnn_gemv_kernel, 512, (size_t)0, 4 * B, num_cus, &occ);
        if (ce != hipSuccess) return ce;
    }
    hipLaunchKernelGGL(rnn_gemv_kernel, dim3(32 * ((B + 7) / 8)), dim3(512), 0, s, ihb, whh_frag, hall, gran, xccw, B, T, gd);
    return hipGetLastError();
}


// =====================================================================================================================
// "latency1": the chain above as ONE persistent kernel (B <= 8 streams).
//
// The chain is 4 launches per layer + 4: twenty kernels of 5-9 us each for well under 2 us of matrix work apiece — what they pay for is
// the launch ramp, the cold first fetch of their operands and the drain (profiles/r03/timeline_B1.txt: 170 us of a 178-us forward).
// Here `nw` persistent 512-thread workgroups walk the SAME stages, separated by a grid barrier instead of a kernel boundary:
//
//   * a workgroup is two independent 4-wave HALVES: every stage of the chain that runs on 256 threads (in_linear, QKV + attention of
//     one head, out-projection, FFN1, RNN input projection, output projection) is one work item of a half — same K split over 4
//     waves, same LDS reduction order, same instructions, hence bit-identical to the launch chain; FFN2 (K = 1024 split over 8
//     waves) is a work item of the whole workgroup.  Items go to half 0 of every worker first, so a stage with <= nw items has one
//     item per CU.
//   * grid barrier = one FLAG WORD per worker (value: launch epoch * 32 + stage), polled by the 64 lanes of wave 0 with one load —
//     no atomic read-modify-write on the critical path, no reset between launches (flags only grow; the epoch word is bumped by
//     worker 0 at its exit, after everyone has read it), so a captured HIP graph can replay the kernel.
//   * activations between stages cross CUs through the workspace with COHERENT accesses (stores sc1 = write-through at agent
//     scope, loads sc1 = not from L1): correct on any placement.  The launcher spreads `stride` = 8 block ids between workers so
//     that all of them land on ONE XCD (round-robin dispatch) and share an L2; the kernel verifies that from the XCC ids the workers
//     publish before the first barrier and only then downgrades its stores to plain ones (they stay in the shared L2: a one-way
//     hop of ~0.25 us instead of ~0.6, profiles/r03/hop_probe.txt).
//   * the recurrence (rnn_gemv above, 4 workers per stream) and the output projection run as the tail of the same kernel.
//   * every wait is bounded; a wait that gives up reports through the Guard, raises the sticky poison word of the sync area and the
//     output rows are written as NaN — never finite-but-wrong (include/tip_hip.h, tip_check).
// Sync area: 256 words at the end of the packed weight image (zeroed by the packers, written only by this kernel).
// =====================================================================================================================
namespace l1 {
constexpr int THREADS = 512;
constexpr int XS = lz::RP * lz::LDX;                 // 12480: staged rows / U
constexpr int RED = 4 * 3 * 256;                     // 3072: K-split partials of one half
constexpr int QK = lz::RP * (lz::DH + 4);            // 960
constexpr int VT = lz::DH * (lz::RP + 4);            // 832
constexpr int HALF = XS + RED + 2 * QK + VT;         // 18304 floats per half
constexpr int LDS_BYTES = 2 * HALF * 4;              // 146,432 B
constexpr int W_EPOCH = 0, W_POISON = 1, W_FLAGS = 16, W_XCC = 80;
constexpr int MAX_W = 64, MAX_B = 8;
constexpr int K_RNN = 30;                            // barrier index of "recurrence done" (stages use 1 .. 4 L + 2 <= 26)
static_assert(XS >= lz::RP * lz::LDU, "U fits the row plane");
static_assert(2 * HALF >= 8 * 3 * 256 && 2 * HALF >= 2 * 4 * 132, "FFN2 partials / RNN hidden vector alias the planes");
}  // namespace l1

struct Lat1Args {
    const float* wts; int wbytes;
    const float* x_imu; const float* x_s; const float* keep_mask; float keep_scale;
    float* ws; int ws_bytes;                                  // the forward's whole workspace, one buffer descriptor
    int xa, xb, o, hid, ihb, st0, st1, gran, hall;            // float offsets into ws
    const float* whh_frag; const float* out_frag; const float* out_bias;
    float* y; int ldy, N;
    unsigned* sync;
    int B, T, NI, S, L, last_only, nw, stride;
    Guard gd;
};

typedef unsigned u32x4_l1 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_l1 __attribute__((ext_vector_type(2)));

// coherent accesses to the workspace (byte offsets from its base)
struct Coh {
    __amdgpu_buffer_rsrc_t rs;
    bool plain;   // all workers share an XCD: stores may stay in its L2
    __device__ __forceinline__ float ld1(int off_b) const {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off_b, 0, 16));
    }
    __device__ __forceinline__ f32x4 ld4(int off_b) const {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off_b, 0, 16));
    }
    __device__ __forceinline__ void st1(int off_b, float v) const {
        if (plain) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, off_b, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, off_b, 0, 16);
    }
    __device__ __forceinline__ void st4(int off_b, f32x4 v) const {
        if (plain) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_l1, v), rs, off_b, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_l1, v), rs, off_b, 0, 16);
    }
    __device__ __forceinline__ void st2u(int off_b, unsigned lo, unsigned hi) const {
        const u32x2_l1 v = {lo, hi};
        if (plain) __builtin_amdgcn_raw_buffer_store_b64(v, rs, off_b, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b64(v, rs, off_b, 0, 16);
    }
};

// stage_rows_ln<256> with the source rows read coherently; `wv` = wave inside the half (0..3), `ln` = lane
__device__ __forceinline__ void l1_stage_rows_ln(float* Xs, const Coh& c, int src_b, int T, const float* __restrict__ g,
                                                 const float* __restrict__ be, bool stats_out, int stats_b, int wv, int ln) {
    using namespace lz;
    constexpr int NW = 4, NPASS = (RP + 4 * NW - 1) / (4 * NW);
    const int q = ln & 15, sub = ln >> 4;
    f32x4 v[NPASS][4];
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int row = (p * NW + wv) * 4 + sub;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[p][j] = row < T ? c.ld4(src_b + (row * D + (q + 16 * j) * 4) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
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
            const int row = (p * NW + wv) * 4 + sub;
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
                if (row >= T) v[p][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (stats_out && q == 0 && row < T) { c.st1(stats_b + row * 8, mean); c.st1(stats_b + row * 8 + 4, rstd); }
        }
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int row = (p * NW + wv) * 4 + sub;
        if (row < RP)
#pragma unroll
            for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(Xs + row * LDX + (q + 16 * j) * 4) = v[p][j];
    }
}

// reduce_partials<4> of ONE half: the caller brackets it with the workgroup barriers (both halves run in lockstep)
__device__ __forceinline__ void l1_put_partials(float* red, const f32x4 (&acc)[lz::RB], int wv, int ln) {
#pragma unroll
    for (int r = 0; r < lz::RB; ++r) *reinterpret_cast<f32x4*>(red + ((wv * 3 + r) * 64 + ln) * 4) = acc[r];
}
template <int NW>
__device__ __forceinline__ f32x4 l1_get_partials(const float* red, int wv, int ln) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (wv < lz::RB) {
#pragma unroll
        for (int w = 0; w < NW; ++w) s += *reinterpret_cast<const f32x4*>(red + ((w * 3 + wv) * 64 + ln) * 4);
    }
    return s;
}

__device__ unsigned long long g_lat1_trace[128];   // TIP_LAT1_TRACE=1: s_memtime stamps of worker 0 (stage done / barrier passed)
template <bool TRACE>
__global__ __launch_bounds__(l1::THREADS) void lat1_kernel(Lat1Args a) {
    using namespace lz;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ int s_same;
    if ((int)blockIdx.x % a.stride) return;
    const int wid = (int)blockIdx.x / a.stride;
    if (wid >= a.nw) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hw = wave >> 2, wv = wave & 3, th = tid & 255;
    const int l15 = lane & 15, lg = lane >> 4;
    const int nw = a.nw, T = a.T, B = a.B;
    float* Hb = smem + hw * l1::HALF;          // this half's planes
    float* Xs = Hb;
    float* red = Hb + l1::XS;
    float* Qs = red + l1::RED;
    float* Ks = Qs + l1::QK;
    float* Vts = Ks + l1::QK;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wts), 0, a.wbytes, 0x00020000);
    Coh c;
    c.rs = __builtin_amdgcn_make_buffer_rsrc(a.ws, 0, a.ws_bytes, 0x00020000);
    c.plain = false;
    unsigned* sync = a.sync;
    const unsigned epoch = __hip_atomic_load(sync + l1::W_EPOCH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned tagbase = epoch * 32u;
    const unsigned spin_lim = guard_spin_limit(a.gd.fault, 1u << 22);
    // (grid barriers outlast a recurrence whose members each sit out one shortened granule wait)
    const unsigned bar_lim = (a.gd.fault & 7) ? (1u << 19) : (1u << 22);
    if (tid == 0) {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        __hip_atomic_store(sync + l1::W_XCC + wid, (xcc & 0xf) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_same = 0;
    }
    int barrier_no = 0;
    bool gave_up = false;     // (wave 0 only) a barrier wait of this worker timed out: never spin again
    // grid barrier.  wait = false: announce arrival only (workers with nothing left to consume)
    int tr_n = 0;
    auto stamp = [&]() {
        if (TRACE && wid == 0 && tid == 0 && tr_n < 127) g_lat1_trace[1 + tr_n++] = __builtin_amdgcn_s_memtime();
    };
    stamp();
    // `pre`: the next stage's first weight fetch, issued right behind the barrier.  (Measured, profiles/r04/lat1_trace*.txt: issuing it
    // BEFORE the wait — so that it would fly under the barrier — made every barrier 1.5 us longer, because the poll loads return in
    // order behind it, and polling through the scalar path instead did not win that back; the stage bodies did not get shorter.)
    auto grid_barrier = [&](int k, bool wait, auto&& pre) {
        stamp();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's coherent stores have been acknowledged ...
        __syncthreads();                                   // ... and everybody else's, before the one flag store that publishes them
        if (wave == 0) {
            const unsigned target = tagbase + (unsigned)k;
            if (lane == 0) {
                if (c.plain) *const_cast<volatile unsigned*>(sync + l1::W_FLAGS + wid) = target;
                else __hip_atomic_store(sync + l1::W_FLAGS + wid, target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (wait) {
                const unsigned lim = gave_up ? 1u : bar_lim;
                bool ok = false;
                for (unsigned spins = 0; spins < lim; ++spins) {
                    const unsigned v = lane < nw ? __hip_atomic_load(sync + l1::W_FLAGS + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
                    ok = __builtin_amdgcn_ballot_w64((int)(v - target) < 0) == 0ull;
                    if (ok) break;
                }
                if (!ok) {
                    if (!gave_up && lane == 0) {
                        atomicAdd(&g_spin_timeouts_latency, 1u);
                        guard_report(a.gd.err);
                        __hip_atomic_store(sync + l1::W_POISON, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    gave_up = true;
                }
                if (k == 1) {
                    // placement: do all workers share an XCD?  (their ids were published before they arrived here)
                    const unsigned x = lane < nw ? __hip_atomic_load(sync + l1::W_XCC + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    const unsigned x0 = __builtin_amdgcn_readfirstlane(x);
                    const bool same = ok && __builtin_amdgcn_ballot_w64(lane < nw && x != x0) == 0ull;
                    if (lane == 0) s_same = same ? 1 : 0;
                }
            }
        }
        __syncthreads();
        if (k == 1) c.plain = s_same != 0;
        barrier_no = k;
        stamp();
        pre();
    };
    auto no_pre = [] {};

    // offsets inside the fused section (tip_fused.hip)
    constexpr size_t IN_W = 0, IN_B = (size_t)D * KIN, LAYER0 = IN_B + D;
    constexpr size_t QKV_W = 0, QKV_B = QKV_W + (size_t)3 * D * D, WO_W = QKV_B + 3 * D, WO_B = WO_W + (size_t)D * D;
    constexpr size_t W1_W = WO_B + D, W1_B = W1_W + (size_t)F * D, W2_W = W1_B + F, W2_B = W2_W + (size_t)D * F;
    constexpr size_t G1 = W2_B + D, BE1 = G1 + D, G2 = BE1 + D, BE2 = G2 + D, LAYER_FLOATS = BE2 + D;

    // ------------------------------------------------------------------------------------------------------------------
    // stage: prologue + in_linear (lat_in_kernel).  item = win * 16 + nb
    // ------------------------------------------------------------------------------------------------------------------
    auto stage_in = [&](int item, bool act) {
        const int nb = item & 15, win = item >> 4;
        float* U = Xs;
        constexpr int KB = KIN / 16;
        const int kb0 = wv * 4;
        const int nkb = KB - kb0 < 4 ? KB - kb0 : 4;
        f32x4 w[4];
        f32x4 acc[RB] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        float bv_in = 0.f;
        if (act) {
            load_kslice<4>(w, rsrc, lane * 16, (int)(IN_W * 4) + (nb * KB + kb0) * 1024, nkb);
            bv_in = a.wts[IN_B + nb * 16 + l15];
            if (nb == 0) {   // this stream's RNN hand-off granules: tags start at 0 every launch
                for (int i = th; i < 2 * R; i += 256) c.st2u((a.gran + win * 4 * R) * 4 + i * 8, 0u, 0u);
            }
            const float* xi = a.x_imu + (size_t)win * T * a.NI;
            const float* xs = a.x_s + (size_t)win * T * a.S;
            const float* km = a.keep_mask ? a.keep_mask + (size_t)win * T * a.S : nullptr;
            const int NI = a.NI, S = a.S;
            constexpr int RPW = RP / 4, NCH = (KIN + 4 + 63) / 64;
            float v[RPW][NCH];
#pragma unroll
            for (int i = 0; i < RPW; ++i) {
                const int row = wv + i * 4;
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    const int cc = ch * 64 + lane;
                    const int rc = row < T ? row : T - 1;
                    const bool imu = cc < NI;
                    const int cs = cc - NI < 0 ? 0 : (cc - NI < S ? cc - NI : S - 1);
                    const float* pa = imu ? xi + (size_t)rc * NI + cc : xs + (size_t)rc * S + cs;
                    const float x = *pa;
                    float kv = 1.f;
                    if (km) kv = *(imu ? pa : km + (size_t)rc * S + cs);
                    const float xs_v = (x != x ? 0.f : x) * kv * (km ? a.keep_scale : 1.f);
                    v[i][ch] = (row < T && cc < NI + S) ? (imu ? x : xs_v) : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < RPW; ++i)
#pragma unroll
                for (int ch = 0; ch < NCH; ++ch) {
                    const int cc = ch * 64 + lane;
                    if (cc < LDU) U[(wv + i * 4) * LDU + cc] = v[i][ch];
                }
        }
        __syncthreads();
        if (act) {
            mma_kslice<4>(acc, U + l15 * LDU + lg * 4 + kb0 * 16, LDU, w, nkb);
            l1_put_partials(red, acc, wv, lane);
        }
        __syncthreads();
        if (act) {
            const f32x4 s = l1_get_partials<4>(red, wv, lane);
            if (wv < RB) {
                const int col = nb * 16 + l15;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = wv * 16 + lg * 4 + e;
                    if (row < T) c.st1((a.xa + (win * T + row) * D + col) * 4, s[e] + bv_in);
                }
            }
        }
        __syncthreads();
    };

    // ------------------------------------------------------------------------------------------------------------------
    // stage: [pending LN] + Q/K/V projection of one head + its causal attention (lat_qkv_attn_kernel).  item = win * 16 + head
    // ------------------------------------------------------------------------------------------------------------------
    // weights (and biases) of a stage's FIRST item are requested BEFORE the grid barrier in front of the stage: they do not depend on
    // the activations, and with all workers on one XCD (4 MB of L2 for 13 MB of weights) they come from the Infinity Cache / HBM
    f32x4 pw[12];
    float pbias[4];
    auto pre_qkv = [&](int item, bool act, int w_off_b, int b_off) {
        const int head = item & 15, kb0 = wv * 4;
        if (act) {
#pragma unroll
            for (int which = 0; which < 3; ++which)
                load_kslice<4>(reinterpret_cast<f32x4(&)[4]>(pw[which * 4]), rsrc, lane * 16, w_off_b + ((which * 16 + head) * 16 + kb0) * 1024, 4);
#pragma unroll
            for (int which = 0; which < 3; ++which) pbias[which] = a.wts[b_off + (which * 16 + head) * 16 + l15];
        }
    };
    auto stage_qkv = [&](int item, bool act, bool pre, const float* g, const float* be, int w_off_b, int b_off) {
        const int head = item & 15, win = item >> 4;
        const int kb0 = wv * 4;
        if (!pre) pre_qkv(item, act, w_off_b, b_off);
        f32x4 (&w)[3][4] = reinterpret_cast<f32x4(&)[3][4]>(pw[0]);
        float (&bqkv)[4] = pbias;
        if (act) {
            l1_stage_rows_ln(Xs, c, (a.xa + win * T * D) * 4, T, g, be, g != nullptr && head == 0, (a.st0 + win * RP * 2) * 4, wv, lane);
        }
        __syncthreads();
#pragma unroll
        for (int which = 0; which < 3; ++which) {
            if (act) {
                f32x4 acc[RB] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                mma_kslice<4>(acc, Xs + l15 * LDX + lg * 4 + kb0 * 16, LDX, w[which], 4);
                l1_put_partials(red, acc, wv, lane);
            }
            __syncthreads();
            if (act) {
                const f32x4 sres = l1_get_partials<4>(red, wv, lane);
                if (wv < RB) {
                    const float bv = bqkv[which];
                    if (which < 2) {
                        float* dst = which == 0 ? Qs : Ks;
#pragma unroll
                        for (int e = 0; e < 4; ++e) dst[(wv * 16 + lg * 4 + e) * (DH + 4) + l15] = sres[e] + bv;
                    } else {
                        *reinterpret_cast<f32x4*>(Vts + l15 * (RP + 4) + wv * 16 + lg * 4) = sres + bv;
                    }
                }
            }
            __syncthreads();
        }
        if (act && wv == 0) attention_head_mfma<DH + 4, RP + 4>(Qs, Ks, Vts, 0, lane);
        __syncthreads();
        if (act) {
            const int row = th >> 2, c0 = (th & 3) * 4;
            if (row < T) c.st4((a.o + win * T * D + (head * T + row) * 16 + c0) * 4, *reinterpret_cast<const f32x4*>(Qs + row * (DH + 4) + c0));
        }
        __syncthreads();
    };

    // ------------------------------------------------------------------------------------------------------------------
    // stage: LN (optional) -> one 16-column block of X W^T (lat_ln_gemm_kernel): FFN1 (+ReLU, blocked output), RNN-ih.
    //   item = win * nblk + nb
    // ------------------------------------------------------------------------------------------------------------------
    auto pre_ln_gemm = [&](int item, bool act, int nblk, int w_off_b, int b_off) {
        const int nb = item % nblk, kb0 = wv * 4;
        if (act) {
            load_kslice<4>(reinterpret_cast<f32x4(&)[4]>(pw[0]), rsrc, lane * 16, w_off_b + (nb * 16 + kb0) * 1024, 4);
            pbias[0] = a.wts[b_off + nb * 16 + l15];
        }
    };
    auto stage_ln_gemm = [&](int item, bool act, bool pre, int nblk, bool relu, bool blocked, int src, const float* g, const float* be, int w_off_b,
                             int b_off, int out, int ldo, bool stats, int stats_off) {
        const int nb = item % nblk, win = item / nblk;
        const int kb0 = wv * 4;
        if (!pre) pre_ln_gemm(item, act, nblk, w_off_b, b_off);
        f32x4 (&w)[4] = reinterpret_cast<f32x4(&)[4]>(pw[0]);
        const float bv = pbias[0];
        if (act) {
            l1_stage_rows_ln(Xs, c, (src + win * T * D) * 4, T, g, be, stats && nb == 0, (stats_off + win * RP * 2) * 4, wv, lane);
        }
        __syncthreads();
        if (act) {
            f32x4 acc[RB] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            mma_kslice<4>(acc, Xs + l15 * LDX + lg * 4 + kb0 * 16, LDX, w, 4);
            l1_put_partials(red, acc, wv, lane);
        }
        __syncthreads();
        if (act) {
            const f32x4 s = l1_get_partials<4>(red, wv, lane);
            if (wv < RB) {
                const int col = nb * 16 + l15;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = wv * 16 + lg * 4 + e;
                    float v = s[e] + bv;
                    if (relu) v = fmaxf(v, 0.f);
                    if (row < T) {
                        if (blocked) c.st1((out + win * T * ldo + (nb * T + row) * 16 + l15) * 4, v);
                        else c.st1((out + win * T * ldo + row * ldo + col) * 4, v);
                    }
                }
            }
        }
        __syncthreads();
    };

    // ------------------------------------------------------------------------------------------------------------------
    // stage: residual GEMM (lat_res_gemm_kernel<NW, KBT>): out = resid(xpre) + A W^T + b.  NW = 4: item of a half; NW = 8: of the
    // workgroup.  item = win * 16 + nb
    // ------------------------------------------------------------------------------------------------------------------
    auto pre_res4 = [&](int item, bool act, const float* g, const float* be, int w_off_b, int b_off) {
        const int nb = item & 15, kb0 = wv * 4, ecol = nb * 16 + l15;
        if (act) {
#pragma unroll
            for (int k = 0; k < 4; ++k) pw[k] = ldfrag(rsrc, lane * 16, w_off_b + (nb * 16 + kb0 + k) * 1024);
            pbias[0] = a.wts[b_off + ecol];
            pbias[1] = g ? g[ecol] : 1.f;
            pbias[2] = g ? be[ecol] : 0.f;
        }
    };
    auto stage_res4 = [&](int item, bool act, bool pre, int A, int xpre, int stats, const float* g, const float* be, int w_off_b, int b_off, int out) {
        constexpr int KBW = 4;
        const int nb = item & 15, win = item >> 4;
        const int kb0 = wv * KBW;
        f32x4 acc[RB] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const int ecol = nb * 16 + l15;
        const int ewave = wv < RB ? wv : 0;
        if (!pre) pre_res4(item, act, g, be, w_off_b, b_off);
        f32x4 (&w)[4] = reinterpret_cast<f32x4(&)[4]>(pw[0]);
        const float bv = pbias[0], gc = pbias[1], bc = pbias[2];
        float xres[4] = {0.f, 0.f, 0.f, 0.f}, smean[4] = {0.f, 0.f, 0.f, 0.f}, srstd[4] = {1.f, 1.f, 1.f, 1.f};
        if (act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = ewave * 16 + lg * 4 + e, rc = row < T ? row : T - 1;
                xres[e] = c.ld1((xpre + win * T * D + rc * D + ecol) * 4);
                smean[e] = g ? c.ld1((stats + win * RP * 2 + rc * 2) * 4) : 0.f;
                srstd[e] = g ? c.ld1((stats + win * RP * 2 + rc * 2 + 1) * 4) : 1.f;
            }
            const int Ab = (A + win * T * D + lg * 4) * 4;   // K = 256: a window's blocked image has T * 256 floats
            f32x4 av[RB][KBW];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int row = r * 16 + l15;
#pragma unroll
                for (int k = 0; k < KBW; ++k)
                    av[r][k] = row < T ? c.ld4(Ab + (((kb0 + k) * T + row) * 16) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int k = 0; k < KBW; ++k) {
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][k].x, w[k].x, acc[r], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][k].y, w[k].y, acc[r], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][k].z, w[k].z, acc[r], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][k].w, w[k].w, acc[r], 0, 0, 0);
            }
            l1_put_partials(red, acc, wv, lane);
        }
        __syncthreads();
        if (act) {
            const f32x4 s = l1_get_partials<4>(red, wv, lane);
            if (wv < RB) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int row = wv * 16 + lg * 4 + e;
                    float x = xres[e];
                    if (g) x = (x - smean[e]) * srstd[e] * gc + bc;
                    if (row < T) c.st1((out + win * T * D + row * D + ecol) * 4, x + s[e] + bv);
                }
            }
        }
        __syncthreads();
    };
    auto pre_res8 = [&](int item, bool act, const float* g, const float* be, int w_off_b, int b_off) {
        const int nb = item & 15, kb0 = wave * 8, ecol = nb * 16 + l15;
        if (act) {
#pragma unroll
            for (int k = 0; k < 8; ++k) pw[k] = ldfrag(rsrc, lane * 16, w_off_b + (nb * 64 + kb0 + k) * 1024);
            pbias[0] = a.wts[b_off + ecol];
            pbias[1] = g[ecol];
            pbias[2] = be[ecol];
        }
    };
    auto stage_res8 = [&](int item, bool act, bool pre, int A, int xpre, int stats, const float* g, const float* be, int w_off_b, int b_off, int out) {
        constexpr int KBW = 8, NW = 8;
        float* red8 = smem;   // [8][3][64][4]
        const int nb = item & 15, win = item >> 4;
        const int kb0 = wave * KBW;
        f32x4 acc[RB] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const int ecol = nb * 16 + l15;
        const int ewave = wave < RB ? wave : 0;
        if (!pre) pre_res8(item, act, g, be, w_off_b, b_off);
        f32x4 (&w)[8] = reinterpret_cast<f32x4(&)[8]>(pw[0]);
        const float bv = pbias[0], gc = pbias[1], bc = pbias[2];
        float xres[4] = {0.f, 0.f, 0.f, 0.f}, smean[4] = {0.f, 0.f, 0.f, 0.f}, srstd[4] = {1.f, 1.f, 1.f, 1.f};
        if (act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = ewave * 16 + lg * 4 + e, rc = row < T ? row : T - 1;
                xres[e] = c.ld1((xpre + win * T * D + rc * D + ecol) * 4);
                smean[e] = c.ld1((stats + win * RP * 2 + rc * 2) * 4);
                srstd[e] = c.ld1((stats + win * RP * 2 + rc * 2 + 1) * 4);
            }
            const int Ab = (A + win * T * F + lg * 4) * 4;   // K = 1024
            f32x4 av[RB][KBW];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int row = r * 16 + l15;
#pragma unroll
                for (int k = 0; k < KBW; ++k)
                    av[r][k] = row < T ? c.ld4(Ab + (((kb0 + k) * T + row) * 16) * 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int k = 0; k < KBW; ++k) {
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][k].x, w[k].x, acc[r], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][k].y, w[k].y, acc[r], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][k].z, w[k].z, acc[r], 0, 0, 0);
#pragma unroll
                for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][k].w, w[k].w, acc[r], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < RB; ++r) *reinterpret_cast<f32x4*>(red8 + ((wave * 3 + r) * 64 + lane) * 4) = acc[r];
        }
        __syncthreads();
        if (act && wave < RB) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w8 = 0; w8 < NW; ++w8) s += *reinterpret_cast<const f32x4*>(red8 + ((w8 * 3 + wave) * 64 + lane) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = wave * 16 + lg * 4 + e;
                const float x = (xres[e] - smean[e]) * srstd[e] * gc + bc;
                if (row < T) c.st1((out + win * T * D + row * D + ecol) * 4, x + s[e] + bv);
            }
        }
        __syncthreads();
    };

    // half-granular stages: items go to half 0 of every worker first
    auto half_item = [&](int base) { return base + hw * nw + wid; };

    // ================================================================================================================
    int k = 0;
    const int n16 = 16 * B, n64 = 64 * B, n32 = 32 * B;
    auto first_half = [&](int n) { const int it = half_item(0); return it < n ? it : 0; };
    for (int base = 0; base < n16; base += 2 * nw) {
        const int it = half_item(base);
        stage_in(it < n16 ? it : 0, it < n16);
    }
    const float* pg = nullptr;
    const float* pb = nullptr;
    grid_barrier(++k, true, [&] { pre_qkv(first_half(n16), half_item(0) < n16, (int)((LAYER0 + QKV_W) * 4), (int)(LAYER0 + QKV_B)); });
#pragma unroll 1
    for (int l = 0; l < a.L; ++l) {
        const size_t lo = LAYER0 + (size_t)l * LAYER_FLOATS;
        const float* LW = a.wts + lo;
        for (int base = 0; base < n16; base += 2 * nw) {
            const int it = half_item(base);
            stage_qkv(it < n16 ? it : 0, it < n16, base == 0, pg, pb, (int)((lo + QKV_W) * 4), (int)(lo + QKV_B));
        }
        grid_barrier(++k, true, [&] { pre_res4(first_half(n16), half_item(0) < n16, pg, pb, (int)((lo + WO_W) * 4), (int)(lo + WO_B)); });
        for (int base = 0; base < n16; base += 2 * nw) {
            const int it = half_item(base);
            stage_res4(it < n16 ? it : 0, it < n16, base == 0, a.o, a.xa, a.st0, pg, pb, (int)((lo + WO_W) * 4), (int)(lo + WO_B), a.xb);
        }
        grid_barrier(++k, true, [&] { pre_ln_gemm(first_half(n64), half_item(0) < n64, 64, (int)((lo + W1_W) * 4), (int)(lo + W1_B)); });
        for (int base = 0; base < n64; base += 2 * nw) {
            const int it = half_item(base);
            stage_ln_gemm(it < n64 ? it : 0, it < n64, base == 0, 64, true, true, a.xb, LW + G1, LW + BE1, (int)((lo + W1_W) * 4), (int)(lo + W1_B),
                          a.hid, F, true, a.st1);
        }
        grid_barrier(++k, true, [&] { pre_res8(wid < n16 ? wid : 0, wid < n16, LW + G1, LW + BE1, (int)((lo + W2_W) * 4), (int)(lo + W2_B)); });
        for (int base = 0; base < n16; base += nw) {
            const int it = base + wid;
            stage_res8(it < n16 ? it : 0, it < n16, base == 0, a.hid, a.xb, a.st1, LW + G1, LW + BE1, (int)((lo + W2_W) * 4), (int)(lo + W2_B), a.xa);
        }
        pg = LW + G2;
        pb = LW + BE2;
        grid_barrier(++k, true, [&] {
            if (l + 1 < a.L) {
                pre_qkv(first_half(n16), half_item(0) < n16, (int)((lo + LAYER_FLOATS + QKV_W) * 4), (int)(lo + LAYER_FLOATS + QKV_B));
            } else {
                const size_t ih_off = LAYER0 + (size_t)a.L * LAYER_FLOATS;
                pre_ln_gemm(first_half(n32), half_item(0) < n32, 32, (int)(ih_off * 4), (int)(ih_off + (size_t)R * D));
            }
        });
    }
    {
        const size_t ih_off = LAYER0 + (size_t)a.L * LAYER_FLOATS;
        for (int base = 0; base < n32; base += 2 * nw) {
            const int it = half_item(base);
            stage_ln_gemm(it < n32 ? it : 0, it < n32, base == 0 && a.L > 0, 32, false, false, a.xa, pg, pb, (int)(ih_off * 4), (int)(ih_off + (size_t)R * D),
                          a.ihb, R, false, 0);
        }
    }
    // who runs the recurrence: 4 workers per stream that share an XCD (stride 8: every worker does; stride 1: ids 8 apart)
    const bool spread = a.stride == 1;
    int rwin = -1, rwg = 0;
    if (spread) { if (wid < 32 && (wid & 7) < B) { rwin = wid & 7; rwg = wid >> 3; } }
    else if (wid < 4 * B) { rwin = wid >> 2; rwg = wid & 3; }
    // output projection items: (column block, 48-row tile) of M = B (last rows) or B * T rows
    const int hM = a.last_only ? B : B * T;
    const int nbo = (a.N + 15) / 16, nhead = nbo * ((hM + RP - 1) / RP);
    const bool has_head = wid < nhead;   // half 0 of worker `wid` owns item `wid` (+ multiples of 2 nw)
    grid_barrier(++k, rwin >= 0, no_pre);
    // ---- recurrence (rnn_gemv_kernel) ---------------------------------------------------------------------------------
    if (rwin >= 0 && !((a.gd.fault & 4) && rwin == 0 && rwg == 1)) {
        const int win = rwin, wg = rwg;
        float* hs2 = smem;    // [2][4 * 132]
        bool poisoned = false;
        const int row = wg * 128 + wave * 16 + l15;
        // (placement was checked at the first barrier: with all workers on one XCD the granules may be plain 8-byte stores)
        const bool same_xcd = c.plain;
        float4 w[32];
        {
            const float4* wf = reinterpret_cast<const float4*>(a.whh_frag);
            const int nb = row >> 4;
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const int kb = lg * 8 + (j >> 2), lgp = j & 3;
                w[j] = wf[(size_t)(nb * 32 + kb) * 64 + lgp * 16 + l15];
            }
        }
        u64* hbw = reinterpret_cast<u64*>(a.ws + a.gran) + (size_t)win * 2 * R;
        const int ihw_b = (a.ihb + win * T * R) * 4;
        const int hw_b = (a.hall + win * T * R) * 4;
        float ih_next = c.ld1(ihw_b + row * 4);
        for (int t = 0; t < T; ++t) {
            const float ihv = ih_next;
            float acc = 0.f;
            if (t == 0 && T > 1) ih_next = c.ld1(ihw_b + (R + row) * 4);
            if (t > 0) {
                const u64* gp = hbw + (size_t)((t - 1) & 1) * R + tid;
                u64 v = 0;
                const unsigned lim = poisoned ? 1u : spin_lim;
                for (unsigned spins = 0; spins < lim; ++spins) {
                    v = __hip_atomic_load(gp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(v >> 32) == (unsigned)t) break;
                }
                if ((unsigned)(v >> 32) != (unsigned)t) {
                    if (!poisoned) {   // (the missing slice poisons this stream's rows below: no global poison word for it)
                        atomicAdd(&g_spin_timeouts_latency, 1u);
                        guard_report(a.gd.err);
                    }
                    if (!poisoned && lg == 0) c.st1(hw_b + ((t - 1) * R + row) * 4, __uint_as_float(kPoisonBits));
                    poisoned = true;
                    v = (u64)kPoisonBits;
                }
                float* hs = hs2 + (t & 1) * (4 * 132);
                hs[(tid >> 7) * 132 + (tid & 127)] = __uint_as_float((unsigned)v);
                __syncthreads();
                if (t + 1 < T) ih_next = c.ld1(ihw_b + ((t + 1) * R + row) * 4);
                const float* hq = hs + lg * 132;
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 32; ++j) {
                    const float4 hv = *reinterpret_cast<const float4*>(hq + j * 4);
                    a01 = __builtin_elementwise_fma((f2){w[j].x, w[j].y}, (f2){hv.x, hv.y}, a01);
                    a23 = __builtin_elementwise_fma((f2){w[j].z, w[j].w}, (f2){hv.z, hv.w}, a23);
                }
                acc = (a01[0] + a01[1]) + (a23[0] + a23[1]);
                acc = lg4_sum(acc);
            }
            if (lg == 0) {
                const float hv = tip_tanh(acc + ihv);
                const u64 gran = ((u64)(unsigned)(t + 1) << 32) | (u64)__float_as_uint(hv);
                u64* gdst = hbw + (size_t)(t & 1) * R + row;
                if (same_xcd) *gdst = gran;
                else __hip_atomic_store(gdst, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                c.st1(hw_b + (t * R + row) * 4, hv);
            }
        }
    }
    grid_barrier(l1::K_RNN, has_head, no_pre);
    // ---- output projection (lat_head_kernel): item = mtile * nbo + nb ----------------------------------------------------
    if (has_head) {
        constexpr int KBT = R / 16, KBW = KBT / 4;
        const bool bad = __hip_atomic_load(sync + l1::W_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u;
        const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.out_frag), 0, nbo * R * 64, 0x00020000);
        const long long lda = a.last_only ? (long long)T * R : R;
        const int a0 = a.hall + (a.last_only ? (T - 1) * R : 0);
        for (int base = 0; base < nhead; base += 2 * nw) {
            const int it = half_item(base);
            const bool act = it < nhead;
            const int nb = act ? it % nbo : 0, m0 = (act ? it / nbo : 0) * RP;
            const int kb0 = wv * KBW;
            f32x4 acc[RB] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (act) {
                f32x4 w[KBW];
#pragma unroll
                for (int kk = 0; kk < KBW; ++kk) w[kk] = ldfrag(rsw, lane * 16, (nb * KBT + kb0 + kk) * 1024);
                f32x4 av[RB][KBW];
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const int row = m0 + r * 16 + l15;
                    const int ap = (int)((a0 + (long long)(row < hM ? row : 0) * lda + kb0 * 16 + lg * 4) * 4);
#pragma unroll
                    for (int kk = 0; kk < KBW; ++kk) av[r][kk] = row < hM ? c.ld4(ap + kk * 64) : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int kk = 0; kk < KBW; ++kk) {
#pragma unroll
                    for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][kk].x, w[kk].x, acc[r], 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][kk].y, w[kk].y, acc[r], 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][kk].z, w[kk].z, acc[r], 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < RB; ++r) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[r][kk].w, w[kk].w, acc[r], 0, 0, 0);
                }
                l1_put_partials(red, acc, wv, lane);
            }
            __syncthreads();
            if (act) {
                const f32x4 sres = l1_get_partials<4>(red, wv, lane);
                if (wv < RB) {
                    const int col = nb * 16 + l15;
                    if (col < a.N) {
                        const float bv = a.out_bias[col];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int row = m0 + wv * 16 + lg * 4 + e;
                            if (row < hM) a.y[(size_t)row * a.ldy + col] = bad ? __uint_as_float(kPoisonBits) : sres[e] + bv;
                        }
                    }
                }
            }
            __syncthreads();
        }
    }
    stamp();
    if (TRACE && wid == 0 && tid == 0) g_lat1_trace[0] = (unsigned long long)tr_n;
    // the next launch's epoch (every worker read this launch's before the first barrier)
    if (wid == 0 && tid == 0) __hip_atomic_store(sync + l1::W_EPOCH, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

bool latency1_supported(const Dims& d, int B, int T) {
    return latency_supported(d, B, T) && B <= l1::MAX_B && d.L <= 6 && d.S <= 16 * l1::MAX_W;
}

hipError_t launch_latency1_plan(const Dims& d, const float* fused_w, const float* whh_frag, const float* out_frag, const float* out_bias,
                                const float* x_imu, const float* x_s, const float* keep_mask, float keep_scale, float* ws_base,
                                size_t ws_bytes, size_t lat_off, size_t hall_off, unsigned* sync, float* y, bool last_only, int B, int T,
                                int num_cus, const Guard& gd, hipStream_t s) {
    using namespace lz;
    if (ws_bytes > 0x7fffffffull) return hipErrorInvalidValue;
    static PerDeviceFlag attr_flag; bool& attr_set = attr_flag.cur();
    if (!attr_set) {
        for (const void* f : {reinterpret_cast<const void*>(lat1_kernel<false>), reinterpret_cast<const void*>(lat1_kernel<true>)}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, l1::LDS_BYTES);
            if (e != hipSuccess) return e;
        }
        attr_set = true;
    }
    // TIP_LAT1_SPREAD=1: workers on consecutive block ids (all XCDs, coherent stores) instead of 8 ids apart (one XCD): measurement
    static int spread = -1;
    if (spread < 0) spread = (getenv("TIP_LAT1_SPREAD") && getenv("TIP_LAT1_SPREAD")[0] == '1') ? 1 : 0;
    Lat1Args a;
    a.stride = spread ? 1 : 8;
    a.nw = spread ? 64 : 32;
    if (a.nw * (spread ? 1 : 8) > num_cus * 8 || a.nw > num_cus) return hipErrorCooperativeLaunchTooLarge;   // (a CU-masked stream)
    {
        static PerDeviceInt occ_dev; int& occ = occ_dev.cur();   // every worker must be resident while the others wait for it
        hipError_t ce = check_coresident(lat1_kernel<false>, l1::THREADS, (size_t)l1::LDS_BYTES, a.nw, spread ? num_cus : num_cus / 8, &occ);
        if (ce != hipSuccess) return ce;
    }
    const size_t bt = (size_t)B * T;
    const size_t xa = lat_off, xb = xa + bt * 256, o = xb + bt * 256 + bt * 768, hid = o + bt * 256, ihb = hid + bt * 1024;
    const size_t st0 = ihb + bt * 512, st1 = st0 + (size_t)B * 48 * 2, gran = st1 + (size_t)B * 48 * 2;
    a.wts = fused_w; a.wbytes = (int)(fused_packed_floats(d) * 4);
    a.x_imu = x_imu; a.x_s = x_s; a.keep_mask = keep_mask; a.keep_scale = keep_scale;
    a.ws = ws_base; a.ws_bytes = (int)ws_bytes;
    a.xa = (int)xa; a.xb = (int)xb; a.o = (int)o; a.hid = (int)hid; a.ihb = (int)ihb; a.st0 = (int)st0; a.st1 = (int)st1; a.gran = (int)gran;
    a.hall = (int)hall_off;
    a.whh_frag = whh_frag; a.out_frag = out_frag; a.out_bias = out_bias;
    a.y = y; a.ldy = d.S; a.N = d.S;
    a.sync = sync;
    a.B = B; a.T = T; a.NI = d.n_imu_total; a.S = d.S; a.L = d.L; a.last_only = last_only ? 1 : 0;
    a.gd = gd;
    static int trace = -1;
    if (trace < 0) trace = getenv("TIP_LAT1_TRACE") ? 1 : 0;
    if (trace) hipLaunchKernelGGL(lat1_kernel<true>, dim3(a.nw * a.stride), dim3(l1::THREADS), l1::LDS_BYTES, s, a);
    else hipLaunchKernelGGL(lat1_kernel<false>, dim3(a.nw * a.stride), dim3(l1::THREADS), l1::LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace tip
extern "C" int tip_debug_read_lat1_trace(unsigned long long* out, int n) {
    if (!out || n < 0 || n > 128) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(tip::g_lat1_trace), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -5;
}

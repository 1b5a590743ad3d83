// tip_s16.hip — EXPLORATORY plan "fused16" (TIP_PLAN_FUSED16, opt-in, never AUTO's choice, never the headline):
// the fused one-window encoder with every GEMM's fp32 operands EMULATED on the fp16 matrix cores.
//
// gfx950 multiplies f32-input MFMAs at 1/16 of the f16 rate (MI355X_MICROARCH.md), and the encoder is bound by exactly that
// (docs/DESIGN_NOTES_r01-r03.md section 9).  Here both operands of a product are split
//     x = xh + xl * 2^-11,   xh = fp16(x),   xl = fp16((x - xh) * 2^11)
// (weights once, at pack time; activations in the epilogue that produces them) and
//     x * w ~= xh wh + (xh wl + xl wh) * 2^-11              (the xl wl * 2^-22 term is dropped)
// runs as THREE v_mfma_f32_16x16x32_f16 per 16 x 16 x 32 block with fp32 accumulation — products of two fp16 numbers are exact
// in fp32 — into two accumulators (the hh sum and the cross sum), combined once per phase.  Operands keep 22 significant bits
// (fp32: 24); measured against fp64 the result is NOT worse than an fp32 MFMA chain (tools/probes/ffn_split16_probe.hip: 0.4-0.5x
// the error of a sequential fp32 dot product for gains 1 ... 16), because the accumulation, not the operand rounding, dominates.
// The attention core (S = K Q^T, softmax, P V: 2.5 % of the FLOPs), LayerNorm, the residual stream, biases and every epilogue
// stay in fp32 exactly as in fused_encoder_kernel.  dtype of the plan: "f32 emulated as split fp16 (22-bit operands), fp32
// accumulate" — reported under bench.py's `extra` with that label only.
//
// Structure = fused_encoder_kernel (tip_fused.hip): one 512-thread workgroup per window, 48 padded rows, the residual stream X
// [48][260] fp32 in LDS, weights streamed from L2 in fragment order, one head pair per wave from projection to attention output
// in registers (attention_head_regs' operand trick: swapped MFMA operands leave Q^T / K^T in the accumulators).  New: every
// GEMM input lives in LDS as two fp16 planes [48][272] (hi, lo; 544-byte rows: conflict-free 16-byte fragment reads), written by
// the epilogue / LayerNorm pass that produces it.  LDS 154 368 B.
// Weight image: the fused fp32 section re-expressed as [col block][32-k block][hi | lo][64 lanes][8 halfs] at the same float
// offsets (4 bytes per weight either way), built from that section by s16_pack_host / launch_s16_repack, so every fold of the
// fp32 image (channel shuffle, root-velocity columns, 1/sqrt(d_head), b_ih + b_hh) carries over.
// Reference: /root/reference/simple_transformer_with_state.py:63-99.
#include <stdlib.h>
#include <string.h>

#include "tip_internal.h"
#include "tip_layernorm.h"

// The exploratory plans are compiled into the MEASUREMENT build only (csrc/Makefile: `make measure`, -DTIP_EXPLORATORY).  In the
// default library every entry point of this file is a stub that says "not served": tip_set_option refuses TIP_PLAN_FUSED16 /
// TIP_PLAN_GENERAL16 and a non-zero TIP_OPT_PACK_SPLIT16 with TIP_ERR_UNSUPPORTED_CONFIG.
#ifndef TIP_EXPLORATORY
namespace tip {
bool pgemm16_shape_ok(int, int, int) { return false; }
hipError_t launch_pgemm16(const float*, int, const float*, size_t, const float*, const float*, int, float*, int, int, int, int, int,
                          hipStream_t) { return hipErrorInvalidValue; }
void s16_convert_host(const float*, float*, int, int) {}
hipError_t launch_s16_convert(const float*, float*, int, int, hipStream_t) { return hipErrorInvalidValue; }
bool s16_supported(const Dims&, int) { return false; }
size_t s16_packed_floats(const Dims&) { return 0; }
void s16_pack_host(const Dims&, const float*, float*) {}
hipError_t launch_s16_repack(const Dims&, const float*, float*, hipStream_t) { return hipErrorInvalidValue; }
hipError_t launch_fused_encoder_s16(const Dims&, const float*, const float*, const float*, const float*, const float*, float, float*,
                                    float*, int, int, int, hipStream_t) { return hipErrorInvalidValue; }
}  // namespace tip
extern "C" int tip_debug_read_s16_trace(unsigned long long*, int) { return -1; }
#else

namespace tip {

typedef _Float16 s16h8 __attribute__((ext_vector_type(8)));
typedef float s16f4 __attribute__((ext_vector_type(4)));

namespace sz {
constexpr int D = 256, F = 1024, RP = 48, RB = 3, TMAX = 40, R = 512, KIN = 224;
constexpr int LDX = D + 4;            // fp32 residual stream
constexpr int LDA = D + 16;           // halfs: split planes
constexpr int PL = RP * LDA;          // halfs per plane
constexpr int X_BYTES = RP * LDX * 4;
constexpr int LDS_BYTES = X_BYTES + 4 * PL * 2;   // X | Ah | Al | Ch | Cl = 154 368 B
constexpr int THREADS = 512;
constexpr float SC = 2048.f, ISC = 1.f / 2048.f;
// the fused fp32 section's offsets (floats; tip_fused.hip, namespace fz) — the s16 section uses the same ones
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
}  // namespace sz

bool s16_supported(const Dims& d, int T) { return fused_supported(d, T) && fused_has_rnn_ih(d); }
size_t s16_packed_floats(const Dims& d) { return (fused_packed_floats(d) && fused_has_rnn_ih(d)) ? fused_packed_floats(d) : 0; }

// the weight matrices of the fused section: (float offset, N, K padded to a multiple of 32)
struct S16Mat {
    size_t off;
    int N, K;
};
static int s16_mats(const Dims& d, std::vector<S16Mat>& m) {
    using namespace sz;
    m.clear();
    m.push_back({IN_W, D, KIN});
    for (int l = 0; l < d.L; ++l) {
        const size_t L = LAYER0 + (size_t)l * LAYER_FLOATS;
        m.push_back({L + QKV_W, 3 * D, D});
        m.push_back({L + WO_W, D, D});
        m.push_back({L + W1_W, F, D});
        m.push_back({L + W2_W, D, F});
    }
    m.push_back({LAYER0 + (size_t)d.L * LAYER_FLOATS, R, D});
    return (int)m.size();
}

// element (col n, k) of a matrix stored in 16x16x4 fragment order [N/16][K/16][64][4]
__host__ __device__ __forceinline__ size_t s16_src_index(int n, int k, int K) {
    return (((size_t)(n >> 4) * (K >> 4) + (k >> 4)) * 64 + ((k & 15) >> 2) * 16 + (n & 15)) * 4 + (k & 3);
}

void s16_pack_host(const Dims& d, const float* fused_src, float* dst) {
    std::vector<S16Mat> mats;
    s16_mats(d, mats);
    memset(dst, 0, s16_packed_floats(d) * sizeof(float));
    for (const S16Mat& m : mats) {
        _Float16* out = reinterpret_cast<_Float16*>(dst + m.off);
        const float* src = fused_src + m.off;
        const int KB = m.K / 32;
        for (int nb = 0; nb < m.N / 16; ++nb)
            for (int kb = 0; kb < KB; ++kb)
                for (int lane = 0; lane < 64; ++lane)
                    for (int i = 0; i < 8; ++i) {
                        const int n = nb * 16 + (lane & 15), k = kb * 32 + (lane >> 4) * 8 + i;
                        const float w = src[s16_src_index(n, k, m.K)];
                        const _Float16 h = (_Float16)w;
                        const _Float16 l = (_Float16)((w - (float)h) * sz::SC);
                        const size_t base = ((size_t)(nb * KB + kb) * 2) * 512;
                        out[base + lane * 8 + i] = h;
                        out[base + 512 + lane * 8 + i] = l;
                    }
    }
}

__global__ void s16_repack_kernel(const float* __restrict__ src, _Float16* __restrict__ out, int N, int K) {
    const int KB = K / 32;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // (nb, kb, lane)
    if (idx >= (long long)(N / 16) * KB * 64) return;
    const int lane = (int)(idx & 63);
    const int kb = (int)((idx >> 6) % KB), nb = (int)((idx >> 6) / KB);
    const int n = nb * 16 + (lane & 15);
    s16h8 h, l;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = kb * 32 + (lane >> 4) * 8 + i;
        const float w = src[s16_src_index(n, k, K)];
        h[i] = (_Float16)w;
        l[i] = (_Float16)((w - (float)h[i]) * sz::SC);
    }
    const size_t base = ((size_t)(nb * KB + kb) * 2) * 512;
    *reinterpret_cast<s16h8*>(out + base + lane * 8) = h;
    *reinterpret_cast<s16h8*>(out + base + 512 + lane * 8) = l;
}

hipError_t launch_s16_repack(const Dims& d, const float* fused_src, float* dst, hipStream_t s) {
    std::vector<S16Mat> mats;
    s16_mats(d, mats);
    hipError_t e = hipMemsetAsync(dst, 0, s16_packed_floats(d) * sizeof(float), s);
    if (e != hipSuccess) return e;
    for (const S16Mat& m : mats) {
        const long long n = (long long)(m.N / 16) * (m.K / 32) * 64;
        hipLaunchKernelGGL(s16_repack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, fused_src + m.off,
                           reinterpret_cast<_Float16*>(dst + m.off), m.N, m.K);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void s16_split(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x;
    l = (_Float16)((x - (float)h) * sz::SC);
}

// one fragment: lane l gets bytes [16 l, 16 l + 16) of the 1-KiB block at byte offset soff of the s16 section
__device__ __forceinline__ s16h8 s16_ldw(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(s16h8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// acc_h[r][n] += Ah(rows of block r) Wh(block n);  acc_x[r][n] += Ah Wl + Al Wh     for k-blocks 0 .. KB-1
//   Ap / Alp: this lane's fragment address in the hi / lo plane (row l15 of block 0, halfs 8 lg ..); block n of the weight is
//   at byte offset soff + n * nstride_b, k-block kb at + kb * 2048 (hi) / + 1024 (lo).
//   Blocks n < PLAIN_FROM run with SWAPPED operands (weights as A): the accumulator holds the TRANSPOSED tile — lane (l15, lg) =
//   (output channels 16 n + 4 lg + e, row 16 r + l15) — i.e. four CONSECUTIVE channels of one row per lane: every epilogue is a
//   16-byte fp32 or 8-byte fp16 access instead of four scattered scalars, and Q^T / K^T are what the attention wants anyway.
//   Blocks n >= PLAIN_FROM keep the plain layout (rows 16 r + 4 lg + e, channel 16 n + l15): V, the B operand of P V.
template <int NBW>
struct S16Ring {
    s16h8 bh[3][NBW], bl[3][NBW];
};
// slots 0, 1 <- k-blocks 0, 1 of the phase whose first block is at soff
template <int NBW>
__device__ __forceinline__ void s16_prime(S16Ring<NBW>& g, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff, int nstride_b) {
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int n = 0; n < NBW; ++n) {
            g.bh[p][n] = s16_ldw(rsrc, voff, soff + n * nstride_b + p * 2048);
            g.bl[p][n] = s16_ldw(rsrc, voff, soff + n * nstride_b + p * 2048 + 1024);
        }
}
// the phase proper on a PRIMED ring.  KB % 3 == 2 (8 k-blocks) leaves slots 0, 1 free at the end: the caller primes them for the
// next phase right behind the last MFMAs, in front of its epilogue and barrier (cross-phase prefetch).
template <int NBW, int KB, int PLAIN_FROM = NBW, int RBN = sz::RB, int LD = sz::LDA>
__device__ __forceinline__ void s16_gemm_run(s16f4 (&acc_h)[RBN][NBW], s16f4 (&acc_x)[RBN][NBW], const _Float16* Ap,
                                             const _Float16* Alp, S16Ring<NBW>& g, __amdgpu_buffer_rsrc_t rsrc, int voff, int soff,
                                             int nstride_b) {
    // A fragments (LDS) one k-block ahead as well: with the scheduling barrier per k-block they would otherwise be requested and
    // waited for in front of every block's MFMAs
    s16h8 ah[2][RBN], al[2][RBN];
#pragma unroll
    for (int r = 0; r < RBN; ++r) {
        ah[0][r] = *reinterpret_cast<const s16h8*>(Ap + r * 16 * LD);
        al[0][r] = *reinterpret_cast<const s16h8*>(Alp + r * 16 * LD);
    }
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        if (kb + 2 < KB) {
#pragma unroll
            for (int n = 0; n < NBW; ++n) {
                g.bh[(kb + 2) % 3][n] = s16_ldw(rsrc, voff, soff + n * nstride_b + (kb + 2) * 2048);
                g.bl[(kb + 2) % 3][n] = s16_ldw(rsrc, voff, soff + n * nstride_b + (kb + 2) * 2048 + 1024);
            }
        }
        if (kb + 1 < KB) {
#pragma unroll
            for (int r = 0; r < RBN; ++r) {
                ah[(kb + 1) & 1][r] = *reinterpret_cast<const s16h8*>(Ap + r * 16 * LD + (kb + 1) * 32);
                al[(kb + 1) & 1][r] = *reinterpret_cast<const s16h8*>(Alp + r * 16 * LD + (kb + 1) * 32);
            }
        }
        const int c = kb % 3, a = kb & 1;
        // three sweeps over the tiles (hh, hl, lh) rather than three MFMAs per tile: the two cross products of a tile go into the
        // same accumulator, and back to back the second would wait for the first's result
#pragma unroll
        for (int pass = 0; pass < 3; ++pass)
#pragma unroll
            for (int r = 0; r < RBN; ++r)
#pragma unroll
                for (int n = 0; n < NBW; ++n) {
                    const s16h8 wv = pass == 1 ? g.bl[c][n] : g.bh[c][n];
                    const s16h8 av = pass == 2 ? al[a][r] : ah[a][r];
                    s16f4& acc = pass == 0 ? acc_h[r][n] : acc_x[r][n];
                    acc = n < PLAIN_FROM ? __builtin_amdgcn_mfma_f32_16x16x32_f16(wv, av, acc, 0, 0, 0)
                                         : __builtin_amdgcn_mfma_f32_16x16x32_f16(av, wv, acc, 0, 0, 0);
                }
        __builtin_amdgcn_sched_barrier(0);
    }
}
// prime + run in one go (phases without a cross-phase prefetch)
template <int NBW, int KB, int PLAIN_FROM = NBW>
__device__ __forceinline__ void s16_gemm(s16f4 (&acc_h)[sz::RB][NBW], s16f4 (&acc_x)[sz::RB][NBW], const _Float16* Ap, const _Float16* Alp,
                                         __amdgpu_buffer_rsrc_t rsrc, int voff, int soff, int nstride_b) {
    S16Ring<NBW> g;
    s16_prime<NBW>(g, rsrc, voff, soff, nstride_b);
    s16_gemm_run<NBW, KB, PLAIN_FROM>(acc_h, acc_x, Ap, Alp, g, rsrc, voff, soff, nstride_b);
}

typedef _Float16 s16h4 __attribute__((ext_vector_type(4)));
// four consecutive values -> their hi / lo halfs at p_hi / p_lo (8-byte stores)
__device__ __forceinline__ void s16_split4_store(const s16f4& v, _Float16* p_hi, _Float16* p_lo) {
    s16h4 h, l;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = (_Float16)v[i];
        l[i] = (_Float16)((v[i] - (float)h[i]) * sz::SC);
    }
    *reinterpret_cast<s16h4*>(p_hi) = h;
    *reinterpret_cast<s16h4*>(p_lo) = l;
}

// LayerNorm of rows 0 .. 47 of X (tip_layernorm.h, layernorm_rows16: sixteen lanes per row, DPP row sums — same arithmetic), writing
// the normalised rows to X AND, split, to the planes the next GEMM reads: no separate split pass, no barrier in between
__device__ __forceinline__ void s16_layernorm_split(float* X, const float* __restrict__ g, const float* __restrict__ be, _Float16* Ph,
                                                    _Float16* Pl, int wave, int lane) {
    constexpr int DCOLS = 256, LD = sz::LDX;
    const int q = lane & 15, sub = lane >> 4;
    float4 gg[4], bb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        gg[j] = *reinterpret_cast<const float4*>(g + (q + 16 * j) * 4);
        bb[j] = *reinterpret_cast<const float4*>(be + (q + 16 * j) * 4);
    }
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        if (p * 32 + wave * 4 < sz::RP) {
            const int row = p * 32 + wave * 4 + sub;
            float* xr = X + row * LD + q * 4;
            float4 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float4*>(xr + j * 64);
            float s = ((v[0].x + v[0].y) + (v[0].z + v[0].w)) + ((v[1].x + v[1].y) + (v[1].z + v[1].w));
            s += ((v[2].x + v[2].y) + (v[2].z + v[2].w)) + ((v[3].x + v[3].y) + (v[3].z + v[3].w));
            const float mean = row16_sum(s) * (1.f / DCOLS);
            float qs[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j].x -= mean; v[j].y -= mean; v[j].z -= mean; v[j].w -= mean;
                qs[j] = (v[j].x * v[j].x + v[j].y * v[j].y) + (v[j].z * v[j].z + v[j].w * v[j].w);
            }
            const float var = row16_sum((qs[0] + qs[1]) + (qs[2] + qs[3])) * (1.f / DCOLS);
            const float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s16f4 o;
                o[0] = v[j].x * rstd * gg[j].x + bb[j].x;
                o[1] = v[j].y * rstd * gg[j].y + bb[j].y;
                o[2] = v[j].z * rstd * gg[j].z + bb[j].z;
                o[3] = v[j].w * rstd * gg[j].w + bb[j].w;
                *reinterpret_cast<s16f4*>(xr + j * 64) = o;
                s16_split4_store(o, Ph + row * sz::LDA + (q + 16 * j) * 4, Pl + row * sz::LDA + (q + 16 * j) * 4);
            }
        }
    }
}

template <int NBW>
__device__ __forceinline__ void s16_zero(s16f4 (&a)[sz::RB][NBW], s16f4 (&b)[sz::RB][NBW]) {
#pragma unroll
    for (int r = 0; r < sz::RB; ++r)
#pragma unroll
        for (int n = 0; n < NBW; ++n) a[r][n] = b[r][n] = (s16f4){0.f, 0.f, 0.f, 0.f};
}

// causal attention of one head from register tiles (tip_attention.h, attention_head_regs — fp32 16x16x4 MFMAs, unchanged), the
// output written as the SPLIT planes the out-projection reads
__device__ __forceinline__ void s16_attention_head(const s16f4 (&qt)[3], const s16f4 (&kt)[3], const s16f4 (&v)[3], _Float16* Oh,
                                                   _Float16* Ol, int c0, int lane) {
    constexpr int RB = 3;
    const int l15 = lane & 15, lg = lane >> 4;
    s16f4 S[RB][RB];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int cb = 0; cb <= r; ++cb) {
            s16f4 t = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int e = 0; e < 4; ++e) t = __builtin_amdgcn_mfma_f32_16x16x4f32(kt[cb][e], qt[r][e], t, 0, 0, 0);
            S[r][cb] = t;
        }
    float mx[RB], rsum[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        float m = -INFINITY;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (lg * 4 + e <= l15) m = fmaxf(m, S[r][r][e]);
#pragma unroll
            for (int cb = 0; cb < r; ++cb) m = fmaxf(m, S[r][cb][e]);
        }
        mx[r] = m;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) mx[r] = lg4_max(mx[r]);
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        float sm = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int cb = 0; cb < r; ++cb) {
                const float p = __expf(S[r][cb][e] - mx[r]);
                S[r][cb][e] = p;
                sm += p;
            }
            const float pd = (lg * 4 + e <= l15) ? __expf(S[r][r][e] - mx[r]) : 0.f;
            S[r][r][e] = pd;
            sm += pd;
        }
        rsum[r] = sm;
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) rsum[r] = lg4_sum(rsum[r]);
#pragma unroll
    for (int r = 0; r < RB; ++r) rsum[r] = 1.0f / rsum[r];
    // P V with the operands the other way round (V tile as A: channel l15 / keys 4 lg + e; S tile as B: query l15 / the same keys):
    // the output tile comes out TRANSPOSED — lane (l15, lg) = (channels 4 lg + e, query l15) — so a lane normalises with its own
    // row sum (no shuffle) and stores four consecutive channels of one row: 8-byte writes into the split planes
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        s16f4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb <= r; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) o = __builtin_amdgcn_mfma_f32_16x16x4f32(v[kb][e], S[r][kb][e], o, 0, 0, 0);
        o *= rsum[r];
        s16_split4_store(o, Oh + (r * 16 + l15) * sz::LDA + c0 + lg * 4, Ol + (r * 16 + l15) * sz::LDA + c0 + lg * 4);
    }
}

// measurement only (TIP_S16_TRACE=1): s_memtime stamps of workgroup 0 / thread 0 (layer 1 for the per-layer slots)
__device__ unsigned long long g_s16_trace[64];
#define S16_STAMP(slot, cond) do { if (TRACE && blockIdx.x == 0 && tid == 0 && (cond)) g_s16_trace[slot] = __builtin_amdgcn_s_memtime(); } while (0)

template <bool TRACE>
__global__ __launch_bounds__(sz::THREADS) void fused_encoder_s16_kernel(
    const float* __restrict__ wts, const float* __restrict__ w16, const float* __restrict__ x_imu, const float* __restrict__ x_s,
    const float* __restrict__ keep_mask, float keep_scale, float* __restrict__ ih_out, unsigned* __restrict__ hall_sentinel, int B,
    int T, int NI, int S, int L, int wbytes) {
    using namespace sz;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* X = reinterpret_cast<float*>(smem_raw);
    _Float16* Ah = reinterpret_cast<_Float16*>(smem_raw + X_BYTES);
    _Float16* Al = Ah + PL;
    _Float16* Ch = Al + PL;
    _Float16* Cl = Ch + PL;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w16), 0, wbytes, 0x00020000);
    const int voff = lane * 16;
    const int aoff = l15 * LDA + lg * 8;   // this lane's A-fragment offset inside a 16-row block of a split plane (halfs)
    // transposed accumulator tile (r, column block nb): lane = row 16 r + l15, channels 16 nb + 4 lg .. + 3
    auto bias4 = [&](const float* bp, int nb) -> s16f4 {
        const float4 b = *reinterpret_cast<const float4*>(bp + nb * 16 + lg * 4);
        return (s16f4){b.x, b.y, b.z, b.w};
    };

    for (int win = blockIdx.x; win < B; win += gridDim.x) {
        S16_STAMP(0, true);
        // ---- prologue (:63-78): U = [x_imu | scrubbed, masked x_s | 0] as split planes (Ch, Cl), K padded to 224 ----
        {
            uint4* z = reinterpret_cast<uint4*>(Ch);          // both C planes, contiguous
            for (int i = tid; i < 2 * PL * 2 / 16; i += THREADS) z[i] = make_uint4(0u, 0u, 0u, 0u);
        }
        __syncthreads();
        {
            // rows wave, wave + 8, ... (five per wave at T = 40): unit-stride loads, no index division, and ALL of a wave's loads in
            // flight before the first is used (one row after the other was a chain of five HBM round trips)
            constexpr int NR = (TMAX + 7) / 8;
            float vi[NR][2], vs[NR][3], vk[NR][3];
            // (unconditional loads from clamped addresses: a guarded load `ok ? p[i] : 0` compiles to a branch with a full wait
            //  behind it — 25 dependent HBM round trips, 35 k cycles measured)
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
                    vk[j][q] = km[c] * keep_scale;
                }
            }
            // LDS addresses: one base per plane and part + COMPILE-TIME offsets (8 j rows, 64 q columns) — with the row / column
            // arithmetic left in the index the compiler materialised forty address registers and spilled them
            _Float16* ph_i = Ch + wave * LDA + lane;
            _Float16* pl_i = Cl + wave * LDA + lane;
            _Float16* ph_s = ph_i + NI;
            _Float16* pl_s = pl_i + NI;
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int r = wave + 8 * j;
                if (r < T) {
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        if (lane + 64 * q < NI) {
                            _Float16 h, l;
                            s16_split(vi[j][q], h, l);
                            ph_i[8 * j * LDA + 64 * q] = h;
                            pl_i[8 * j * LDA + 64 * q] = l;
                        }
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        if (lane + 64 * q < S) {
                            float v = vs[j][q];
                            if (v != v) v = 0.f;          // :65
                            if (keep_mask) v *= vk[j][q];  // :77 (x * mask * 1/(1-p))
                            _Float16 h, l;
                            s16_split(v, h, l);
                            ph_s[8 * j * LDA + 64 * q] = h;
                            pl_s[8 * j * LDA + 64 * q] = l;
                        }
                }
            }
        }
        __syncthreads();
        S16_STAMP(1, true);
        // ---- in_linear (:79) + channel shuffle (folded into the packed rows): X and its split copy (Ah, Al) ----
        {
            s16f4 ach[RB][2], acx[RB][2];
            s16_zero<2>(ach, acx);
            s16_gemm<2, KIN / 32>(ach, acx, Ch + aoff, Cl + aoff, rsrc, voff, (int)(IN_W * 4) + (wave * 2) * (KIN / 32) * 2048,
                                  (KIN / 32) * 2048);
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const s16f4 bv = bias4(wts + IN_B, wave * 2 + n);
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    const s16f4 v = ach[r][n] + acx[r][n] * ISC + bv;
                    const int row = r * 16 + l15, col = (wave * 2 + n) * 16 + lg * 4;
                    *reinterpret_cast<s16f4*>(X + row * LDX + col) = v;
                    s16_split4_store(v, Ah + row * LDA + col, Al + row * LDA + col);
                }
            }
        }
        __syncthreads();
        S16_STAMP(2, true);

#pragma unroll 1
        for (int layer = 0; layer < L; ++layer) {
            S16_STAMP(8, layer == 1);
            const float* LW = wts + LAYER0 + (size_t)layer * LAYER_FLOATS;
            const int lb = (int)((LAYER0 + (size_t)layer * LAYER_FLOATS) * 4);   // byte offset of the layer in either section
            // ---- Q | K | V projection + attention: heads 2 wave, 2 wave + 1, projection to attention output in registers ----
#pragma unroll 1
            for (int hh = 0; hh < 2; ++hh) {
                const int head = wave * 2 + hh;
                s16f4 ph[RB][3], px[RB][3];
                s16_zero<3>(ph, px);
                // column blocks of the packed QKV weight: Q head -> block head, K -> 16 + head, V -> 32 + head (V in the plain layout)
                s16_gemm<3, D / 32, 2>(ph, px, Ah + aoff, Al + aoff, rsrc, voff, lb + (int)(QKV_W * 4) + head * (D / 32) * 2048,
                                       16 * (D / 32) * 2048);
                s16f4 qt[RB], kt[RB], vv[RB];
                const s16f4 bq = bias4(LW + QKV_B, head), bk = bias4(LW + QKV_B + D, head);
                const float bvv = LW[QKV_B + 2 * D + head * 16 + l15];
#pragma unroll
                for (int r = 0; r < RB; ++r) {
                    qt[r] = ph[r][0] + px[r][0] * ISC + bq;
                    kt[r] = ph[r][1] + px[r][1] * ISC + bk;
                    vv[r] = ph[r][2] + px[r][2] * ISC + bvv;
                }
                S16_STAMP(9 + 2 * hh, layer == 1);
                s16_attention_head(qt, kt, vv, Ch, Cl, head * 16, lane);
                S16_STAMP(10 + 2 * hh, layer == 1);
            }
            __syncthreads();
            S16_STAMP(13, layer == 1);
            // ---- out-projection + residual, LayerNorm1 ----
            {
                s16f4 ach[RB][2], acx[RB][2];
                s16_zero<2>(ach, acx);
                s16_gemm<2, D / 32>(ach, acx, Ch + aoff, Cl + aoff, rsrc, voff, lb + (int)(WO_W * 4) + (wave * 2) * (D / 32) * 2048,
                                    (D / 32) * 2048);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const s16f4 bv = bias4(LW + WO_B, wave * 2 + n);
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        s16f4* xp = reinterpret_cast<s16f4*>(X + (r * 16 + l15) * LDX + (wave * 2 + n) * 16 + lg * 4);
                        *xp = *xp + (ach[r][n] + acx[r][n] * ISC + bv);
                    }
                }
            }
            __syncthreads();
            S16_STAMP(14, layer == 1);
            s16_layernorm_split(X, LW + G1, LW + BE1, Ah, Al, wave, lane);
            __syncthreads();
            S16_STAMP(16, layer == 1);
            // ---- feed-forward: 4 hidden chunks of 256; linear2 accumulates in registers.  ONE weight ring for the eight phases: each
            //      phase primes the next one's first two k-blocks right behind its last MFMAs, so they travel under the epilogue and
            //      the barrier instead of in front of the next phase's first MFMA ----
            {
                s16f4 oh[RB][2], ox[RB][2];
                s16_zero<2>(oh, ox);
                S16Ring<2> g;
                const int w1s = (D / 32) * 2048, w2s = (F / 32) * 2048;
                auto w1off = [&](int f) { return lb + (int)(W1_W * 4) + (f * 16 + wave * 2) * (D / 32) * 2048; };
                auto w2off = [&](int f) { return lb + (int)(W2_W * 4) + ((wave * 2) * (F / 32) + f * 8) * 2048; };
                s16_prime<2>(g, rsrc, voff, w1off(0), w1s);
#pragma unroll 1
                for (int f = 0; f < 4; ++f) {
                    {
                        s16f4 ach[RB][2], acx[RB][2];
                        s16_zero<2>(ach, acx);
                        s16_gemm_run<2, D / 32>(ach, acx, Ah + aoff, Al + aoff, g, rsrc, voff, w1off(f), w1s);
                        s16_prime<2>(g, rsrc, voff, w2off(f), w2s);
#pragma unroll
                        for (int n = 0; n < 2; ++n) {
                            const s16f4 bv = bias4(LW + W1_B + f * 256, wave * 2 + n);
#pragma unroll
                            for (int r = 0; r < RB; ++r) {
                                s16f4 v = ach[r][n] + acx[r][n] * ISC + bv;
#pragma unroll
                                for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                                const int o = (r * 16 + l15) * LDA + (wave * 2 + n) * 16 + lg * 4;
                                s16_split4_store(v, Ch + o, Cl + o);
                            }
                        }
                    }
                    S16_STAMP(17 + 3 * f, layer == 1);
                    __syncthreads();
                    S16_STAMP(18 + 3 * f, layer == 1);
                    s16_gemm_run<2, 8>(oh, ox, Ch + aoff, Cl + aoff, g, rsrc, voff, w2off(f), w2s);
                    if (f < 3) s16_prime<2>(g, rsrc, voff, w1off(f + 1), w1s);
                    __syncthreads();
                    S16_STAMP(19 + 3 * f, layer == 1);
                }
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const s16f4 bv = bias4(LW + W2_B, wave * 2 + n);
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        s16f4* xp = reinterpret_cast<s16f4*>(X + (r * 16 + l15) * LDX + (wave * 2 + n) * 16 + lg * 4);
                        *xp = *xp + (oh[r][n] + ox[r][n] * ISC + bv);
                    }
                }
            }
            __syncthreads();
            S16_STAMP(30, layer == 1);
            s16_layernorm_split(X, LW + G2, LW + BE2, Ah, Al, wave, lane);
            __syncthreads();
            S16_STAMP(31, layer == 1);
        }
        S16_STAMP(40, true);
        // ---- RNN input projection: IH = X W_ih^T + (b_ih + b_hh) -> HBM (:98), 16-byte stores ----
        {
            const size_t ih_off = LAYER0 + (size_t)L * LAYER_FLOATS;
            float* io = ih_out + (size_t)win * T * R;
#pragma unroll 1
            for (int half = 0; half < 2; ++half) {
                s16f4 ach[RB][2], acx[RB][2];
                s16_zero<2>(ach, acx);
                const int nb0 = wave * 4 + half * 2;
                s16_gemm<2, D / 32>(ach, acx, Ah + aoff, Al + aoff, rsrc, voff, (int)(ih_off * 4) + nb0 * (D / 32) * 2048, (D / 32) * 2048);
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const s16f4 bv = bias4(wts + ih_off + (size_t)R * D, nb0 + n);
#pragma unroll
                    for (int r = 0; r < RB; ++r) {
                        const int row = r * 16 + l15;
                        if (row < T) *reinterpret_cast<s16f4*>(io + (size_t)row * R + (nb0 + n) * 16 + lg * 4) = ach[r][n] + acx[r][n] * ISC + bv;
                    }
                }
            }
        }
        if (hall_sentinel) {
            uint4* hp = reinterpret_cast<uint4*>(hall_sentinel + (size_t)win * T * R);
            for (int i = tid; i < T * (R / 4); i += THREADS) hp[i] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
        }
        __syncthreads();
        S16_STAMP(41, true);
    }
}

extern "C" int tip_debug_read_s16_trace(unsigned long long* out, int n) {
    if (!out || n < 0 || n > 64) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_s16_trace), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -5;
}

hipError_t launch_fused_encoder_s16(const Dims& d, const float* fused_w, const float* s16_w, const float* x_imu, const float* x_s,
                                    const float* keep_mask, float keep_scale, float* ih_out, float* hall_sentinel, int B, int T,
                                    int num_cus, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    if (!s16_supported(d, T)) return hipErrorInvalidValue;
    static PerDeviceFlag attr_flag;
    bool& attr_set = attr_flag.cur();
    if (!attr_set) {
        for (const void* f : {reinterpret_cast<const void*>(fused_encoder_s16_kernel<false>), reinterpret_cast<const void*>(fused_encoder_s16_kernel<true>)}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, sz::LDS_BYTES);
            if (e != hipSuccess) return e;
        }
        attr_set = true;
    }
    const int grid = B < num_cus ? B : num_cus;
    static const bool trace = tip_env("TIP_S16_TRACE") && tip_env("TIP_S16_TRACE")[0] == '1';
    if (trace) {
        hipLaunchKernelGGL(fused_encoder_s16_kernel<true>, dim3(grid), dim3(sz::THREADS), sz::LDS_BYTES, s, fused_w, s16_w, x_imu, x_s, keep_mask,
                           keep_scale, ih_out, reinterpret_cast<unsigned*>(hall_sentinel), B, T, d.n_imu_total, d.S, d.L,
                           (int)(s16_packed_floats(d) * 4));
        return hipGetLastError();
    }
    hipLaunchKernelGGL(fused_encoder_s16_kernel<false>, dim3(grid), dim3(sz::THREADS), sz::LDS_BYTES, s, fused_w, s16_w, x_imu, x_s, keep_mask,
                       keep_scale, ih_out, reinterpret_cast<unsigned*>(hall_sentinel), B, T, d.n_imu_total, d.S, d.L,
                       (int)(s16_packed_floats(d) * 4));
    return hipGetLastError();
}

// =====================================================================================================================
// General plan, exploratory twin of pgemm_kernel (tip_fused2.hip / tip_pgemm.h): C[M,N] = epi(A[M,K] W^T + bias (+ res)) for the BIG
// linears of any configuration (the scaled model's d = 1024, ffn = 4096) with split-fp16 operands.  An 80-row x 256-column panel per
// 512-thread workgroup; the A panel is read from HBM / L2 as fp32, SPLIT on the way into LDS (hi / lo planes, K-chunks of 128,
// double-buffered, one barrier per chunk); the weights never touch LDS: their split copy in the packed image
// ([column block][32-k block][hi | lo][64 lanes][8 halfs], PackedLinear::s_off, present when TIP_S16_GENERAL=1 at tip_create) is
// streamed into the three-slot register ring, primed across chunk boundaries.  Transposed accumulator tiles: 16-byte epilogues.
// Selected by TIP_PLAN_GENERAL16 only.
// =====================================================================================================================
namespace pg16 {
constexpr int ROWS = 80, RB = 5, NBW = 2, KC = 128, LDA = KC + 16, THREADS = 512, COLS = 8 * NBW * 16;   // 256 columns per workgroup
constexpr int PLANE = ROWS * LDA;                       // halfs
constexpr int LDS_BYTES = 2 * 2 * PLANE * 2;            // [buffer][hi | lo]: 92 160 B
}  // namespace pg16

template <int FLAGS>   // 1 = relu, 2 = residual
__global__ __launch_bounds__(pg16::THREADS) void pgemm16_kernel(const float* __restrict__ A, int lda, const float* __restrict__ w16,
                                                                int wbytes, const float* __restrict__ bias, const float* __restrict__ res,
                                                                int ldres, float* __restrict__ C, int ldc, int M, int N, int K) {
    using namespace pg16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    _Float16* planes = reinterpret_cast<_Float16*>(smem_raw);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w16), 0, wbytes, 0x00020000);
    const int voff = lane * 16;
    const int row0 = blockIdx.y * ROWS;
    const int nb0 = blockIdx.x * (COLS / 16) + wave * NBW;
    const int KB32 = K >> 5, nchunks = K / KC;
    const int nstride = KB32 * 2048;
    const int wsoff = nb0 * nstride;
    float4 st[5];
    auto fetch = [&](int c) {
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int f = tid + u * THREADS, r = f >> 5, k4 = f & 31;
            const int rr = row0 + r < M ? row0 + r : M - 1;           // clamped: unconditional loads
            st[u] = *reinterpret_cast<const float4*>(A + (size_t)rr * lda + c * KC + k4 * 4);
            if (row0 + r >= M) st[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto stage = [&](int buf) {
        _Float16* ph = planes + buf * 2 * PLANE;
        _Float16* pl = ph + PLANE;
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int f = tid + u * THREADS, r = f >> 5, k4 = f & 31;
            const s16f4 v = {st[u].x, st[u].y, st[u].z, st[u].w};
            s16_split4_store(v, ph + r * LDA + k4 * 4, pl + r * LDA + k4 * 4);
        }
    };
    s16f4 acc_h[RB][NBW], acc_x[RB][NBW];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int n = 0; n < NBW; ++n) acc_h[r][n] = acc_x[r][n] = (s16f4){0.f, 0.f, 0.f, 0.f};
    S16Ring<NBW> g;
    s16_prime<NBW>(g, rsrc, voff, wsoff, nstride);
    fetch(0);
    stage(0);
    __syncthreads();
    const int aoff = l15 * LDA + lg * 8;
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) fetch(c + 1);
        const _Float16* ph = planes + (c & 1) * 2 * PLANE + aoff;
        s16_gemm_run<NBW, KC / 32, NBW, RB, LDA>(acc_h, acc_x, ph, ph + PLANE, g, rsrc, voff, wsoff + c * (KC / 32) * 2048, nstride);
        // next chunk's first two k-blocks (a harmless re-read of the last ones after the last chunk: the ring is dead then)
        s16_prime<NBW>(g, rsrc, voff, wsoff + (c + 1 < nchunks ? c + 1 : c) * (KC / 32) * 2048, nstride);
        if (c + 1 < nchunks) stage((c + 1) & 1);
        __syncthreads();
    }
#pragma unroll
    for (int n = 0; n < NBW; ++n) {
        const int col = (nb0 + n) * 16 + lg * 4;
        if (col >= N) continue;
        const float4 b4 = *reinterpret_cast<const float4*>(bias + col);
        const s16f4 bv = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int row = row0 + r * 16 + l15;
            if (row < M) {
                s16f4 v = acc_h[r][n] + acc_x[r][n] * sz::ISC + bv;
                if (FLAGS & 2) v = v + *reinterpret_cast<const s16f4*>(res + (size_t)row * ldres + col);
                if (FLAGS & 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] > 0.f ? v[e] : 0.f;
                }
                *reinterpret_cast<s16f4*>(C + (size_t)row * ldc + col) = v;
            }
        }
    }
}

bool pgemm16_shape_ok(int M, int N, int K) { return N % pg16::COLS == 0 && K % pg16::KC == 0 && M >= 4 * pg16::ROWS; }

hipError_t launch_pgemm16(const float* A, int lda, const float* w16, size_t w_floats, const float* bias, const float* res, int ldres,
                          float* C, int ldc, int M, int N, int K, int flags, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    if (!pgemm16_shape_ok(M, N, K) || w_floats * 4 > 0x7fffffffULL || (lda & 3) || (ldc & 3) || ((flags & 2) && (ldres & 3)))
        return hipErrorInvalidValue;
    static PerDeviceFlag attr_flag;
    bool& attr_set = attr_flag.cur();
    if (!attr_set) {
        for (const void* f : {reinterpret_cast<const void*>(pgemm16_kernel<0>), reinterpret_cast<const void*>(pgemm16_kernel<1>),
                              reinterpret_cast<const void*>(pgemm16_kernel<2>), reinterpret_cast<const void*>(pgemm16_kernel<3>)}) {
            hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, pg16::LDS_BYTES);
            if (e != hipSuccess) return e;
        }
        attr_set = true;
    }
    const dim3 grid(N / pg16::COLS, (M + pg16::ROWS - 1) / pg16::ROWS), block(pg16::THREADS);
    const int wb = (int)(w_floats * 4);
    switch (flags & 3) {
        case 0: hipLaunchKernelGGL(pgemm16_kernel<0>, grid, block, pg16::LDS_BYTES, s, A, lda, w16, wb, bias, res, ldres, C, ldc, M, N, K); break;
        case 1: hipLaunchKernelGGL(pgemm16_kernel<1>, grid, block, pg16::LDS_BYTES, s, A, lda, w16, wb, bias, res, ldres, C, ldc, M, N, K); break;
        case 2: hipLaunchKernelGGL(pgemm16_kernel<2>, grid, block, pg16::LDS_BYTES, s, A, lda, w16, wb, bias, res, ldres, C, ldc, M, N, K); break;
        default: hipLaunchKernelGGL(pgemm16_kernel<3>, grid, block, pg16::LDS_BYTES, s, A, lda, w16, wb, bias, res, ldres, C, ldc, M, N, K); break;
    }
    return hipGetLastError();
}

// split copy of ONE matrix given in 16x16x4 fragment order (the general plan's PackedLinear::f_off copies): host and device
void s16_convert_host(const float* src_frag, float* dst, int N, int K) {
    _Float16* out = reinterpret_cast<_Float16*>(dst);
    const int KB = K / 32;
    for (int nb = 0; nb < N / 16; ++nb)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                    const int n = nb * 16 + (lane & 15), k = kb * 32 + (lane >> 4) * 8 + i;
                    const float w = src_frag[s16_src_index(n, k, K)];
                    const _Float16 h = (_Float16)w;
                    const _Float16 l = (_Float16)((w - (float)h) * sz::SC);
                    const size_t base = ((size_t)(nb * KB + kb) * 2) * 512;
                    out[base + lane * 8 + i] = h;
                    out[base + 512 + lane * 8 + i] = l;
                }
}
hipError_t launch_s16_convert(const float* src_frag, float* dst, int N, int K, hipStream_t s) {
    const long long n = (long long)(N / 16) * (K / 32) * 64;
    hipLaunchKernelGGL(s16_repack_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src_frag, reinterpret_cast<_Float16*>(dst), N, K);
    return hipGetLastError();
}

}  // namespace tip
#endif  // TIP_EXPLORATORY

// tip_attention.h — causal single-head attention on the matrix cores, shared by the fused and latency plans.
#pragma once
#include <hip/hip_runtime.h>

#include "tip_layernorm.h"   // lg4_sum / lg4_max: lane ^ 16, lane ^ 32 reductions on permlane swaps

namespace tip {

typedef float f32x4_att __attribute__((ext_vector_type(4)));

// Causal attention of ONE head by ONE wave, on the matrix cores (v_mfma_f32_16x16x4_f32, K = d_head = 16):
//   S^T = K Q^T  6 lower-triangular 16x16 tiles (24 MFMAs), computed TRANSPOSED: lane holds (keys 4*lg + e, query l15).
//                That is exactly the A-operand layout of P V under the 4-k-step idiom (.x/.y/.z/.w = keys 4*lg + 0..3),
//                so the un-normalised P feeds the second product straight from its accumulators — no P^T round trip
//                through LDS.  A query's statistics live in the 4 lanes sharing l15: two xor-shuffles per reduction.
//   P V          B fragments = V^T rows (b128 from the V^T plane); the output tile (queries 4*lg + e, channel l15) takes
//                1/rowsum of its queries by shuffle from the lanes that own them.
// Q/K planes hold the head at column c0; output overwrites the head's Q columns (out-projection A operand); rows at or
// beyond row_limit are not written (the two-window kernel packs another window's rows right behind).
// DROP (the training step's dropout on the probabilities, as attention_head_regs<.., TRAIN> below): keep = hash(dkey, (bh * T + query) *
// T + key) >= thresh, kept probabilities times dscale; thresh 0 = off.
// R0, R1: the query row blocks [R0, R1) this wave takes (default: all three; the few-stream plan gives each of three waves one — a
// query's arithmetic does not depend on which wave carries its block)
template <int LDC, int LDV, bool DROP = false, int R0 = 0, int R1 = 3>
__device__ __forceinline__ void attention_head_mfma(float* Qc, float* Kc, const float* Vt, int c0, int lane, int row_limit = 48,
                                                    unsigned dkey = 0, unsigned thresh = 0, float dscale = 1.f,
                                                    unsigned long long bh = 0, int T = 0) {
    constexpr int RB = 3;   // 48 padded rows
    const int l15 = lane & 15, lg = lane >> 4;
    float4 qf[RB], kf[RB];
#pragma unroll
    for (int r = 0; r < R1; ++r) {
        qf[r] = *reinterpret_cast<const float4*>(Qc + (r * 16 + l15) * LDC + c0 + lg * 4);
        kf[r] = *reinterpret_cast<const float4*>(Kc + (r * 16 + l15) * LDC + c0 + lg * 4);
    }
    f32x4_att S[RB][RB];   // S[r][cb]: queries of block r (column l15) x keys of block cb (rows 4*lg + e)
#pragma unroll
    for (int r = R0; r < R1; ++r)
#pragma unroll
        for (int cb = 0; cb <= r; ++cb) {
            f32x4_att t = {0.f, 0.f, 0.f, 0.f};
            t = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[cb].x, qf[r].x, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[cb].y, qf[r].y, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[cb].z, qf[r].z, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[cb].w, qf[r].w, t, 0, 0, 0);
            S[r][cb] = t;
        }
    // softmax over keys for query 16r + l15 (un-normalised).  Causal mask (:56-58) only bites on the diagonal tiles:
    // key-in-block 4*lg + e <= query-in-block l15.  The three row-block reductions advance together through each shuffle.
    float mx[RB], rsum[RB];
#pragma unroll
    for (int r = R0; r < R1; ++r) {
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
    for (int r = R0; r < R1; ++r) mx[r] = lg4_max(mx[r]);
#pragma unroll
    for (int r = R0; r < R1; ++r) {
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
    for (int r = R0; r < R1; ++r) rsum[r] = lg4_sum(rsum[r]);
#pragma unroll
    for (int r = R0; r < R1; ++r) rsum[r] = 1.0f / rsum[r];
    if (DROP && thresh) {
#pragma unroll
        for (int r = R0; r < R1; ++r) {
            const unsigned pb = (unsigned)((bh * T + (r * 16 + l15)) * T);
#pragma unroll
            for (int cb = 0; cb <= r; ++cb)
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    S[r][cb][e] = tip_drop_hash_k(dkey, pb + cb * 16 + lg * 4 + e) >= thresh ? S[r][cb][e] * dscale : 0.f;
        }
    }
    // P V per query block: A = P tiles from registers, B = V^T fragments
#pragma unroll
    for (int r = R0; r < R1; ++r) {
        f32x4_att o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb <= r; ++kb) {
            const float4 vb = *reinterpret_cast<const float4*>(Vt + (c0 + l15) * LDV + kb * 16 + lg * 4);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(S[r][kb][0], vb.x, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(S[r][kb][1], vb.y, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(S[r][kb][2], vb.z, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(S[r][kb][3], vb.w, o, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float inv = __shfl(rsum[r], lg * 4 + e, 64);   // the lane whose l15 is this query
            if (r * 16 + lg * 4 + e < row_limit) Qc[(r * 16 + lg * 4 + e) * LDC + c0 + l15] = o[e] * inv;
        }
    }
}

// The same attention with Q, K and V arriving IN REGISTERS, for a wave that has just projected its own head:
//   qt[r], kt[r]  Q^T / K^T tiles of row block r: lane holds (channels 4*lg + e, row l15) — what the projection yields when
//                 its MFMA operands are swapped (weights as A, activations as B);
//   v[r]          V tile of row block r in the plain accumulator layout: (keys 4*lg + e, channel l15).
// These are precisely the A/B fragments of S^T = K Q^T and the B fragments of P V, so the head goes from projection to
// output without touching LDS and without a workgroup barrier in between.  O is written to the plane `Oc` (out-projection
// A operand) at column c0.
// TRAIN: dropout on the probabilities (keep = hash(seed, site, (bh*T + query)*T + key) >= thresh) and the softmax statistics
// (row max, 1 / row sum) of every real query written to ast[(bh*T + query)*2 ..] for the backward.
template <int LDC, bool TRAIN = false>
__device__ __forceinline__ void attention_head_regs(const f32x4_att (&qt)[3], const f32x4_att (&kt)[3], const f32x4_att (&v)[3],
                                                    float* Oc, int c0, int lane, int row_limit = 48, float* ast = nullptr,
                                                    unsigned long long bh = 0, int T = 0, unsigned long long seed = 0,
                                                    unsigned site = 0, unsigned thresh = 0, float dscale = 1.f) {
    constexpr int RB = 3;
    const int l15 = lane & 15, lg = lane >> 4;
    f32x4_att S[RB][RB];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int cb = 0; cb <= r; ++cb) {
            f32x4_att t = {0.f, 0.f, 0.f, 0.f};
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
    if (TRAIN) {
        // Straight-line code: the statistics go out through a buffer of this head's T (max, 1 / sum) pairs — lanes other than lg == 0 and
        // queries >= T aim past its extent — and the keep decisions are drawn for the pad queries too (their probabilities only feed
        // pad rows of O, which nobody reads).  With `if (q < T) { if (lg == 0) ...; if (thresh) ... }` the hash sat inside two nested
        // divergent regions per row block.
        const unsigned dkey = tip_drop_key_s(seed, site);
        const __amdgpu_buffer_rsrc_t ars = tip_rows_buffer(ast + bh * T * 2, T * 8);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int q = r * 16 + l15;
            const ln_u32x2 mr = {__float_as_uint(mx[r]), __float_as_uint(rsum[r])};
            __builtin_amdgcn_raw_buffer_store_b64(mr, ars, lg == 0 ? q * 8 : T * 8, 0, 0);
            if (thresh) {
                const unsigned pb = (unsigned)((bh * T + q) * T);
#pragma unroll
                for (int cb = 0; cb <= r; ++cb)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        S[r][cb][e] = tip_drop_hash_k(dkey, pb + cb * 16 + lg * 4 + e) >= thresh ? S[r][cb][e] * dscale : 0.f;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        f32x4_att o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb <= r; ++kb)
#pragma unroll
            for (int e = 0; e < 4; ++e) o = __builtin_amdgcn_mfma_f32_16x16x4f32(S[r][kb][e], v[kb][e], o, 0, 0, 0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float inv = __shfl(rsum[r], lg * 4 + e, 64);
            if (r * 16 + lg * 4 + e < row_limit) Oc[(r * 16 + lg * 4 + e) * LDC + c0 + l15] = o[e] * inv;
        }
    }
}

}  // namespace tip

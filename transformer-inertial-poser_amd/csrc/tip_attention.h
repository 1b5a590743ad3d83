// tip_attention.h — causal single-head attention on the matrix cores, shared by the fused and latency plans.
#pragma once
#include <hip/hip_runtime.h>

namespace tip {

typedef float f32x4_att __attribute__((ext_vector_type(4)));

// Causal attention of ONE head by ONE wave, on the matrix cores (v_mfma_f32_16x16x4_f32, K = d_head = 16):
//   S = Q K^T   6 lower-triangular 16x16 tiles (24 MFMAs); the scores stay in the C layout
//               (lane: key = l15, queries 4*lg + e), so a row max / sum is an xor-shuffle over 16 lanes
//   P V         per query block r: the un-normalised P_r is written TRANSPOSED into the head's (now dead) K slot,
//               read back as A fragments, multiplied with V^T fragments (b128), divided by the row sums at the end.
// Q/K planes hold the head at column c0; output overwrites the head's Q columns (out-projection A operand); rows at or
// beyond row_limit are not written (the two-window kernel packs another window's rows right behind).
template <int LDC, int LDV>
__device__ __forceinline__ void attention_head_mfma(float* Qc, float* Kc, const float* Vt, int c0, int lane, int row_limit = 48) {
    constexpr int RB = 3;   // 48 padded rows
    const int l15 = lane & 15, lg = lane >> 4;
    float4 qf[RB], kf[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        qf[r] = *reinterpret_cast<const float4*>(Qc + (r * 16 + l15) * LDC + c0 + lg * 4);
        kf[r] = *reinterpret_cast<const float4*>(Kc + (r * 16 + l15) * LDC + c0 + lg * 4);
    }
    f32x4_att S[RB][RB];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int cb = 0; cb <= r; ++cb) {
            f32x4_att t = {0.f, 0.f, 0.f, 0.f};
            t = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[r].x, kf[cb].x, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[r].y, kf[cb].y, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[r].z, kf[cb].z, t, 0, 0, 0);
            t = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[r].w, kf[cb].w, t, 0, 0, 0);
            S[r][cb] = t;
        }
    // row softmax (un-normalised): element e of S[r][cb] is query 16r + 4lg + e, key 16cb + l15; causal mask (:56-58)
    // only bites on the diagonal tiles: key-in-block <= query-in-block.  The 12 row reductions of a lane advance
    // TOGETHER through each xor-shuffle step, so their latencies overlap instead of forming 96 dependent hops.
    float mx[RB][4], rsum[RB][4];
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool vis = l15 <= lg * 4 + e;
            float m = vis ? S[r][r][e] : -INFINITY;
#pragma unroll
            for (int cb = 0; cb < r; ++cb) m = fmaxf(m, S[r][cb][e]);
            mx[r][e] = m;
        }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1)
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) mx[r][e] = fmaxf(mx[r][e], __shfl_xor(mx[r][e], off, 64));
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool vis = l15 <= lg * 4 + e;
            float sm = 0.f;
#pragma unroll
            for (int cb = 0; cb < r; ++cb) {
                const float p = __expf(S[r][cb][e] - mx[r][e]);
                S[r][cb][e] = p;
                sm += p;
            }
            const float pd = vis ? __expf(S[r][r][e] - mx[r][e]) : 0.f;
            S[r][r][e] = pd;
            rsum[r][e] = sm + pd;
        }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1)
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) rsum[r][e] += __shfl_xor(rsum[r][e], off, 64);
#pragma unroll
    for (int r = 0; r < RB; ++r)
#pragma unroll
        for (int e = 0; e < 4; ++e) rsum[r][e] = 1.0f / rsum[r][e];
    // P V per query block
    float* Pt = Kc + c0;   // P^T[key j][query-in-block i] at Pt[j*LDC + i]  (the head's K columns: dead after kf was read)
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int cb = 0; cb <= r; ++cb)
            *reinterpret_cast<f32x4_att*>(Pt + (cb * 16 + l15) * LDC + lg * 4) = S[r][cb];
        __builtin_amdgcn_wave_barrier();
        f32x4_att o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb <= r; ++kb) {
            const float4 vb = *reinterpret_cast<const float4*>(Vt + (c0 + l15) * LDV + kb * 16 + lg * 4);
            const float* pp = Pt + (kb * 16 + lg * 4) * LDC + l15;
            const float p0 = pp[0], p1 = pp[LDC], p2 = pp[2 * LDC], p3 = pp[3 * LDC];
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(p0, vb.x, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(p1, vb.y, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(p2, vb.z, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x4f32(p3, vb.w, o, 0, 0, 0);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (r * 16 + lg * 4 + e < row_limit) Qc[(r * 16 + lg * 4 + e) * LDC + c0 + l15] = o[e] * rsum[r][e];
    }
}


}  // namespace tip

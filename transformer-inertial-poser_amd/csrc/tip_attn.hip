// tip_attn.hip — causal multi-head attention on the matrix cores for ANY configuration (general plan, training step):
// torch scaled_dot_product_attention with the additive causal mask of simple_transformer_with_state.py:56-58, as called by
// nn.TransformerEncoderLayer (:26-29,:91).  T <= 128, head width 16 / 32 / 64.
//
// One wave per (window, head), no LDS at all.  The trick that removes every re-layout: the score tile is computed
// TRANSPOSED, S^T = K Q^T (A operand = K rows, B operand = Q rows — both are plain 16-byte loads of 4 consecutive channels
// from the [M, 3D] QKV matrix).  Its accumulator layout — lane holds (keys 4*lg + r, query l15) — is exactly the A-operand
// layout of the next product P V with the 4-k-step MFMA idiom (.x/.y/.z/.w = keys 4*lg + 0..3), so P feeds P V straight
// from registers.  Softmax statistics of a query live in the 4 lanes that share l15: two xor-shuffles reduce them.
// The fused inference kernels keep their own LDS-plane variant (tip_attention.h); this one serves everything else.
#include "tip_internal.h"
#include "tip_layernorm.h"

namespace tip {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// NBMAX = ceil(T / 16) upper bound (3: T <= 48, 5: T <= 80, 8: T <= 128)
template <int DH, int NBMAX>
__global__ __launch_bounds__(256) void mattn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                        float* __restrict__ ast, int B, int T, int H, float q_scale,
                                                        AttnDrop drop) {
    constexpr int KB = DH / 16;
    const int lane = threadIdx.x & 63;
    const int bh = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (bh >= B * H) return;
    const int b = bh / H, h = bh - b * H;
    const int D = H * DH, ld = 3 * D;
    const int l15 = lane & 15, lg = lane >> 4;
    // This window's q | k | v rows and output rows as buffers of T rows: rows past the window read 0 / are dropped by the range check
    // (no branch around every load and store; the window index is wave-uniform but derived from threadIdx, hence the forced-uniform
    // pointers).  Byte offsets inside a window stay far below 2^31 (T * 3D * 4).
    const __amdgpu_buffer_rsrc_t qrs = tip_rows_buffer(qkv + (size_t)b * T * ld, T * ld * 4);
    const __amdgpu_buffer_rsrc_t ors = tip_rows_buffer(out + (size_t)b * T * D, T * D * 4);
    const int hoff = h * DH * 4;
    const int nb = (T + 15) >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int ib = 0; ib < nb; ++ib) {
        const int q = ib * 16 + l15;                 // this lane's query (as B-operand row and as statistics owner)
        f32x4 qf[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            qf[kb] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qrs, q * ld * 4 + hoff + (kb * 16 + lg * 4) * 4, 0, 0));
        }
        // ---- S^T tiles: element (key jb*16 + 4*lg + r, query q) ---------------------------------------------------------
        f32x4 st[NBMAX];
        float m = -INFINITY;
#pragma unroll
        for (int jb = 0; jb < NBMAX; ++jb) {
            st[jb] = zero4;
            if (jb <= ib) {
                const int key = jb * 16 + l15;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    const f32x4 kf = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qrs, key * ld * 4 + hoff + (D + kb * 16 + lg * 4) * 4, 0, 0));
                    st[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.x, qf[kb].x, st[jb], 0, 0, 0);
                    st[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.y, qf[kb].y, st[jb], 0, 0, 0);
                    st[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.z, qf[kb].z, st[jb], 0, 0, 0);
                    st[jb] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf.w, qf[kb].w, st[jb], 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = jb * 16 + lg * 4 + r;
                    const float sv = (kk <= q && kk < T) ? st[jb][r] * q_scale : -INFINITY;   // causal mask (:56-58)
                    st[jb][r] = sv;
                    m = fmaxf(m, sv);
                }
            }
        }
        m = lg4_max(m);
        if (q >= T) m = 0.f;                         // padded query rows: keep everything finite, nothing is stored
        float l = 0.f;
        const unsigned long long pbase = ((unsigned long long)bh * T + q) * T;
#pragma unroll
        for (int jb = 0; jb < NBMAX; ++jb) {
            if (jb <= ib) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = __expf(st[jb][r] - m);      // exp(-inf) = 0 for masked keys
                    l += p;
                    float pk = p;
                    if (drop.thresh) {
                        const int kk = jb * 16 + lg * 4 + r;
                        pk = tip_drop_hash_k(drop.key, pbase + kk) >= drop.thresh ? p * drop.scale : 0.f;
                    }
                    st[jb][r] = pk;
                }
            }
        }
        l = lg4_sum(l);
        const float inv = q < T ? 1.0f / l : 0.f;
        if (ast && lg == 0 && q < T) {
            ast[((size_t)bh * T + q) * 2] = m;
            ast[((size_t)bh * T + q) * 2 + 1] = inv;
        }
        // ---- O = P V: A = P tiles straight from the accumulators, B = V gathered as (key 4*lg + e, channel l15) ------------
        f32x4 o[KB];
#pragma unroll
        for (int cb = 0; cb < KB; ++cb) o[cb] = zero4;
#pragma unroll
        for (int jb = 0; jb < NBMAX; ++jb) {
            if (jb <= ib) {
                const int k0 = jb * 16 + lg * 4;
#pragma unroll
                for (int cb = 0; cb < KB; ++cb) {
                    const int vo = k0 * ld * 4 + hoff + (2 * D + cb * 16 + l15) * 4;
                    const float v0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(qrs, vo, 0, 0));
                    const float v1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(qrs, vo + ld * 4, 0, 0));
                    const float v2 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(qrs, vo + 2 * ld * 4, 0, 0));
                    const float v3 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(qrs, vo + 3 * ld * 4, 0, 0));
                    o[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(st[jb][0], v0, o[cb], 0, 0, 0);
                    o[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(st[jb][1], v1, o[cb], 0, 0, 0);
                    o[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(st[jb][2], v2, o[cb], 0, 0, 0);
                    o[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(st[jb][3], v3, o[cb], 0, 0, 0);
                }
            }
        }
        // O layout: lane holds (queries 4*lg + r, channel l15): fetch 1/l of those queries from the lanes that own them
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ir = __shfl(inv, lg * 4 + r, 64);
            const int qq = ib * 16 + lg * 4 + r;
#pragma unroll
            for (int cb = 0; cb < KB; ++cb)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[cb][r] * ir), ors, qq * D * 4 + hoff + (cb * 16 + l15) * 4, 0, 0);
        }
    }
}

bool mattn_supported(int dh, int T) { return (dh == 16 || dh == 32 || dh == 64) && T >= 1 && T <= 128; }

template <int DH>
static hipError_t mattn_fwd_dh(const float* qkv, float* out, float* ast, int B, int T, int H, float q_scale, AttnDrop drop,
                               hipStream_t s) {
    const dim3 grid((B * H + 3) / 4), block(256);
    if (T <= 48) hipLaunchKernelGGL((mattn_fwd_kernel<DH, 3>), grid, block, 0, s, qkv, out, ast, B, T, H, q_scale, drop);
    else if (T <= 80) hipLaunchKernelGGL((mattn_fwd_kernel<DH, 5>), grid, block, 0, s, qkv, out, ast, B, T, H, q_scale, drop);
    else hipLaunchKernelGGL((mattn_fwd_kernel<DH, 8>), grid, block, 0, s, qkv, out, ast, B, T, H, q_scale, drop);
    return hipGetLastError();
}

hipError_t launch_mattn_fwd(const float* qkv, float* out, float* ast, int B, int T, int H, int dh, float q_scale, AttnDrop drop,
                            hipStream_t s) {
    if (B <= 0) return hipSuccess;
    switch (dh) {
        case 16: return mattn_fwd_dh<16>(qkv, out, ast, B, T, H, q_scale, drop, s);
        case 32: return mattn_fwd_dh<32>(qkv, out, ast, B, T, H, q_scale, drop, s);
        case 64: return mattn_fwd_dh<64>(qkv, out, ast, B, T, H, q_scale, drop, s);
        default: return hipErrorInvalidValue;
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Backward of the same attention (training step).  Gradients that reduce over KEYS (dQ = dS K) want the probability tile
// as (key 4*lg + r, query l15) — the forward's S^T layout; gradients that reduce over QUERIES (dK = dS^T Q, dV = P^T dO)
// want it as (query 4*lg + r, key l15).  Both layouts are produced by the matrix cores (S^T = K Q^T and S = Q K^T are one
// operand swap apart), so again nothing is re-laid-out through LDS; per-query statistics (row max, 1/row sum, D = dO.O)
// live in the lanes that own the query and are fetched by shuffle for the second layout.
//   P = exp(s - m) / l (recomputed), Pd = P * keep, dPd = dO V^T, dS = P * (dPd * keep - D),
//   dQ = q_scale * dS K, dK = q_scale * dS^T Q, dV = Pd^T dO.
// One wave per (window, head).
// ---------------------------------------------------------------------------------------------------------------------
template <int DH, int NBMAX>
__global__ __launch_bounds__(64) void mattn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ o_saved,
                                                       const float* __restrict__ ast, const float* __restrict__ d_o,
                                                       float* __restrict__ dqkv, int B, int T, int H, float q_scale, AttnDrop drop) {
    constexpr int KB = DH / 16;
    const int lane = threadIdx.x & 63;
    const int bh = blockIdx.x;
    const int b = bh / H, h = bh - b * H;
    const int D = H * DH, ld = 3 * D;
    const int l15 = lane & 15, lg = lane >> 4;
    const float* base = qkv + (size_t)b * T * ld + h * DH;
    const float* gbase = d_o + (size_t)b * T * D + h * DH;
    const float* obase = o_saved + (size_t)b * T * D + h * DH;
    float* dbase = dqkv + (size_t)b * T * ld + h * DH;
    const int nb = (T + 15) >> 4;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    f32x4 dk[NBMAX][KB], dv[NBMAX][KB];
#pragma unroll
    for (int jb = 0; jb < NBMAX; ++jb)
#pragma unroll
        for (int cb = 0; cb < KB; ++cb) dk[jb][cb] = dv[jb][cb] = zero4;

    for (int ib = 0; ib < nb; ++ib) {
        const int q = ib * 16 + l15;
        // row fragments of Q_i and dO_i (lane: row l15, channels kb*16 + 4*lg ..) and this query's statistics
        f32x4 qf[KB], gf[KB];
        float dd = 0.f;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            qf[kb] = gf[kb] = zero4;
            if (q < T) {
                qf[kb] = *reinterpret_cast<const f32x4*>(base + (size_t)q * ld + kb * 16 + lg * 4);
                gf[kb] = *reinterpret_cast<const f32x4*>(gbase + (size_t)q * D + kb * 16 + lg * 4);
                const f32x4 of = *reinterpret_cast<const f32x4*>(obase + (size_t)q * D + kb * 16 + lg * 4);
                dd += (gf[kb].x * of.x + gf[kb].y * of.y) + (gf[kb].z * of.z + gf[kb].w * of.w);
            }
        }
        dd = lg4_sum(dd);
        const float mq = q < T ? ast[((size_t)bh * T + q) * 2] : 0.f;
        const float iq = q < T ? ast[((size_t)bh * T + q) * 2 + 1] : 0.f;
        float m2[4], i2[4], d2[4];      // the same statistics for queries 4*lg + r (second layout)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            m2[r] = __shfl(mq, lg * 4 + r, 64);
            i2[r] = __shfl(iq, lg * 4 + r, 64);
            d2[r] = __shfl(dd, lg * 4 + r, 64);
        }
        f32x4 dq[KB];
#pragma unroll
        for (int cb = 0; cb < KB; ++cb) dq[cb] = zero4;

#pragma unroll
        for (int jb = 0; jb < NBMAX; ++jb) {
            if (jb <= ib) {
                const int key = jb * 16 + l15;
                f32x4 kf[KB], vf[KB];
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    kf[kb] = vf[kb] = zero4;
                    if (key < T) {
                        kf[kb] = *reinterpret_cast<const f32x4*>(base + (size_t)key * ld + D + kb * 16 + lg * 4);
                        vf[kb] = *reinterpret_cast<const f32x4*>(base + (size_t)key * ld + 2 * D + kb * 16 + lg * 4);
                    }
                }
                // ---- layout 1: (key 4*lg + r, query l15) -> dQ ------------------------------------------------------------
                f32x4 s1 = zero4, p1 = zero4;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kb].x, qf[kb].x, s1, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kb].y, qf[kb].y, s1, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kb].z, qf[kb].z, s1, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[kb].w, qf[kb].w, s1, 0, 0, 0);
                    p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kb].x, gf[kb].x, p1, 0, 0, 0);
                    p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kb].y, gf[kb].y, p1, 0, 0, 0);
                    p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kb].z, gf[kb].z, p1, 0, 0, 0);
                    p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[kb].w, gf[kb].w, p1, 0, 0, 0);
                }
                f32x4 ds1;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kk = jb * 16 + lg * 4 + r;
                    float v = 0.f;
                    if (kk <= q && q < T) {
                        const float p = __expf(s1[r] * q_scale - mq) * iq;
                        float kf_ = 1.f;
                        if (drop.thresh)
                            kf_ = tip_drop_hash_k(drop.key, ((unsigned long long)bh * T + q) * T + kk) >= drop.thresh ? drop.scale : 0.f;
                        v = p * (p1[r] * kf_ - dd);
                    }
                    ds1[r] = v;
                }
                {
                    const int k0 = jb * 16 + lg * 4;
#pragma unroll
                    for (int cb = 0; cb < KB; ++cb) {
                        const float* kp = base + (size_t)k0 * ld + D + cb * 16 + l15;
                        const float k0v = k0 + 0 < T ? kp[0] : 0.f;
                        const float k1v = k0 + 1 < T ? kp[(size_t)ld] : 0.f;
                        const float k2v = k0 + 2 < T ? kp[(size_t)2 * ld] : 0.f;
                        const float k3v = k0 + 3 < T ? kp[(size_t)3 * ld] : 0.f;
                        dq[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds1[0], k0v, dq[cb], 0, 0, 0);
                        dq[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds1[1], k1v, dq[cb], 0, 0, 0);
                        dq[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds1[2], k2v, dq[cb], 0, 0, 0);
                        dq[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds1[3], k3v, dq[cb], 0, 0, 0);
                    }
                }
                // ---- layout 2: (query 4*lg + r, key l15) -> dK, dV --------------------------------------------------------
                f32x4 s2 = zero4, p2 = zero4;
#pragma unroll
                for (int kb = 0; kb < KB; ++kb) {
                    s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[kb].x, kf[kb].x, s2, 0, 0, 0);
                    s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[kb].y, kf[kb].y, s2, 0, 0, 0);
                    s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[kb].z, kf[kb].z, s2, 0, 0, 0);
                    s2 = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[kb].w, kf[kb].w, s2, 0, 0, 0);
                    p2 = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[kb].x, vf[kb].x, p2, 0, 0, 0);
                    p2 = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[kb].y, vf[kb].y, p2, 0, 0, 0);
                    p2 = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[kb].z, vf[kb].z, p2, 0, 0, 0);
                    p2 = __builtin_amdgcn_mfma_f32_16x16x4f32(gf[kb].w, vf[kb].w, p2, 0, 0, 0);
                }
                f32x4 ds2, pd2;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qq = ib * 16 + lg * 4 + r;
                    float dsv = 0.f, pdv = 0.f;
                    if (key <= qq && qq < T) {
                        const float p = __expf(s2[r] * q_scale - m2[r]) * i2[r];
                        float kf_ = 1.f;
                        if (drop.thresh)
                            kf_ = tip_drop_hash_k(drop.key, ((unsigned long long)bh * T + qq) * T + key) >= drop.thresh ? drop.scale : 0.f;
                        pdv = p * kf_;
                        dsv = p * (p2[r] * kf_ - d2[r]);
                    }
                    ds2[r] = dsv;
                    pd2[r] = pdv;
                }
                {
                    const int q0 = ib * 16 + lg * 4;
#pragma unroll
                    for (int cb = 0; cb < KB; ++cb) {
                        const float* gp = gbase + (size_t)q0 * D + cb * 16 + l15;
                        const float* qp = base + (size_t)q0 * ld + cb * 16 + l15;
                        float g4[4], q4[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            g4[e] = q0 + e < T ? gp[(size_t)e * D] : 0.f;
                            q4[e] = q0 + e < T ? qp[(size_t)e * ld] : 0.f;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            dv[jb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(pd2[e], g4[e], dv[jb][cb], 0, 0, 0);
                            dk[jb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds2[e], q4[e], dk[jb][cb], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // dQ_i: lane holds (queries 4*lg + r, channel l15)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qq = ib * 16 + lg * 4 + r;
            if (qq < T) {
#pragma unroll
                for (int cb = 0; cb < KB; ++cb) dbase[(size_t)qq * ld + cb * 16 + l15] = dq[cb][r] * q_scale;
            }
        }
    }
    // dK_j, dV_j: lane holds (keys 4*lg + r, channel l15)
#pragma unroll
    for (int jb = 0; jb < NBMAX; ++jb) {
        if (jb < nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kk = jb * 16 + lg * 4 + r;
                if (kk < T) {
#pragma unroll
                    for (int cb = 0; cb < KB; ++cb) {
                        dbase[(size_t)kk * ld + D + cb * 16 + l15] = dk[jb][cb][r] * q_scale;
                        dbase[(size_t)kk * ld + 2 * D + cb * 16 + l15] = dv[jb][cb][r];
                    }
                }
            }
        }
    }
}

template <int DH>
static hipError_t mattn_bwd_dh(const float* qkv, const float* o_saved, const float* ast, const float* d_o, float* dqkv, int B, int T,
                               int H, float q_scale, AttnDrop drop, hipStream_t s) {
    const dim3 grid(B * H), block(64);
    if (T <= 48) hipLaunchKernelGGL((mattn_bwd_kernel<DH, 3>), grid, block, 0, s, qkv, o_saved, ast, d_o, dqkv, B, T, H, q_scale, drop);
    else if (T <= 80) hipLaunchKernelGGL((mattn_bwd_kernel<DH, 5>), grid, block, 0, s, qkv, o_saved, ast, d_o, dqkv, B, T, H, q_scale, drop);
    else hipLaunchKernelGGL((mattn_bwd_kernel<DH, 8>), grid, block, 0, s, qkv, o_saved, ast, d_o, dqkv, B, T, H, q_scale, drop);
    return hipGetLastError();
}

hipError_t launch_mattn_bwd(const float* qkv, const float* o_saved, const float* ast, const float* d_o, float* dqkv, int B, int T,
                            int H, int dh, float q_scale, AttnDrop drop, hipStream_t s) {
    if (B <= 0) return hipSuccess;
    switch (dh) {
        case 16: return mattn_bwd_dh<16>(qkv, o_saved, ast, d_o, dqkv, B, T, H, q_scale, drop, s);
        case 32: return mattn_bwd_dh<32>(qkv, o_saved, ast, d_o, dqkv, B, T, H, q_scale, drop, s);
        case 64: return mattn_bwd_dh<64>(qkv, o_saved, ast, d_o, dqkv, B, T, H, q_scale, drop, s);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace tip

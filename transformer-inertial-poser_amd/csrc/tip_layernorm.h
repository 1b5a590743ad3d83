// tip_layernorm.h — LayerNorm over the rows of an LDS-resident activation tile, shared by the fused encoder kernels (so that the
// one-window, two-window and hybrid plans normalise every row with the same instructions: their bit-identity tests rely on it).
#pragma once
#include <hip/hip_runtime.h>

namespace tip {

// v + (v of another lane of the same 16-lane row), by a DPP modifier on the add's operand: no LDS crossbar trip.
// hipcc lowers every __shfl_xor to ds_bpermute_b32 (an LDS round trip, ~100 cycles of latency each); a 64-lane row reduction
// was 6 of them per statistic, 72 per wave and LayerNorm — 3.4 us per LayerNorm, 5 % of the encoder.
template <int CTRL>
__device__ __forceinline__ float dpp_peer(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// sum over the 16 lanes of a DPP row; every lane ends with the same bits (each step adds a value to its mirror image)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_peer<0xB1>(v);    // quad_perm [1,0,3,2]: lane ^ 1
    v += dpp_peer<0x4E>(v);    // quad_perm [2,3,0,1]: lane ^ 2
    v += dpp_peer<0x141>(v);   // row_half_mirror: the other quad of this half
    v += dpp_peer<0x140>(v);   // row_mirror: the other half of the row
    return v;
}

// Cross-ROW exchanges (lane ^ 16, lane ^ 32) without the LDS crossbar: gfx950's v_permlane16_swap_b32 / v_permlane32_swap_b32.
//   swap16(a, b): a' = [a.row0, b.row0, a.row2, b.row2], b' = [a.row1, b.row1, a.row3, b.row3]      (rows of 16 lanes)
//   swap32(a, b): a' = [a.lanes 0-31, b.lanes 0-31],     b' = [a.lanes 32-63, b.lanes 32-63]
// (semantics measured: tools/probes/permlane_probe.hip).  Through inline asm: hipcc's __builtin_amdgcn_permlane{16,32}_swap hands
// back vdst' in BOTH elements of its result (ROCm 7.2).  The s_nops cover the VALU-write -> permlane-read wait states that the
// compiler inserts around its own permlane instructions but not inside an asm statement.
__device__ __forceinline__ void swap16(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap32(float& a, float& b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
// v (op) v-of-lane^16, then (op) lane^32: the four lanes that share lane & 15 all end with the combination of their four
// values — same pairing and (commutative) operand order as x = op(x, shfl_xor(x, 16)); x = op(x, shfl_xor(x, 32)), same bits.
__device__ __forceinline__ float lg4_sum(float v) {
    float w = v;
    swap16(v, w);
    v += w;
    w = v;
    swap32(v, w);
    return v + w;
}
__device__ __forceinline__ float lg4_max(float v) {
    float w = v;
    swap16(v, w);
    v = fmaxf(v, w);
    w = v;
    swap32(v, w);
    return fmaxf(v, w);
}

// sum over all 64 lanes, every lane gets it: four DPP adds inside each 16-lane row, then the four rows by two permlane swaps —
// six VALU instructions where the __shfl_xor butterfly is six LDS round trips
__device__ __forceinline__ float wave64_sum(float v) { return lg4_sum(row16_sum(v)); }

// LayerNorm (eps 1e-5, biased variance, affine) over rows 0 .. ROWS-1 of X [rows][LD] (256 columns).  SIXTEEN lanes per row —
// a wave normalises four rows at a time, lane (sub = lane >> 4, q = lane & 15) holding columns 4 (q + 16 j) .. + 3, j = 0..3 of
// row 32 p + 4 wave + sub — so both row statistics are a 16-value local sum plus row16_sum: four DPP adds instead of six LDS
// round trips.  SAVE (training forward): the pre-norm rows, (mean, rstd) and the normalised rows also go to HBM for rows < T.
typedef float ln_f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned ln_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned ln_u32x2 __attribute__((ext_vector_type(2)));

// A wave-uniform pointer forced into SGPRs (a buffer descriptor built from VGPRs turns every buffer instruction into a readfirstlane
// "waterfall" loop), and the first `bytes` bytes behind it as a raw buffer: loads past the extent return 0, stores past it are
// dropped — the pad rows of a window are masked by the hardware's range check instead of by a branch around every access.
template <typename P>
__device__ __forceinline__ P* tip_uniform_ptr(P* p) {
    const unsigned long long v = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<P*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tip_rows_buffer(const float* p, int bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tip_uniform_ptr(p)), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
constexpr int kTipNT = 2;   // buffer-store cache policy: non-temporal (streaming stash: read next by the backward, not by this kernel)

template <int ROWS, int LD, bool SAVE = false>
__device__ __forceinline__ void layernorm_rows16(float* X, const float* __restrict__ g, const float* __restrict__ be, int wave,
                                                 int lane, float* zs = nullptr, float* sts = nullptr, float* xs = nullptr, int T = 0) {
    constexpr int DCOLS = 256, NPASS = (ROWS + 31) / 32;
    const int q = lane & 15, sub = lane >> 4;
    // SAVE: the three stash arrays as buffers of T rows (rows >= T are dropped by the range check)
    const __amdgpu_buffer_rsrc_t zrs = tip_rows_buffer(SAVE ? zs : X, SAVE ? T * DCOLS * 4 : 0);
    const __amdgpu_buffer_rsrc_t xrs = tip_rows_buffer(SAVE ? xs : X, SAVE ? T * DCOLS * 4 : 0);
    const __amdgpu_buffer_rsrc_t srs = tip_rows_buffer(SAVE ? sts : X, SAVE ? T * 8 : 0);
    float4 gg[4], bb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        gg[j] = *reinterpret_cast<const float4*>(g + (q + 16 * j) * 4);
        bb[j] = *reinterpret_cast<const float4*>(be + (q + 16 * j) * 4);
    }
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        if (p * 32 + wave * 4 < ROWS) {                     // wave-uniform (ROWS is a multiple of 4 x the waves that take part)
            const int row = p * 32 + wave * 4 + sub;
            float* xr = X + row * LD + q * 4;
            float4 v[4];
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = *reinterpret_cast<const float4*>(xr + j * 64);
                if (SAVE) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ln_u32x4, v[j]), zrs, (row * DCOLS + (q + 16 * j) * 4) * 4, 0, kTipNT);
            }
            s = ((v[0].x + v[0].y) + (v[0].z + v[0].w)) + ((v[1].x + v[1].y) + (v[1].z + v[1].w));
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
                float4 o;
                o.x = v[j].x * rstd * gg[j].x + bb[j].x;
                o.y = v[j].y * rstd * gg[j].y + bb[j].y;
                o.z = v[j].z * rstd * gg[j].z + bb[j].z;
                o.w = v[j].w * rstd * gg[j].w + bb[j].w;
                *reinterpret_cast<float4*>(xr + j * 64) = o;
                if (SAVE) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ln_u32x4, o), xrs, (row * DCOLS + (q + 16 * j) * 4) * 4, 0, kTipNT);
            }
            if (SAVE) {   // one 8-byte store per row from lane q == 0 (the other lanes aim past the extent)
                const ln_u32x2 mr = {__float_as_uint(mean), __float_as_uint(rstd)};
                __builtin_amdgcn_raw_buffer_store_b64(mr, srs, q == 0 ? row * 8 : T * 8, 0, 0);
            }
        }
    }
}

}  // namespace tip

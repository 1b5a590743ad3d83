// tip_head.hip — output projection y = h W_out^T + b (/root/reference/simple_transformer_with_state.py:102) with the weight
// RESIDENT IN REGISTERS.
//
// N = 131 columns (9 blocks of 16), K = rnn_hidden = 512.  head_gemm_kernel (tip_general.hip) gives every wave 16 rows x 3
// column blocks and lets it stream its 98 KB of weight fragments from L2: 1 920 waves at B = 256 pull 190 MB through the
// CUs' vector L1s for a 1.4-GFLOP product, three passes over HALL (grid.y = 3) — 19.7 us, the matrix pipe 34 % busy.
//
// Here a workgroup is 8 waves that split K: wave w keeps k-blocks 4 w .. 4 w + 3 of ALL nine column blocks — 36 fragments,
// 144 VGPRs, loaded once — and the workgroup walks over row groups, reading every row of HALL exactly once (each wave its own
// 64-float slice of the row, straight into MFMA A fragments: no LDS staging of the operand).  The eight k-partials of a tile
// meet in LDS: every wave writes its accumulators in fragment order (conflict-free 16-byte writes), one barrier, and wave w
// sums column block w of all eight partials in a fixed order, adds the bias and stores.  The partial buffers alternate, so the
// only synchronisation is that one barrier per tile, and the reduction of tile i runs under the MFMAs of tile i + 1.
//
// Row tiling.  A window is T = 40 rows = two 16-row tiles (v_mfma_f32_16x16x4_f32) + 8 rows, and 256 windows on 256 CUs leave
// no room for a third padded tile per CU: rows 32-39 go through v_mfma_f32_4x4x1_16b_f32 fed by THE SAME weight registers (the
// hybrid row tiling of fused_encoder_h_kernel, tip_fused.hip): lane (l15, lg) of a 16x16x4 B fragment is column l15 at
// k = 16 kb + 4 lg + s, i.e. block (lg, l15 >> 2) / column l15 & 3 of a 4x4x1 whose A operand is h[row lane & 3] at the same k;
// the accumulator holds one k-partial per lg, combined by three permlane swaps.  Per window and SIMD: 2 x 9 216 + 2 x 2 592
// MFMA issue cycles instead of 3 x 9 216.
//   MODE 0  groups of 40 rows: tiles B(0) B(16) T(32) T(36)    — full output, T a multiple of 40 (a group never straddles windows)
//   MODE 2  groups of 4 rows:  tile T(0)                       — last-row output of the same windows
// A row's summation order depends on the tile kind only (B: per wave one MFMA chain over its 16 k-steps; T: four chains, one
// per lg, combined (p0 + p2) + (p1 + p3); then the eight waves ((w0 + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7))).  In MODE 0 row
// t of a window is a T row iff t mod 40 >= 32, so row T - 1 always is: MODE 2 reproduces it bit for bit (forward_last ==
// forward[:, -1], tested).  Other window lengths keep head_gemm_kernel in both forms.
#include "tip_head.h"

namespace tip {

__device__ unsigned long long g_hd_trace[32];
__device__ unsigned long long g_hd_wg[2 * 1024];   // [wg][entry, exit] in s_memrealtime ticks (100 MHz, device-wide counter)

template <int MODE, bool TRACE = false>
__global__ __launch_bounds__(hd::THREADS) void head_ksplit_kernel(const float* __restrict__ A, unsigned lda_b, unsigned a_bytes,
                                                                  const float* __restrict__ wfrag, const float* __restrict__ bias,
                                                                  float* __restrict__ Y, int ldy, int M, int N, int ngroups) {
    head_ksplit_body<MODE, TRACE, 0>(A, lda_b, a_bytes, wfrag, bias, Y, ldy, M, N, ngroups, HdGridStride{});
}

extern "C" int tip_debug_read_head_wg(unsigned long long* out, int n) {
    if (!out || n < 0 || n > 2048) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hd_wg), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -5;
}

extern "C" int tip_debug_read_head_trace(unsigned long long* out, int n) {
    if (!out || n < 0 || n > 32) return -1;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_hd_trace), sizeof(unsigned long long) * n) == hipSuccess ? 0 : -5;
}

// Y[M, N] = A[M, 512] W^T + b on the register-resident kernel above.  rows_per_window: T (a multiple of 40) for the full output,
// ignored for last_only (M = one row per window).  Returns hipErrorInvalidValue for shapes it does not serve (the caller then
// keeps head_gemm_kernel).
hipError_t launch_head_ksplit(const float* A, long long lda, const float* wfrag, const float* bias, float* Y, int ldy, int M, int N,
                              int K, bool last_only, int num_cus, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    if (K != 512 || N <= 128 || N > 144) return hipErrorInvalidValue;
    if (lda * 4 > 0x7fffffffLL || ((long long)(M - 1) * lda + K) * 4 > 0xffffffffLL || (long long)M * ldy * 4 > 0x7fffffffLL)
        return hipErrorInvalidValue;
    const unsigned a_bytes = (unsigned)(((long long)(M - 1) * lda + K) * 4);
    hipError_t e = hipSuccess;
    static PerDeviceFlag attr_flag[2];
    auto prep = [&](const void* fn, int slot) -> hipError_t {
        bool& done = attr_flag[slot].cur();
        if (done) return hipSuccess;
        hipError_t e2 = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, hd::LDS_BYTES);
        if (e2 == hipSuccess) done = true;
        return e2;
    };
    static const bool trace = tip_env("TIP_HEAD_TRACE") && tip_env("TIP_HEAD_TRACE")[0] == '1';
    if (trace && !last_only && M % 40 == 0) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(head_ksplit_kernel<0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, hd::LDS_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((head_ksplit_kernel<0, true>), dim3(std::min(M / 40, num_cus)), dim3(hd::THREADS), hd::LDS_BYTES, s, A,
                           (unsigned)(lda * 4), a_bytes, wfrag, bias, Y, ldy, M, N, M / 40);
        return hipGetLastError();
    }
    if (!last_only) {
        if (M % 40) return hipErrorInvalidValue;
        const int ngroups = M / 40;
        e = prep(reinterpret_cast<const void*>(head_ksplit_kernel<0>), 0);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(head_ksplit_kernel<0>, dim3(std::min(ngroups, num_cus)), dim3(hd::THREADS), hd::LDS_BYTES, s, A,
                           (unsigned)(lda * 4), a_bytes, wfrag, bias, Y, ldy, M, N, ngroups);
    } else {
        const int ngroups = (M + 3) / 4;
        e = prep(reinterpret_cast<const void*>(head_ksplit_kernel<2>), 1);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(head_ksplit_kernel<2>, dim3(std::min(ngroups, num_cus)), dim3(hd::THREADS), hd::LDS_BYTES, s, A,
                           (unsigned)(lda * 4), a_bytes, wfrag, bias, Y, ldy, M, N, ngroups);
    }
    return hipGetLastError();
}

}  // namespace tip

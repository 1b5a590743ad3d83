// EXPLORATORY probe (never the headline, never the product path): the feed-forward block of one encoder layer
// (/root/reference/simple_transformer_with_state.py:26-29: linear1 256 -> 1024, ReLU, linear2 1024 -> 256) with fp32 operands
// EMULATED on the fp16 matrix cores, which run 16x the fp32-input MFMA rate on gfx950 (MI355X_MICROARCH.md):
//     x = xh + xl * 2^-11,   xh = fp16(x),   xl = fp16((x - xh) * 2^11)        (both operands, weights split at pack time)
//     x * w ~= xh wh + (xh wl + xl wh) * 2^-11                                   (xl wl * 2^-22 dropped)
// three v_mfma_f32_16x16x32_f16 passes per product, fp32 accumulation (fp16 x fp16 products are exact in fp32), two accumulators
// (the hh sum and the cross sum, combined once in the epilogue).  Questions it answers: (i) what does the phase cost against the
// fp32-MFMA phase of fused_encoder_h_kernel (75 us per layer at B = 256) and against its own MFMA-issue and weight-stream floors,
// (ii) what is the error against fp64, relative to the error of plain fp32 arithmetic, as the operands leave the benign
// random-init regime (gain sweep as in tests/golden/make_golden.py --cond).
// One window (40 rows, padded to 48) per 512-thread workgroup, activations hi/lo in LDS, weights hi/lo streamed from L2 in
// fragment order — the structure of the fused encoder's FFN phase.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/ffn_split16_probe.out tools/probes/ffn_split16_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int T = 40, ROWS = 48, D = 256, F = 1024, LDA = D + 16;   // LDA in halfs: 544-byte rows, conflict-free 16-byte reads
constexpr int THREADS = 512;
constexpr float SC = 2048.f, ISC = 1.f / 2048.f;
constexpr int LDS_BYTES = 4 * ROWS * LDA * 2;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void split(float x, _Float16& h, _Float16& l) {
    h = (_Float16)x;
    l = (_Float16)((x - (float)h) * SC);
}

// W?f: fragment order [col block n][k block kb][hi, lo][64 lanes][8 halfs]; lane (l15, lg) = W[16 n + l15][32 kb + 8 lg + i]
__global__ __launch_bounds__(THREADS) void ffn_split16_kernel(const float* __restrict__ X, const h8* __restrict__ W1f,
                                                              const float* __restrict__ b1, const h8* __restrict__ W2f,
                                                              const float* __restrict__ b2, float* __restrict__ Y) {
    extern __shared__ _Float16 lds[];
    _Float16* Xh = lds;
    _Float16* Xl = Xh + ROWS * LDA;
    _Float16* Hh = Xl + ROWS * LDA;
    _Float16* Hl = Hh + ROWS * LDA;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, lg = lane >> 4;
    const float* x = X + (size_t)blockIdx.x * T * D;
    for (int i = tid; i < ROWS * D; i += THREADS) {
        const int r = i / D, c = i - r * D;
        _Float16 h = (_Float16)0.f, l = (_Float16)0.f;
        if (r < T) split(x[i], h, l);
        Xh[r * LDA + c] = h;
        Xl[r * LDA + c] = l;
    }
    __syncthreads();
    f4 o_h[3][2], o_x[3][2];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 2; ++c) o_h[r][c] = o_x[r][c] = (f4){0.f, 0.f, 0.f, 0.f};
    const int aoff = l15 * LDA + lg * 8;   // this lane's A-fragment offset inside a 16-row block (halfs)
#pragma unroll 1
    for (int f = 0; f < 4; ++f) {
        // ---- linear1, hidden columns 256 f .. + 255: wave w owns column blocks 16 f + 2 w, + 1 ----
        {
            f4 a_h[3][2], a_x[3][2];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) a_h[r][c] = a_x[r][c] = (f4){0.f, 0.f, 0.f, 0.f};
            const h8* wp = W1f + ((size_t)(f * 16 + wave * 2) * (D / 32)) * 2 * 64 + lane;
#pragma unroll
            for (int kb = 0; kb < D / 32; ++kb) {
                h8 bh[2], bl[2], ah[3], al[3];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    bh[c] = wp[((size_t)c * (D / 32) + kb) * 2 * 64];
                    bl[c] = wp[((size_t)c * (D / 32) + kb) * 2 * 64 + 64];
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    ah[r] = *reinterpret_cast<const h8*>(Xh + r * 16 * LDA + aoff + kb * 32);
                    al[r] = *reinterpret_cast<const h8*>(Xl + r * 16 * LDA + aoff + kb * 32);
                }
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        a_h[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r], bh[c], a_h[r][c], 0, 0, 0);
                        a_x[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r], bl[c], a_x[r][c], 0, 0, 0);
                        a_x[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[r], bh[c], a_x[r][c], 0, 0, 0);
                    }
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int col = (wave * 2 + c) * 16 + l15;
                const float bv = b1[f * 256 + col];
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = fmaxf(a_h[r][c][e] + a_x[r][c][e] * ISC + bv, 0.f);
                        _Float16 h, l;
                        split(v, h, l);
                        Hh[(r * 16 + lg * 4 + e) * LDA + col] = h;
                        Hl[(r * 16 + lg * 4 + e) * LDA + col] = l;
                    }
            }
        }
        __syncthreads();
        // ---- linear2 partial over this hidden chunk: wave w owns output column blocks 2 w, + 1 ----
        {
            const h8* wp = W2f + ((size_t)(wave * 2) * (F / 32) + f * 8) * 2 * 64 + lane;
#pragma unroll
            for (int kb = 0; kb < 8; ++kb) {
                h8 bh[2], bl[2], ah[3], al[3];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    bh[c] = wp[((size_t)c * (F / 32) + kb) * 2 * 64];
                    bl[c] = wp[((size_t)c * (F / 32) + kb) * 2 * 64 + 64];
                }
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    ah[r] = *reinterpret_cast<const h8*>(Hh + r * 16 * LDA + aoff + kb * 32);
                    al[r] = *reinterpret_cast<const h8*>(Hl + r * 16 * LDA + aoff + kb * 32);
                }
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        o_h[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r], bh[c], o_h[r][c], 0, 0, 0);
                        o_x[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[r], bl[c], o_x[r][c], 0, 0, 0);
                        o_x[r][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[r], bh[c], o_x[r][c], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    }
    float* y = Y + (size_t)blockIdx.x * T * D;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int col = (wave * 2 + c) * 16 + l15;
        const float bv = b2[col];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int row = r * 16 + lg * 4 + e;
                if (row < T) y[row * D + col] = o_h[r][c][e] + o_x[r][c][e] * ISC + bv;
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double urand() {   // splitmix64 -> [0, 1)
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
static double nrand() { return std::sqrt(-2.0 * std::log(1.0 - urand())) * std::cos(6.283185307179586 * urand()); }

static void pack_frag(const std::vector<float>& W, int N, int K, std::vector<_Float16>& out) {
    out.assign((size_t)N * K * 2, (_Float16)0.f);
    const int KB = K / 32;
    for (int n = 0; n < N / 16; ++n)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int i = 0; i < 8; ++i) {
                    const int l15 = lane & 15, lg = lane >> 4;
                    const float w = W[(size_t)(n * 16 + l15) * K + kb * 32 + lg * 8 + i];
                    const _Float16 h = (_Float16)w;
                    const _Float16 l = (_Float16)((w - (float)h) * SC);
                    const size_t base = (((size_t)n * KB + kb) * 2) * 64 * 8;
                    out[base + (size_t)lane * 8 + i] = h;
                    out[base + 64 * 8 + (size_t)lane * 8 + i] = l;
                }
}

int main() {
    const int B = 256, NREF = 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(ffn_split16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    float *dX, *db1, *db2, *dY;
    h8 *dW1, *dW2;
    CK(hipMalloc(&dX, (size_t)B * T * D * 4));
    CK(hipMalloc(&dY, (size_t)B * T * D * 4));
    CK(hipMalloc(&db1, F * 4));
    CK(hipMalloc(&db2, D * 4));
    CK(hipMalloc(&dW1, (size_t)F * D * 4));
    CK(hipMalloc(&dW2, (size_t)F * D * 4));
    printf("feed-forward block, 40 x 256 -> 1024 -> 256 per window, %d windows per launch, split-fp16 (3 f16 MFMA passes per product)\n", B);
    printf("%-22s %12s %12s %12s %8s\n", "operands", "max|y|", "fp32 err", "split16 err", "ratio");
    for (int gi = 0; gi < 5; ++gi) {
        const double gain = (double[]){1.0, 2.0, 4.0, 8.0, 16.0}[gi];
        // X: LayerNorm output (unit variance, gamma up to x gain), weights U(+-gain / sqrt(fan_in))
        std::vector<float> X((size_t)B * T * D), W1((size_t)F * D), W2((size_t)D * F), b1(F), b2(D);
        for (auto& v : X) v = (float)(nrand() * std::min(gain, 3.0));
        for (auto& v : W1) v = (float)((urand() * 2 - 1) * gain / 16.0);
        for (auto& v : W2) v = (float)((urand() * 2 - 1) * gain / 32.0);
        for (auto& v : b1) v = (float)((urand() * 2 - 1) / 16.0);
        for (auto& v : b2) v = (float)((urand() * 2 - 1) / 32.0);
        std::vector<_Float16> W1f, W2f;
        pack_frag(W1, F, D, W1f);
        pack_frag(W2, D, F, W2f);
        CK(hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db1, b1.data(), F * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db2, b2.data(), D * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW1, W1f.data(), W1f.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dW2, W2f.data(), W2f.size() * 2, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(ffn_split16_kernel, dim3(B), dim3(THREADS), LDS_BYTES, 0, dX, dW1, db1, dW2, db2, dY);
        CK(hipDeviceSynchronize());
        std::vector<float> Y((size_t)NREF * T * D);
        CK(hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost));
        double e32 = 0, e16 = 0, ymax = 0;
        std::vector<double> h64(F);
        std::vector<float> h32(F);
        for (int w = 0; w < NREF; ++w)
            for (int t = 0; t < T; ++t) {
                const float* xr = &X[((size_t)w * T + t) * D];
                for (int j = 0; j < F; ++j) {
                    double s = b1[j];
                    float s32 = 0.f;
                    for (int k = 0; k < D; ++k) {
                        s += (double)xr[k] * (double)W1[(size_t)j * D + k];
                        s32 = fmaf(xr[k], W1[(size_t)j * D + k], s32);
                    }
                    h64[j] = s > 0 ? s : 0;
                    s32 += b1[j];
                    h32[j] = s32 > 0 ? s32 : 0.f;
                }
                for (int c = 0; c < D; ++c) {
                    double s = b2[c];
                    float s32 = 0.f;
                    for (int j = 0; j < F; ++j) {
                        s += h64[j] * (double)W2[(size_t)c * F + j];
                        s32 = fmaf(h32[j], W2[(size_t)c * F + j], s32);
                    }
                    s32 += b2[c];
                    ymax = std::max(ymax, std::fabs(s));
                    e32 = std::max(e32, std::fabs((double)s32 - s));
                    e16 = std::max(e16, std::fabs((double)Y[((size_t)w * T + t) * D + c] - s));
                }
            }
        char name[64];
        snprintf(name, sizeof name, "gain %g", gain);
        printf("%-22s %12.3f %12.3e %12.3e %8.2f\n", name, ymax, e32, e16, e16 / e32);
    }
    // timing (operands of the last sweep point)
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(ffn_split16_kernel, dim3(B), dim3(THREADS), LDS_BYTES, 0, dX, dW1, db1, dW2, db2, dY);
    CK(hipDeviceSynchronize());
    const int REP = 200;
    CK(hipEventRecord(e0));
    for (int i = 0; i < REP; ++i) hipLaunchKernelGGL(ffn_split16_kernel, dim3(B), dim3(THREADS), LDS_BYTES, 0, dX, dW1, db1, dW2, db2, dY);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / REP;
    const double flop = 2.0 * 2.0 * T * D * F * B;          // algorithmic (fp32-equivalent) FLOPs per launch
    printf("time per launch (one layer's feed-forward block for %d windows, input staging and output store included): %.1f us\n", B, us);
    printf("  = %.1f TFLOP/s fp32-equivalent (fp32-MFMA peak 157.3); the fp32-MFMA phase of fused_encoder_h_kernel: ~75 us per layer\n",
           flop / us / 1e6);
    printf("  floors per window: MFMA issue 3 passes x 48 rows: %.1f us; weight stream 2 MB (hi + lo) through one CU's vector L1 at 64 B/clk: %.1f us\n",
           3.0 * 2.0 * 2.0 * ROWS * D * F / (4.0 * 16384.0 / 17.0) / 2400.0, 2.0 * F * D * 4 / 64.0 / 2400.0);
    return 0;
}

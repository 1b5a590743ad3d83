// Probe: v_mfma_f32_4x4x1_16b_f32 on gfx950 — operand/result lane maps with A-block broadcast (cbsz/abid) and issue rate.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma4x4_probe.out mfma4x4_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void semantics(const float* a, const float* b, float* d0, float* d1, float* d2) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    f32x4 r0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);   // 16 independent blocks
    f32x4 r1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 4, 3, 0);   // A block 3 broadcast to all 16 blocks
    f32x4 r2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 3, 2, 0);   // two groups of 8: A blocks 2 and 10
    for (int i = 0; i < 4; ++i) { d0[l * 4 + i] = r0[i]; d1[l * 4 + i] = r1[i]; d2[l * 4 + i] = r2[i]; }
}

template <int MODE>
__global__ __launch_bounds__(512) void rate(float* out, int iters, float seed) {
    const int l = threadIdx.x & 63;
    float a0 = seed + l, a1 = seed * 2 + l, b0 = seed - l, b1 = seed + 0.5f * l;
    f32x4 acc[10];
    for (int i = 0; i < 10; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {   // 4x4x1, A broadcast, 10 row blocks per k, 4 k per trip
#define K1(A, B) \
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(A, B, acc[0], 4, 0, 0); \
            acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(A, B, acc[1], 4, 1, 0); \
            acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(A, B, acc[2], 4, 2, 0); \
            acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(A, B, acc[3], 4, 3, 0); \
            acc[4] = __builtin_amdgcn_mfma_f32_4x4x1f32(A, B, acc[4], 4, 4, 0); \
            acc[5] = __builtin_amdgcn_mfma_f32_4x4x1f32(A, B, acc[5], 4, 5, 0); \
            acc[6] = __builtin_amdgcn_mfma_f32_4x4x1f32(A, B, acc[6], 4, 6, 0); \
            acc[7] = __builtin_amdgcn_mfma_f32_4x4x1f32(A, B, acc[7], 4, 7, 0); \
            acc[8] = __builtin_amdgcn_mfma_f32_4x4x1f32(A, B, acc[8], 4, 8, 0); \
            acc[9] = __builtin_amdgcn_mfma_f32_4x4x1f32(A, B, acc[9], 4, 9, 0);
            K1(a0, b0) K1(a1, b1) K1(a0, b1) K1(a1, b0)
#undef K1
        } else if (MODE == 1) {   // 4x4x1, two groups of 8 blocks (cbsz = 3), 5 MFMAs per k, 8 k per trip
#define K2(A, B) \
            acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(A, B, acc[0], 3, 0, 0); \
            acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(A, B, acc[1], 3, 1, 0); \
            acc[2] = __builtin_amdgcn_mfma_f32_4x4x1f32(A, B, acc[2], 3, 2, 0); \
            acc[3] = __builtin_amdgcn_mfma_f32_4x4x1f32(A, B, acc[3], 3, 3, 0); \
            acc[4] = __builtin_amdgcn_mfma_f32_4x4x1f32(A, B, acc[4], 3, 4, 0);
            K2(a0, b0) K2(a1, b1) K2(a0, b1) K2(a1, b0) K2(a0, b0) K2(a1, b1) K2(a0, b1) K2(a1, b0)
#undef K2
        } else {   // reference: 16x16x4, 10 accumulators, 10 MFMAs per trip x 4
#define K3(A, B) \
            for (int i = 0; i < 10; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(A, B, acc[i], 0, 0, 0);
            K3(a0, b0)
#undef K3
        }
    }
    float s = 0.f;
    for (int i = 0; i < 10; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    float ha[64], hb[64], hd[3][256];
    for (int l = 0; l < 64; ++l) { ha[l] = 1 + l; hb[l] = 100 * (1 + l); }
    float *a, *b, *d0, *d1, *d2;
    hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d0, 1024); hipMalloc(&d1, 1024); hipMalloc(&d2, 1024);
    hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, a, b, d0, d1, d2);
    hipMemcpy(hd[0], d0, 1024, hipMemcpyDeviceToHost); hipMemcpy(hd[1], d1, 1024, hipMemcpyDeviceToHost);
    hipMemcpy(hd[2], d2, 1024, hipMemcpyDeviceToHost);
    // expected if D reg r of lane l = A_blk[r] * B[l]: value = (1 + 4*blkA + r) * 100 * (1 + l)
    const char* names[3] = {"cbsz0", "cbsz4 abid3", "cbsz3 abid2"};
    for (int m = 0; m < 3; ++m) {
        int ok_rowreg = 1;
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                int blk = m == 0 ? l / 4 : (m == 1 ? 3 : (l / 32) * 8 + 2);
                float want = (1 + 4 * blk + r) * 100.f * (1 + l);
                if (hd[m][l * 4 + r] != want) ok_rowreg = 0;
            }
        printf("%s: D[lane l][reg r] == A[4*blk + r] * B[l] : %s   (lane5: %g %g %g %g; lane37: %g %g %g %g)\n", names[m],
               ok_rowreg ? "YES" : "no", hd[m][20], hd[m][21], hd[m][22], hd[m][23], hd[m][148], hd[m][149], hd[m][150], hd[m][151]);
    }
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int mode = 0; mode < 3; ++mode)
        for (int threads = 256; threads <= 512; threads += 256) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
                else if (mode == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
                else hipLaunchKernelGGL(rate<2>, dim3(256), dim3(threads), 0, 0, out, iters, 1.0f);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep == 1) {
                    const double mf = mode == 2 ? 10.0 : 40.0;
                    const double fl = mode == 2 ? 2048.0 : 512.0;
                    const double waves = 256.0 * threads / 64;
                    const double tf = waves * iters * mf * fl / (ms * 1e-3) / 1e12;
                    const double cyc = ms * 1e-3 * 2.4e9 / (iters * mf) / (threads / 256.0);
                    printf("mode %d (%s) %d threads/WG: %.3f ms, %.1f TFLOP/s, %.1f cycles @2.4GHz per MFMA per SIMD\n", mode,
                           mode == 0 ? "4x4x1 cbsz4" : mode == 1 ? "4x4x1 cbsz3" : "16x16x4", threads, ms, tf, cyc);
                }
            }
        }
    return 0;
}

// Probe: do fp32 VALU FMAs (v_pk_fma_f32 / v_fma_f32) issue under fp32 MFMAs on gfx950, and at what cost to the MFMA stream?
// A 512-thread workgroup per CU (two waves per SIMD, as the fused encoder); per trip every wave issues 8 v_mfma_f32_16x16x4_f32
// (eight independent accumulators: what one B fragment costs for rows 0-31) and NV vector FMAs on registers of their own,
// spread evenly between the MFMAs.  Reported: cycles per trip (s_memtime of wave 0 of workgroup 0, 100 MHz counter scaled by the
// event-timed duration) and the whole launch's time.
// Build: hipcc --offload-arch=gfx950 -O3 -o coissue_probe.out coissue_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MFMA(ACC) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))
#define MFMA4(ACC) asm volatile("v_mfma_f32_4x4x1_16b_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))
#define PK(ACC, X, W) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(X), "v"(W))
#define PKB(ACC, X, W) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(ACC) : "v"(X), "v"(W))
#define FMA1(ACC, X, W) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(X), "v"(W))
#define XOR1(ACC, X) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(ACC) : "v"(X))
#define EXP1(ACC, X) asm volatile("v_exp_f32 %0, %1" : "=v"(ACC) : "v"(X))
#define MULLO(ACC, X) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(ACC) : "v"(X))
#define ADD1(ACC, X) asm volatile("v_add_f32 %0, %1, %0" : "+v"(ACC) : "v"(X))

// MODE: 0 = MFMAs only; 1 = vector FMAs only; 2 = both; 3 = 8 x 16x16x4 + 8 x 4x4x1 (today's hybrid fragment); NVPER = vector
// FMAs behind each of the 8 MFMAs; KIND: 0 = v_pk_fma_f32, 1 = v_pk_fma_f32 with a broadcast operand (op_sel_hi), 2 = v_fma_f32, 3 = v_xor_b32, 4 = v_exp_f32,
// 5 = v_mul_lo_u32, 6 = v_add_f32 (the TFLOP/s column counts every vector instruction as 64 x 2 FLOP)
template <int MODE, int NVPER, int KIND>
__global__ __launch_bounds__(512) void rate(float* out, int iters, float seed) {
    const int l = threadIdx.x & 63;
    float a = seed + l, b = seed - l;
    f32x4 acc[8], acc4[8];
    f32x2 vacc[8], x[4], w[2];
    for (int i = 0; i < 8; ++i) acc[i] = acc4[i] = (f32x4){0.f, 0.f, 0.f, 0.f}, vacc[i] = (f32x2){0.f, 0.f};
    for (int i = 0; i < 4; ++i) x[i] = (f32x2){seed * i, seed + i};
    w[0] = (f32x2){seed, 1.f}, w[1] = (f32x2){0.5f, seed};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (MODE != 1) MFMA(acc[m]);
            if (MODE == 3) MFMA4(acc4[m]);
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int v = 0; v < NVPER; ++v) {
                    const int j = (m * NVPER + v) & 7;
                    if (KIND == 0) PK(vacc[j], x[j & 3], w[j & 1]);
                    else if (KIND == 1) PKB(vacc[j], x[j & 3], w[j & 1]);
                    else if (KIND == 2) { FMA1(vacc[j].x, x[j & 3].x, w[j & 1].x); }
                    else if (KIND == 3) { XOR1(vacc[j].x, x[j & 3].x); }
                    else if (KIND == 4) { EXP1(vacc[j].x, x[j & 3].x); }
                    else if (KIND == 5) { MULLO(vacc[j].x, x[j & 3].x); }
                    else { ADD1(vacc[j].x, x[j & 3].x); }
                }
            }
        }
    }
    asm volatile("s_nop 15\n s_nop 15" ::: "memory");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3] + acc4[i][1] + vacc[i].x + vacc[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <int MODE, int NVPER, int KIND>
static void run(const char* name, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((rate<MODE, NVPER, KIND>), dim3(256), dim3(512), 0, 0, out, 200, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate<MODE, NVPER, KIND>), dim3(256), dim3(512), 0, 0, out, iters, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: two waves x iters trips; a trip's MFMA pipe time is 8 x 32 cycles
    const double ns_per_trip_pair = ms * 1e6 / iters;
    const double mfma_tflops = MODE == 1 ? 0 : 256.0 * 8 * iters * 8 * 2048.0 / (ms * 1e-3) * 1e-12;
    const double v_flop = (KIND == 2 ? 128.0 : 256.0) * NVPER * 8;
    const double valu_tflops = (MODE == 1 || MODE == 2) ? 256.0 * 8 * iters * v_flop / (ms * 1e-3) * 1e-12 : 0;
    printf("%-44s %8.3f ms  %7.1f ns per trip of both waves of a SIMD   MFMA %6.1f TFLOP/s  VALU %6.1f TFLOP/s  sum %6.1f\n", name, ms,
           ns_per_trip_pair, mfma_tflops, valu_tflops, mfma_tflops + valu_tflops);
}

int main() {
    float* out;
    hipMalloc(&out, 1024);
    run<0, 0, 0>("8 x 16x16x4 only", out);
    run<3, 0, 0>("8 x 16x16x4 + 8 x 4x4x1 (hybrid fragment)", out);
    run<1, 2, 0>("16 pk_fma only", out);
    run<1, 4, 0>("32 pk_fma only", out);
    run<2, 1, 0>("8 MFMA + 8 pk_fma", out);
    run<2, 2, 0>("8 MFMA + 16 pk_fma", out);
    run<2, 2, 1>("8 MFMA + 16 pk_fma (broadcast operand)", out);
    run<2, 3, 0>("8 MFMA + 24 pk_fma", out);
    run<2, 4, 0>("8 MFMA + 32 pk_fma", out);
    run<2, 6, 0>("8 MFMA + 48 pk_fma", out);
    run<2, 4, 2>("8 MFMA + 32 v_fma_f32", out);
    run<1, 4, 2>("32 v_fma_f32 only", out);
    run<2, 4, 3>("8 MFMA + 32 v_xor_b32", out);
    run<1, 4, 3>("32 v_xor_b32 only", out);
    run<2, 4, 6>("8 MFMA + 32 v_add_f32", out);
    run<1, 4, 6>("32 v_add_f32 only", out);
    run<2, 2, 4>("8 MFMA + 16 v_exp_f32", out);
    run<1, 2, 4>("16 v_exp_f32 only", out);
    run<2, 2, 5>("8 MFMA + 16 v_mul_lo_u32", out);
    run<1, 2, 5>("16 v_mul_lo_u32 only", out);
    return 0;
}

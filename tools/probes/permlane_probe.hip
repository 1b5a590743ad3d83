// Probe: v_permlane16_swap_b32 / v_permlane32_swap_b32 semantics on gfx950 and their use as one-instruction xor-16 / xor-32 reductions.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float opaque(float v) { asm volatile("" : "+v"(v)); return v; }
// hipcc's __builtin_amdgcn_permlane{16,32}_swap returns vdst' in BOTH result elements (ROCm 7.2): inline asm instead.
// (a', b') = swap(a, b): 16: odd rows of a <-> even rows of b;  32: lanes 32..63 of a <-> lanes 0..31 of b.
__device__ __forceinline__ void swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b)); }
__global__ void k(const float* x, float* o16a, float* o16b, float* o32a, float* o32b, float* sum) {
    const int l = threadIdx.x;
    const float v = x[l], w = opaque(v);
    float a0 = v, a1 = w;
    swap16(a0, a1);
    o16a[l] = a0; o16b[l] = a1;
    float b0 = v, b1 = w;
    swap32(b0, b1);
    o32a[l] = b0; o32b[l] = b1;
    float s = a0 + a1, s2 = s;
    swap32(s, s2);
    sum[l] = s + s2;
}
int main() {
    float hx[64], h[5][64]; float* d[6];
    for (int i = 0; i < 64; ++i) hx[i] = i;
    for (int i = 0; i < 6; ++i) hipMalloc(&d[i], 256);
    hipMemcpy(d[0], hx, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5]);
    for (int i = 0; i < 5; ++i) hipMemcpy(h[i], d[i + 1], 256, hipMemcpyDeviceToHost);
    const char* nm[5] = {"permlane16_swap vdst'", "permlane16_swap src0'", "permlane32_swap vdst'", "permlane32_swap src0'", "sum over lanes l, l^16, l^32, l^48"};
    for (int i = 0; i < 5; ++i) { printf("%-36s:", nm[i]); for (int l = 0; l < 64; l += (i < 4 ? 8 : 1)) printf(" %g", h[i][l]); printf("\n"); }
    int ok = 1; for (int l = 0; l < 64; ++l) ok &= h[4][l] == (float)((l & 15) * 4 + 16 + 32 + 48);
    printf("xor-16/xor-32 reduction by two swaps + two adds: %s\n", ok ? "CORRECT" : "WRONG");
    return 0;
}

// Probe: issue cost of the integer instructions of the dropout hash on gfx950 (one wave, 8 independent chains, s_memtime around 4096 ops per chain).
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHAINS 8
#define ITER 512
template <int OP>
__global__ void k(unsigned* out, unsigned long long* cyc, unsigned seed) {
    unsigned v[CHAINS];
    for (int c = 0; c < CHAINS; ++c) v[c] = threadIdx.x * 2654435761u + c + seed;
    const unsigned m = seed | 0x9E3779B1u;
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int i = 0; i < ITER; ++i) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            if (OP == 0) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[c]) : "v"(m));
            if (OP == 1) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(v[c]) : "v"(m));
            if (OP == 2) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[c]) : "v"(m));
            if (OP == 3) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(v[c]) : "v"(m));
            if (OP == 4) asm volatile("v_lshrrev_b32 %0, 15, %0" : "+v"(v[c]));
            if (OP == 5) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(v[c]) : "v"(m));
            if (OP == 6) asm volatile("v_exp_f32 %0, %0" : "+v"(v[c]));
        }
    }
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned s = 0;
    for (int c = 0; c < CHAINS; ++c) s ^= v[c];
    out[threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    unsigned* o; unsigned long long* c; unsigned long long h;
    hipMalloc(&o, 1024); hipMalloc(&c, 8);
    const char* nm[7] = {"v_mul_lo_u32", "v_mul_u32_u24", "v_xor_b32", "v_mad_u32_u24", "v_lshrrev_b32", "v_mul_hi_u32", "v_exp_f32"};
#define RUN(OP) for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k<OP>, dim3(1), dim3(64), 0, 0, o, c, 1u); hipDeviceSynchronize(); } \
    hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost); printf("%-16s %6.2f cycles per instruction (one wave)\n", nm[OP], (double)h / (CHAINS * ITER));
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
    return 0;
}

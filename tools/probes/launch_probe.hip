// Measurement probe: how long does a grid of ONE-PER-CU workgroups take to get going and to drain?
// 256 workgroups x 512 threads with a large dynamic LDS allocation (the shape of the fused encoder / head kernels): every
// workgroup stamps s_memrealtime (100 MHz, one counter for the whole device) at entry, spins for a fixed number of core cycles
// (s_memtime), stamps again.  Printed: spread of the entry stamps, spread of the exit stamps, and the kernel's duration by HIP
// events against the spin time — i.e. the part of a kernel's duration that no amount of work inside the workgroups can remove.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/launch_probe.out tools/probes/launch_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(512) void probe(unsigned long long* t0, unsigned long long* t1, unsigned* xcc, long long spin) {
    extern __shared__ float lds[];
    const unsigned long long a = __builtin_amdgcn_s_memrealtime();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) lds[0] = 1.f;
    while ((long long)(__builtin_amdgcn_s_memtime() - c0) < spin) __builtin_amdgcn_s_sleep(8);
    __syncthreads();
    const unsigned long long b = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        t0[blockIdx.x] = a;
        t1[blockIdx.x] = b;
        unsigned x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        xcc[blockIdx.x] = x & 0xf;
    }
}

int main() {
    const int G = 256;
    unsigned long long *t0, *t1;
    unsigned* xcc;
    hipMalloc(&t0, G * 8); hipMalloc(&t1, G * 8); hipMalloc(&xcc, G * 4);
    for (int lds_kb : {0, 64, 127, 147}) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, lds_kb * 1024);
        for (long long spin : {24000LL, 240000LL}) {      // ~10 us and ~100 us at 2.4 GHz
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            float best = 1e9f;
            std::vector<unsigned long long> h0(G), h1(G);
            for (int rep = 0; rep < 20; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(probe, dim3(G), dim3(512), lds_kb * 1024, 0, t0, t1, xcc, spin);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) {
                    best = ms;
                    hipMemcpy(h0.data(), t0, G * 8, hipMemcpyDeviceToHost);
                    hipMemcpy(h1.data(), t1, G * 8, hipMemcpyDeviceToHost);
                }
            }
            const unsigned long long s0 = *std::min_element(h0.begin(), h0.end()), s1 = *std::max_element(h0.begin(), h0.end());
            const unsigned long long f0 = *std::min_element(h1.begin(), h1.end()), f1 = *std::max_element(h1.begin(), h1.end());
            std::vector<unsigned long long> hs(h0);
            std::sort(hs.begin(), hs.end());
            printf("LDS %3d KB  spin %6.1f us | event %7.2f us | first entry -> last entry %5.2f us (median %5.2f) | first exit -> last exit %5.2f us | "
                   "first entry -> last exit %7.2f us\n",
                   lds_kb, spin / 2400.0, best * 1e3, (s1 - s0) / 100.0, (hs[G / 2] - s0) / 100.0, (f1 - f0) / 100.0, (f1 - s0) / 100.0);
        }
    }
    return 0;
}

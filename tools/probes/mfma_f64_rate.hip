// Throughput probe of v_mfma_f64_16x16x4_f64 on gfx950: NCH independent accumulator chains per wave, 256 / 512 threads per workgroup,
// one workgroup per CU x 4.   hipcc --offload-arch=gfx950 -O2 mfma_f64_rate.hip -o mfma_f64_rate.out
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NCH>
__global__ void rate(double* out, int iters) {
    d4 c[NCH];
    for (int i = 0; i < NCH; ++i) c[i] = d4{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NCH; ++i) c[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NCH; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH>
static void run(int threads, int wgs_per_cu) {
    int dev = 0; hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
    const int grid = p.multiProcessorCount * wgs_per_cu, iters = 20000;
    double* d; hipMalloc(&d, (size_t)grid * threads * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<NCH>, dim3(grid), dim3(threads), 0, 0, d, 1000);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<NCH>, dim3(grid), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * (threads / 64) * (double)iters * NCH * 2048.0;
    const double waves_per_simd = (double)wgs_per_cu * (threads / 64) / 4.0;
    printf("chains %d, %3d threads/WG, %d WG/CU (%.1f waves/SIMD): %.3f ms, %.1f TFLOP/s, %.1f cycles @2.4GHz per MFMA per SIMD\n", NCH, threads, wgs_per_cu,
           waves_per_simd, ms, flops / ms * 1e-9, ms * 1e-3 * 2.4e9 / ((double)iters * NCH * waves_per_simd));
    hipFree(d);
}
int main() {
    run<1>(256, 1); run<4>(256, 1); run<8>(256, 1);
    run<1>(512, 1); run<4>(512, 1);
    run<4>(256, 2); run<4>(256, 4);
    return 0;
}

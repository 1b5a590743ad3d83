// Layout probe of v_mfma_f64_16x16x4_f64 on gfx950: which (row, column) does accumulator element e of lane l hold, and which
// (row / column, k) does the A / B operand of lane l feed?   hipcc --offload-arch=gfx950 -O2 mfma_f64_probe.hip -o p && ./p
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
    const int l = threadIdx.x;
    // test 1: assume A lane = (row l%16, k l/16), B lane = (col l%16, k l/16).  A[i][k] = (k == 2) * (i + 1), B[k][j] = (k == 2) * (100 * (j + 1))
    //          -> D[i][j] = 100 (i + 1)(j + 1) if the operand assumption holds (any other k mapping gives 0 or a different pattern)
    const double a = (l / 16 == 2) ? (l % 16 + 1) : 0.0;
    const double b = (l / 16 == 2) ? 100.0 * (l % 16 + 1) : 0.0;
    d4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = c[e];
}
int main() {
    double* d; hipMalloc(&d, 256 * 8);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 5) for (int e = 0; e < 4; ++e) {
        const int v = (int)(h[l * 4 + e] / 100.0 + 0.5);   // (i+1)(j+1)
        printf("lane %2d e %d: value %6.0f  -> (i+1)(j+1) = %d\n", l, e, h[l * 4 + e], v);
    }
    return 0;
}

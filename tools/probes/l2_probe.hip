// Probe: what a kernel boundary does to the L2 of an XCD on gfx950 (is a weight slice read by launch N still an L2 hit for launch N + 1?
// what does it cost to read what ANOTHER XCD's workgroup wrote in the previous launch?).  One wave on a chosen workgroup id (ids go
// round-robin over the 8 XCDs) issues NB independent 16-byte loads per lane (64 lanes x 16 B = 1 KiB each, L1 bypassed: sc0) and stamps
// s_memtime around the batch.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NB = 8;

__global__ void reader(const float* buf, int bytes, int who, int rounds, unsigned long long* out, float* sink) {
    if ((int)blockIdx.x != who) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(buf), 0, bytes, 0x00020000);
    const int lane = threadIdx.x;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int r = 0; r < rounds; ++r) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        f32x4 v[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) v[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane * 16, (r * NB + k) * 1024, 1 /* sc0: skip L1 */));
#pragma unroll
        for (int k = 0; k < NB; ++k) acc += v[k];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();
        if (lane == 0) out[r] = t1 - t0;
    }
    if (acc.x == 123.456f) sink[lane] = acc.y;
}
__global__ void writer(float* buf, int floats, int who, float val) {
    if ((int)blockIdx.x != who) return;
    for (int i = threadIdx.x; i < floats; i += blockDim.x) buf[i] = val + i;
}
__global__ void xcc_of(unsigned* out) {
    unsigned xcc; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) out[blockIdx.x] = xcc & 0xf;
}

__global__ void empty_kernel(float* p, int n) { if (n < 0) p[threadIdx.x] = 0.f; }
// a dependent chain through memory: every launch reads what the previous one wrote (16 workgroups x 256 threads, one float4 each)
__global__ void chain_kernel(const float* in, float* out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    out[i] = in[(i * 37) & 4095] + 1.f;
}

int main() {
    const int BYTES = 4 << 20;
    float *w, *a, *sink; unsigned long long* st; unsigned* xc;
    hipMalloc(&w, BYTES); hipMalloc(&a, BYTES); hipMalloc(&sink, 1024); hipMalloc(&st, 8 * 64); hipMalloc(&xc, 64);
    hipMemset(w, 0, BYTES); hipMemset(a, 0, BYTES);
    unsigned hx[16]; hipLaunchKernelGGL(xcc_of, dim3(16), dim3(64), 0, 0, xc); hipMemcpy(hx, xc, 64, hipMemcpyDeviceToHost);
    printf("xcc of workgroup ids 0..15:"); for (int i = 0; i < 16; ++i) printf(" %u", hx[i]); printf("\n");
    unsigned long long h[64];
    auto run = [&](const char* tag, const float* buf, int who, int rounds) {
        hipLaunchKernelGGL(reader, dim3(16), dim3(64), 0, 0, buf, BYTES, who, rounds, st, sink);
        hipMemcpy(h, st, 8 * rounds, hipMemcpyDeviceToHost);
        printf("%-58s", tag); for (int r = 0; r < rounds; ++r) printf(" %6llu", h[r]); printf("  cycles per batch of %d KiB\n", NB);
    };
    // 1. the same 32 KiB read by four consecutive launches from the same workgroup id (same XCD): a hit from launch 2 on?
    for (int rep = 0; rep < 2; ++rep) {
        hipMemset(w, 0, BYTES); hipDeviceSynchronize();
        run("same id 0, launch 1 (after a memset: cold)", w, 0, 4);
        run("same id 0, launch 2", w, 0, 4);
        run("same id 0, launch 3", w, 0, 4);
        run("id 8 (same XCD as id 0), launch 4", w, 8, 4);
        run("id 1 (another XCD), launch 5", w, 1, 4);
        run("id 1 again, launch 6", w, 1, 4);
    }
    // 2. within one launch: round r+4 re-reads round r?  (rounds walk forward: all distinct -> each is a first touch; then a second launch)
    // 3. producer / consumer across a kernel boundary
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(writer, dim3(16), dim3(256), 0, 0, a, 8192 * 4, 1, 1.0f + rep);     // written on XCD of id 1
        run("written by id 1 (other XCD) in the previous launch", a, 0, 4);
        hipLaunchKernelGGL(writer, dim3(16), dim3(256), 0, 0, a, 8192 * 4, 8, 2.0f + rep);     // written on XCD of id 8 (= XCD of id 0)
        run("written by id 8 (same XCD) in the previous launch", a, 0, 4);
        hipLaunchKernelGGL(writer, dim3(16), dim3(256), 0, 0, a, 8192 * 4, 0, 3.0f + rep);
        run("written by id 0 itself in the previous launch", a, 0, 4);
    }
    // 4. what a kernel boundary costs in a stream: N back-to-back launches of an empty kernel / of a dependent chain, device period
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {16, 64, 128, 512}) {
        for (int threads : {256, 512}) {
            for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(threads), 0, 0, w, 0);
            hipDeviceSynchronize();
            hipEventRecord(e0, 0);
            for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(empty_kernel, dim3(grid), dim3(threads), 0, 0, w, 0);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("empty kernel grid %3d x %3d threads: %.2f us per launch (2000 back to back)\n", grid, threads, ms / 2.0f);
            (void)0;
        }
    }
    {
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 1000; ++i) {
            hipLaunchKernelGGL(chain_kernel, dim3(16), dim3(256), 0, 0, w, a);
            hipLaunchKernelGGL(chain_kernel, dim3(16), dim3(256), 0, 0, a, w);
        }
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("dependent chain (read 16 KiB the previous launch wrote, write 16 KiB), grid 16 x 256: %.2f us per launch\n", ms / 2.0f);
    }
    return 0;
}

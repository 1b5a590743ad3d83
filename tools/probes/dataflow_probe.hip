// Probe: a chain of K dependent stages inside ONE launch (dedicated workgroups per stage, producer -> consumer flags through
// memory, consumers resident and spinning before their inputs exist) against the same chain as K kernel launches.
// Stage s has G workgroups; each reads the 16 KiB its predecessor stage wrote (every workgroup reads ALL of it: an all-to-all
// seam, like a GEMM whose K dimension is the previous stage's output), adds 1 and writes its own 16 KiB / G share.
// Flags: one 64-bit word per producer workgroup = epoch; stores / loads at agent scope (sc1), as the cooperating kernels of
// libtip_hip do across XCDs.  Workgroup ids are stage-major, so a consumer only ever waits for LOWER ids (in-order dispatch:
// no co-residency requirement beyond "somebody earlier is running").
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

// LOCAL: only every 8th workgroup id works (all on ONE XCD): plain stores into the shared L2, L1-bypassing loads
template <bool FLOW, bool LOCAL = false>
__global__ __launch_bounds__(256) void stage_kernel(float* bufs, u64* flags, int G, int stage0, u64 epoch, u64* stamps) {
    if (LOCAL && (blockIdx.x & 7)) return;
    const int bid = LOCAL ? blockIdx.x >> 3 : blockIdx.x;
    const int stage = FLOW ? bid / G : stage0, nb = bid % G;
    const int tid = threadIdx.x, lane = tid & 63;
    const float* in = bufs + (size_t)(stage & 1) * 4096;        // ping-pong 16 KiB buffers
    float* out = bufs + (size_t)((stage + 1) & 1) * 4096;
    __shared__ float red[256];
    if (FLOW && stage > 0) {
        if (tid < 64) {
            const u64* f = flags + (size_t)(stage - 1) * 64;
            for (unsigned spin = 0; spin < (1u << 20); ++spin) {
                const u64 v = lane < G ? __hip_atomic_load(f + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
                if (__builtin_amdgcn_ballot_w64(v != epoch) == 0) break;
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    }
    // read all 16 KiB of the predecessor (coherent loads), reduce, write own share
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(in), 0, 16384, 0x00020000);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k) acc += __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, tid * 16, k * 4096, FLOW ? 16 : 0));
    red[tid] = (acc.x + acc.y) + (acc.z + acc.w);
    __syncthreads();
    const int share = 4096 / G;                                  // floats this workgroup writes
    for (int i = tid; i < share; i += 256) {
        const float v = red[i & 255] * 1e-6f + 1.f;
        if (FLOW && !LOCAL) __hip_atomic_store(out + nb * share + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else out[nb * share + i] = v;
    }
    if (FLOW) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (LOCAL) flags[(size_t)stage * 64 + nb] = epoch;
            else __hip_atomic_store(flags + (size_t)stage * 64 + nb, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (nb == 0 && stamps) stamps[stage] = __builtin_amdgcn_s_memtime();
        }
    }
}

int main() {
    float* bufs; u64 *flags, *stamps;
    const int K = 20;
    hipMalloc(&bufs, 2 * 16384); hipMalloc(&flags, K * 64 * 8); hipMalloc(&stamps, K * 8);
    hipMemset(bufs, 0, 2 * 16384); hipMemset(flags, 0, K * 64 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    u64 epoch = 1;
    for (int G : {16, 32, 64}) {
        float ms;
        // (a) K launches
        for (int rep = 0; rep < 50; ++rep) for (int s = 0; s < K; ++s) hipLaunchKernelGGL(stage_kernel<false>, dim3(G), dim3(256), 0, 0, bufs, flags, G, s, 0ull, nullptr);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int rep = 0; rep < 200; ++rep) for (int s = 0; s < K; ++s) hipLaunchKernelGGL(stage_kernel<false>, dim3(G), dim3(256), 0, 0, bufs, flags, G, s, 0ull, nullptr);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const float per_launch = ms * 1000.f / (200 * K);
        // (b) one launch, dataflow
        for (int rep = 0; rep < 50; ++rep) hipLaunchKernelGGL(stage_kernel<true>, dim3(G * K), dim3(256), 0, 0, bufs, flags, G, 0, epoch++, stamps);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int rep = 0; rep < 200; ++rep) hipLaunchKernelGGL(stage_kernel<true>, dim3(G * K), dim3(256), 0, 0, bufs, flags, G, 0, epoch++, stamps);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const float per_chain = ms * 1000.f / 200;
        // (c) one launch, dataflow, every workgroup on XCD 0
        for (int rep = 0; rep < 50; ++rep) hipLaunchKernelGGL((stage_kernel<true, true>), dim3(G * K * 8), dim3(256), 0, 0, bufs, flags, G, 0, epoch++, stamps);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        for (int rep = 0; rep < 200; ++rep) hipLaunchKernelGGL((stage_kernel<true, true>), dim3(G * K * 8), dim3(256), 0, 0, bufs, flags, G, 0, epoch++, stamps);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        const float per_chain_l = ms * 1000.f / 200;
        u64 h[K]; hipMemcpy(h, stamps, K * 8, hipMemcpyDeviceToHost);
        printf("G = %2d workgroups per stage, %d stages: %d launches %.2f us per stage | one dataflow launch %.2f us per chain = %.2f us per stage | "
               "the same on ONE XCD %.2f us per chain = %.2f us per stage (stamps: %.0f cycles per hop)\n", G, K, K, per_launch, per_chain, per_chain / K,
               per_chain_l, per_chain_l / K, (double)(h[K - 1] - h[1]) / (K - 2));
    }
    return 0;
}

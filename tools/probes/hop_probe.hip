// Probe: one-way latency of a cross-CU hand-off through L2 on gfx950 (ping-pong between two workgroups).
// Variants: store kind (plain / sc1 atomic-relaxed-agent) x load kind (sc1 buffer load / atomic load), same-XCD vs cross-XCD
// placement (block ids 8 apart vs adjacent), payload 4 B / 16 B per lane (one wave = 256 B / 1 KB).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// grid = 16 blocks; the pair is (block 0, block PARTNER).  Other blocks exit.  word w of round i: value i (from side A) etc.
template <int STORE, int WIDE>
__global__ void pingpong(unsigned* buf, int partner, int iters, unsigned long long* cyc, unsigned* xcc_out) {
    const int me = blockIdx.x == 0 ? 0 : (blockIdx.x == partner ? 1 : -1);
    if (me < 0) return;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) xcc_out[me] = xcc & 0xf;
    const int lane = threadIdx.x;
    // two mailboxes of 64 lanes x (WIDE ? 16 : 4) bytes: mailbox m is written by side m
    unsigned* mine = buf + me * 1024 + lane * (WIDE ? 4 : 1);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, 8192, 0x00020000);
    const int their_off = ((1 - me) * 1024 + lane * (WIDE ? 4 : 1)) * 4;
    unsigned long long t0 = 0;
    for (int i = 1; i <= iters; ++i) {
        if (i == 11 && lane == 0) t0 = __builtin_amdgcn_s_memtime();
        if (me == 0) {
            // send i, then wait for the echo i
            if (STORE == 0) { if (WIDE) *reinterpret_cast<u32x4*>(mine) = (u32x4){(unsigned)i, (unsigned)i, (unsigned)i, (unsigned)i}; else *mine = i; }
            else { for (int e = 0; e < (WIDE ? 4 : 1); ++e) __hip_atomic_store(mine + e, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
        // wait for the partner's word i (side 1 waits first, side 0 after sending)
        for (int spin = 0; spin < 200000; ++spin) {   // bounded: a hand-off that never becomes visible must not hang the GPU
            asm volatile("" ::: "memory");   // the poll must re-load every trip (a side-effect-free spin is deleted by the optimiser)
            unsigned got;
            if (WIDE) { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, their_off, 0, 16); got = min(min(v.x, v.y), min(v.z, v.w)); }
            else got = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, their_off, 0, 16);
            if (__builtin_amdgcn_ballot_w64(got < (unsigned)i) == 0) break;
        }
        if (me == 1) {
            if (STORE == 0) { if (WIDE) *reinterpret_cast<u32x4*>(mine) = (u32x4){(unsigned)i, (unsigned)i, (unsigned)i, (unsigned)i}; else *mine = i; }
            else { for (int e = 0; e < (WIDE ? 4 : 1); ++e) __hip_atomic_store(mine + e, (unsigned)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        }
    }
    if (lane == 0 && me == 0) cyc[0] = __builtin_amdgcn_s_memtime() - t0;
}

int main() {
    unsigned* buf; unsigned long long* cyc; unsigned* xcc;
    hipMalloc(&buf, 8192); hipMalloc(&cyc, 8); hipMalloc(&xcc, 8);
    const int iters = 1010;
    for (int partner : {8, 1}) {
        for (int store = (partner == 1 ? 1 : 0); store < 2; ++store)   // plain stores stay in the writer's XCD L2: same-XCD only
            for (int wide = 0; wide < 2; ++wide) {
                hipMemset(buf, 0, 8192);
                if (store == 0 && wide == 0) hipLaunchKernelGGL((pingpong<0, 0>), dim3(16), dim3(64), 0, 0, buf, partner, iters, cyc, xcc);
                if (store == 0 && wide == 1) hipLaunchKernelGGL((pingpong<0, 1>), dim3(16), dim3(64), 0, 0, buf, partner, iters, cyc, xcc);
                if (store == 1 && wide == 0) hipLaunchKernelGGL((pingpong<1, 0>), dim3(16), dim3(64), 0, 0, buf, partner, iters, cyc, xcc);
                if (store == 1 && wide == 1) hipLaunchKernelGGL((pingpong<1, 1>), dim3(16), dim3(64), 0, 0, buf, partner, iters, cyc, xcc);
                hipDeviceSynchronize();
                unsigned long long c; unsigned x[2];
                hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); hipMemcpy(x, xcc, 8, hipMemcpyDeviceToHost);
                printf("partner block %d (xcc %u vs %u) store=%s payload=%s: one-way hop %.0f shader cycles\n", partner, x[0], x[1],
                       store ? "sc1(atomic relaxed agent)" : "plain", wide ? "16B/lane (1 KB)" : "4B/lane (256 B)", (double)c / (2.0 * (iters - 10)));
            }
    }
    return 0;
}

// tip_pack.hip — building the packed weight image ON the GPU from the 56 state-dict tensors (device pointers).
// Same image, bit for bit, as tip_pack_weights builds on the host (tests compare the two): general-plan section
// (row-major, zero padded), W_hh / out-linear in MFMA fragment order, fused-plan section (fragment order), with the folds
// of DESIGN.md section 4 (channel shuffle :88-89 into in_linear rows, root-velocity zeroing :75 into its columns,
// 1/sqrt(d_head) into W_q, b_ih + b_hh).  One descriptor per destination array; a handful of launches, microseconds —
// which is what makes `load_state_dict` / optimiser-step -> inference round trips cheap (the host path packs 3.7 M
// floats in CPU loops and copies 27 MB over PCIe).
#include "tip_internal.h"

namespace tip {

__global__ __launch_bounds__(256) void pack_ops_kernel(PackBatch pb, float* __restrict__ img) {
    const PackOp& op = pb.ops[blockIdx.y];
    const long long total = (long long)op.N * op.K;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int n, k;
        if (op.frag) {
            // i = ((nb*KB + kb)*64 + lane)*4 + s
            const int s = (int)(i & 3), lane = (int)((i >> 2) & 63);
            const long long blk = i >> 8;
            const int KB = op.K >> 4;
            const int kb = (int)(blk % KB), nb = (int)(blk / KB);
            n = nb * 16 + (lane & 15);
            k = kb * 16 + 4 * (lane >> 4) + s;
        } else {
            n = (int)(i / op.K);
            k = (int)(i - (long long)n * op.K);
        }
        int src_row = n;
        if (op.shuffle_h) src_row = (n % op.shuffle_h) * op.shuffle_dh + n / op.shuffle_h;   // dst row a*H + b <- src row b*dh + a
        float v = 0.f;
        if (n < op.src_rows && k < op.src_cols && !(k >= op.z0 && k < op.z1)) {
            const size_t si = op.transpose ? (size_t)k * op.src_rows + src_row : (size_t)src_row * op.src_cols + k;
            v = op.src[si];
            if (op.src2) v += op.src2[si];
            if (n < op.scale_rows) v *= op.scale;
        }
        img[op.dst_off + i] = v;
    }
}

hipError_t run_pack_ops(const std::vector<PackOp>& ops, float* img, hipStream_t s) {
    for (size_t o = 0; o < ops.size(); o += kPackBatch) {
        PackBatch pb;
        const int n = (int)std::min<size_t>(kPackBatch, ops.size() - o);
        long long maxel = 1;
        for (int i = 0; i < n; ++i) {
            pb.ops[i] = ops[o + i];
            maxel = std::max(maxel, (long long)ops[o + i].N * ops[o + i].K);
        }
        int gx = (int)std::min<long long>((maxel + 255) / 256, 256);
        hipLaunchKernelGGL(pack_ops_kernel, dim3(gx, n), dim3(256), 0, s, pb, img);
    }
    return hipGetLastError();
}

}  // namespace tip

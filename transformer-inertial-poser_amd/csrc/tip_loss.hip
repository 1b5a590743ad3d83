// tip_loss.hip — the reference's training losses and their gradient w.r.t. the prediction (SURVEY.md section 8 row f-2:
// learning_utils.py:13-78 as combined at train_model.py:177-189), fused into one reduction pass and one gradient pass over
// the [B*T, W] prediction rows.  HBM-bound streaming work: 2 reads + 1 write of the rows, no matrix cores.
//
// Row layout (W = n_pose + n_vel + 4*n_sbp columns; the reference's is 108 + 3 + 20):
//   [0, n_pose)            6-D joint rotations    loss_q_only_2axis :62 (mean sq. error * 100) and loss_jerk :38-47
//   [n_pose, +n_vel)       root velocity x, y, z  loss_q_only_2axis :64-78 (rows whose GT x/y are NaN are dropped; *6 / *12)
//   [n_pose + n_vel, W)    n_sbp x (flag, r.xyz)  loss_constr_multi :13-35 (rows with any NaN GT dropped; BCE on sigmoid(flag),
//                                                  (pred - 5*gt)^2 * 4 on the offsets, sum / n_sbp * 2.5)
// The separate reference functions are the same kernels with the other column groups given width 0.
#include "tip_internal.h"

namespace tip {
namespace {

constexpr int kLossRows = 16;        // prediction rows per workgroup
constexpr int kLossThreads = 256;
constexpr int kLossMaxW = 256;       // widest row the tiles are sized for (the reference's is 131)
constexpr int kLossDepth = 10;       // loads a thread keeps in flight while staging a tile (19 x 131 / 256 = 9.7)
constexpr int kLossPart = 8;         // doubles per workgroup: S_pose, S_xy, S_z, N_vel, S_bce, S_off, N_sbp, S_jerk

struct LossShape {
    int B, T, n_pose, n_vel, n_sbp4, terms;
};

// stats[] slots (floats, TIP_LOSS_STATS of them)
enum { ST_TOTAL = 0, ST_Q, ST_C, ST_J, ST_KPOSE, ST_KXY, ST_KZ, ST_KBCE, ST_KOFF, ST_KJ, ST_NVEL, ST_NSBP };

// float / double math of the two instantiations (fp32: the reference's default; fp64: train_model.py --double)
template <typename R> struct Mth;
template <> struct Mth<float> {
    static __device__ __forceinline__ float exp(float x) { return expf(x); }
    static __device__ __forceinline__ float log(float x) { return logf(x); }
    static __device__ __forceinline__ float max(float a, float b) { return fmaxf(a, b); }
};
template <> struct Mth<double> {
    static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
    static __device__ __forceinline__ double log(double x) { return ::log(x); }
    static __device__ __forceinline__ double max(double a, double b) { return fmax(a, b); }
};

// Stage rows [first, first + n) of a [M, W] array (row stride ld) into an LDS tile [n][W], kLossDepth loads per thread in
// flight before the first LDS store (this is latency-bound streaming: the bytes in flight set the rate); with ld == W the
// span is contiguous.
template <typename R>
__device__ __forceinline__ void stage_rows(R* __restrict__ tile, const R* __restrict__ src, long long ld, long long first,
                                           int n, int W) {
    const int tid = threadIdx.x;
    const int total = n * W;
    const R* s = src + first * ld;
    // loads are unconditional (index clamped) so that nothing but the address arithmetic sits between them
    auto batch = [&](auto&& index) {
        for (int e0 = tid; e0 < total; e0 += kLossThreads * kLossDepth) {
            R v[kLossDepth];
#pragma unroll
            for (int u = 0; u < kLossDepth; ++u) v[u] = s[index(min(e0 + u * kLossThreads, total - 1))];
#pragma unroll
            for (int u = 0; u < kLossDepth; ++u)
                if (e0 + u * kLossThreads < total) tile[e0 + u * kLossThreads] = v[u];
        }
    };
    if (ld == W) {
        batch([](int e) { return (long long)e; });
    } else {
        batch([&](int e) {
            const int r = e / W;
            return r * ld + (e - r * W);
        });
    }
}

// which rows of this workgroup's tile count for the masked terms (learning_utils.py:19, :67); gt tile already in LDS
template <typename R>
__device__ __forceinline__ void row_masks(const R* gtile, bool have_gt, const LossShape& sh, int nrows, int W,
                                          unsigned char* vmask, unsigned char* cmask) {
    const int tid = threadIdx.x;
    if (tid < kLossRows) {
        vmask[tid] = tid < nrows && have_gt && sh.n_vel > 0;
        cmask[tid] = tid < nrows && have_gt && sh.n_sbp4 > 0;
    }
    __syncthreads();
    if (have_gt) {
        // a NaN in the GT x / y of the root velocity, or in any constraint column, clears its row's flag (racing writers all
        // store 0)
        const int nxy = sh.n_vel ? 2 : 0, G = nxy + sh.n_sbp4;
        for (int e = tid; e < nrows * G; e += kLossThreads) {
            const int r = e / G, k = e - r * G;
            if (isnan(gtile[r * W + sh.n_pose + (k < nxy ? k : k - nxy + sh.n_vel)])) {
                if (k < nxy) vmask[r] = 0; else cmask[r] = 0;
            }
        }
    }
    __syncthreads();
}

// thread -> (column, rows [rbeg, rend)) of the tile: one column per thread, 256 / W blocks of consecutive rows side by side
// (consecutive rows let the third difference slide through registers: one LDS read per row instead of four)
struct ColWalk {
    int col, rbeg, rend;
};
__device__ __forceinline__ ColWalk col_walk(int W, int nrows) {
    const int tid = threadIdx.x;
    const int rg = kLossThreads / W;     // >= 1 (W <= kLossMaxW)
    const int chunk = (kLossRows + rg - 1) / rg;
    const int b = tid / W;
    return ColWalk{tid - b * W, min(b * chunk, nrows), b < rg ? min(b * chunk + chunk, nrows) : 0};
}

// binary_cross_entropy(sigmoid(x), t) as torch evaluates it in fp32: both logs clamped at -100
template <typename R>
__device__ __forceinline__ R bce_sigmoid(R x, R t) {
    const R p = R(1.0) / (R(1.0) + Mth<R>::exp(-x));
    return (t - R(1.0)) * Mth<R>::max(Mth<R>::log(R(1.0) - p), -R(100.0)) - t * Mth<R>::max(Mth<R>::log(p), -R(100.0));
}

// d/dx of the above through torch's two backward formulas (binary_cross_entropy_backward, eps 1e-12; sigmoid_backward):
// equals p - t until the sigmoid saturates in fp32, exactly 0 after.
template <typename R>
__device__ __forceinline__ R bce_sigmoid_grad(R x, R t) {
    const R p = R(1.0) / (R(1.0) + Mth<R>::exp(-x));
    const R pq = (R(1.0) - p) * p;
    return (p - t) / Mth<R>::max(pq, R(1e-12)) * pq;
}

template <typename R>
__global__ __launch_bounds__(kLossThreads) void loss_partial_kernel(const R* __restrict__ pred, long long ldp,
                                                                    const R* __restrict__ gt, long long ldg, LossShape sh,
                                                                    double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    R* lds = reinterpret_cast<R*>(lds_raw);
    __shared__ unsigned char vmask[kLossRows], cmask[kLossRows];
    __shared__ double red[kLossThreads / 64][kLossPart];
    const int tid = threadIdx.x;
    const int M = sh.B * sh.T;
    const int row0 = blockIdx.x * kLossRows;
    const int nrows = min(kLossRows, M - row0);
    const int W = sh.n_pose + sh.n_vel + sh.n_sbp4;
    R* pt = lds;                                // prediction rows row0 .. row0 + nrows + 2 (3 more for the third difference)
    R* gtl = lds + (kLossRows + 3) * W;         // GT rows row0 .. row0 + nrows - 1
    stage_rows(pt, pred, ldp, row0, min(kLossRows + 3, M - row0), W);
    if (gt) stage_rows(gtl, gt, ldg, row0, nrows, W);
    __syncthreads();
    row_masks(gtl, gt != nullptr, sh, nrows, W, vmask, cmask);
    R s_pose = R(0), s_xy = R(0), s_z = R(0), s_bce = R(0), s_off = R(0), s_j = R(0);
    const ColWalk cw = col_walk(W, nrows);
    if (cw.rbeg < cw.rend) {
        const int col = cw.col;
        const int kind = col < sh.n_pose ? 0 : col < sh.n_pose + sh.n_vel ? (col - sh.n_pose < 2 ? 1 : 2)
                                         : (((col - sh.n_pose - sh.n_vel) & 3) == 0 ? 3 : 4);
        if (kind == 0) {
            const bool doq = sh.terms & TIP_LOSS_Q, doj = sh.terms & TIP_LOSS_J;
            int t = (row0 + cw.rbeg) % sh.T;             // frame index inside the window
            const int left = M - row0 - cw.rbeg;          // rows from rbeg to the end of the array (all staged up to +3)
            R ya = pt[cw.rbeg * W + col];
            R yb = left > 1 ? pt[(cw.rbeg + 1) * W + col] : R(0);
            R yc = left > 2 ? pt[(cw.rbeg + 2) * W + col] : R(0);
            for (int r = cw.rbeg; r < cw.rend; ++r) {
                const R yn = row0 + r + 3 < M ? pt[(r + 3) * W + col] : R(0);
                if (doq) {
                    const R d = ya - gtl[r * W + col];
                    s_pose += d * d;
                }
                if (doj && t + 3 < sh.T) {
                    const R j = yn - R(3.0) * yc + R(3.0) * yb - ya;
                    s_j += j * j;
                }
                ya = yb, yb = yc, yc = yn;
                if (++t == sh.T) t = 0;
            }
        } else if (kind != 3) {
            for (int r = cw.rbeg; r < cw.rend; ++r) {
                const R p = pt[r * W + col];
                if (kind <= 2) {
                    if ((sh.terms & TIP_LOSS_Q) && vmask[r]) {
                        const R d = gtl[r * W + col] - p;
                        if (kind == 1) s_xy += d * d; else s_z += d * d;
                    }
                } else if ((sh.terms & TIP_LOSS_C) && cmask[r]) {
                    const R d = p - gtl[r * W + col] * R(5.0);
                    s_off += d * d;
                }
            }
        }
    }
    // the flag columns (exp + two logs each) one per thread rather than a column's 16 rows in sequence
    // (more than kLossThreads / kLossRows = 16 constraints per row: a thread takes several elements)
    if (sh.terms & TIP_LOSS_C) {
        const int nf = sh.n_sbp4 >> 2;
        for (int e = tid; e < nrows * nf; e += kLossThreads) {
            const int r = e / nf, col = sh.n_pose + sh.n_vel + 4 * (e - r * nf);
            if (cmask[r]) s_bce += bce_sigmoid<R>(pt[r * W + col], gtl[r * W + col]);
        }
    }
    double v[kLossPart] = {s_pose, s_xy, s_z, 0.0, s_bce, s_off, 0.0, s_j};
    if (tid < nrows) {
        v[3] = vmask[tid];
        v[6] = cmask[tid];
    }
#pragma unroll
    for (int i = 0; i < kLossPart; ++i)
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) v[i] += __shfl_xor(v[i], o, 64);
    if ((tid & 63) == 0)
#pragma unroll
        for (int i = 0; i < kLossPart; ++i) red[tid >> 6][i] = v[i];
    __syncthreads();
    if (tid < kLossPart) part[(size_t)blockIdx.x * kLossPart + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// one workgroup: fixed-order sum of the partials, then the losses and the gradient coefficients
template <typename R>
__global__ __launch_bounds__(kLossThreads) void loss_final_kernel(const double* __restrict__ part, int nblocks, LossShape sh,
                                                                  R* __restrict__ stats) {
    __shared__ double red[kLossThreads][kLossPart];
    const int tid = threadIdx.x;
    double v[kLossPart] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int b = tid; b < nblocks; b += kLossThreads)
#pragma unroll
        for (int i = 0; i < kLossPart; ++i) v[i] += part[(size_t)b * kLossPart + i];
#pragma unroll
    for (int i = 0; i < kLossPart; ++i) red[tid][i] = v[i];
    __syncthreads();
    for (int s = kLossThreads / 2; s >= 1; s >>= 1) {
        if (tid < s)
#pragma unroll
            for (int i = 0; i < kLossPart; ++i) red[tid][i] += red[tid + s][i];
        __syncthreads();
    }
    if (tid == 0) {
        const double M = (double)sh.B * sh.T;
        const double S_pose = red[0][0], S_xy = red[0][1], S_z = red[0][2], Nv = red[0][3];
        const double S_bce = red[0][4], S_off = red[0][5], Nc = red[0][6], S_j = red[0][7];
        const double n_c = sh.n_sbp4 / 4;
        const double Nj = (double)sh.B * (sh.T > 3 ? sh.T - 3 : 0) * sh.n_pose;
        double lq = 0.0, lc = 0.0, lj = 0.0;   // an empty mean is 0/0 = NaN, as torch's .mean() of an empty tensor
        if (sh.terms & TIP_LOSS_Q) {
            lq = S_pose / (M * sh.n_pose) * 100.0;
            if (sh.n_vel) lq += S_xy / (2.0 * Nv) * 6.0 + S_z / Nv * 12.0;
        }
        if (sh.terms & TIP_LOSS_C) lc = (S_bce / Nc + S_off / (3.0 * Nc) * 4.0) / n_c * 2.5;
        if (sh.terms & TIP_LOSS_J) lj = S_j / Nj * 100.0;
        stats[ST_TOTAL] = (R)((lc + lq) + lj);      // train_model.py:187-189
        stats[ST_Q] = (R)lq;
        stats[ST_C] = (R)lc;
        stats[ST_J] = (R)lj;
        stats[ST_KPOSE] = (R)(200.0 / (M * sh.n_pose));
        stats[ST_KXY] = (R)(6.0 / Nv);               // 2 * 6 / (2 Nv)
        stats[ST_KZ] = (R)(24.0 / Nv);
        stats[ST_KBCE] = (R)(2.5 / n_c / Nc);
        stats[ST_KOFF] = (R)(2.5 / n_c * 8.0 / (3.0 * Nc));
        stats[ST_KJ] = (R)(200.0 / Nj);
        stats[ST_NVEL] = (R)Nv;
        stats[ST_NSBP] = (R)Nc;
        for (int i = ST_NSBP + 1; i < TIP_LOSS_STATS; ++i) stats[i] = R(0);
    }
}

template <typename R>
__global__ __launch_bounds__(kLossThreads) void loss_grad_kernel(const R* __restrict__ pred, long long ldp,
                                                                 const R* __restrict__ gt, long long ldg, LossShape sh,
                                                                 const R* __restrict__ stats, const R* __restrict__ gout,
                                                                 R* __restrict__ dy, long long ldd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    R* lds = reinterpret_cast<R*>(lds_raw);
    __shared__ unsigned char vmask[kLossRows], cmask[kLossRows];
    const int M = sh.B * sh.T;
    const int row0 = blockIdx.x * kLossRows;
    const int nrows = min(kLossRows, M - row0);
    const int W = sh.n_pose + sh.n_vel + sh.n_sbp4;
    // prediction rows row0 - 3 .. row0 + nrows + 2: y[t] enters the third differences u = t-3 .. t, which reach y[t+3]
    const int lead = min(3, row0);
    R* pt = lds + (3 - lead) * W;               // pt[(r + 3) * W + c] = prediction row row0 + r
    R* gtl = lds + (kLossRows + 6) * W;
    stage_rows(pt, pred, ldp, row0 - lead, min(lead + kLossRows + 3, M - row0 + lead), W);
    if (gt) stage_rows(gtl, gt, ldg, row0, nrows, W);
    __syncthreads();
    row_masks(gtl, gt != nullptr, sh, nrows, W, vmask, cmask);
    const R go = gout ? gout[0] : R(1.0);
    const R k_pose = stats[ST_KPOSE], k_xy = stats[ST_KXY], k_z = stats[ST_KZ];
    const R k_bce = stats[ST_KBCE], k_off = stats[ST_KOFF], k_j = stats[ST_KJ];
    const int tid = threadIdx.x;
    // the flag columns (exp, divide) one per thread rather than a column's 16 rows in sequence
    for (int e = tid, nf = sh.n_sbp4 >> 2; e < nrows * nf; e += kLossThreads) {
        const int r = e / nf, col = sh.n_pose + sh.n_vel + 4 * (e - r * nf);
        R g = R(0);
        if ((sh.terms & TIP_LOSS_C) && cmask[r]) g = k_bce * bce_sigmoid_grad<R>(lds[(3 + r) * W + col], gtl[r * W + col]);
        dy[(size_t)(row0 + r) * ldd + col] = g * go;
    }
    const ColWalk cw = col_walk(W, nrows);
    if (cw.rbeg >= cw.rend) return;
    const int col = cw.col;
    const int kind = col < sh.n_pose ? 0 : col < sh.n_pose + sh.n_vel ? (col - sh.n_pose < 2 ? 1 : 2)
                                     : (((col - sh.n_pose - sh.n_vel) & 3) == 0 ? 3 : 4);
    if (kind == 3) return;
    const R* y0 = lds + 3 * W + col;   // y0[r * W] = prediction (row0 + r, col), r >= -3
    if (kind == 0) {
        // jitter[u] = y[u+3] - 3 y[u+2] + 3 y[u+1] - y[u] for frames u <= T-4 of a window (jv = 0 elsewhere);
        // d loss_j / d y[q] = k_j * (-jv[q] + 3 jv[q-1] - 3 jv[q-2] + jv[q-3]).  jv and y slide through registers.
        const bool doq = sh.terms & TIP_LOSS_Q, doj = (sh.terms & TIP_LOSS_J) && sh.T > 3;
        R jm[3] = {R(0), R(0), R(0)};     // jv[q-3], jv[q-2], jv[q-1]
        if (doj) {
#pragma unroll
            for (int k = 3; k >= 1; --k) {
                const int rr = cw.rbeg - k, qq = row0 + rr;
                if (qq >= 0 && qq % sh.T + 3 < sh.T)
                    jm[3 - k] = y0[(rr + 3) * W] - R(3.0) * y0[(rr + 2) * W] + R(3.0) * y0[(rr + 1) * W] - y0[rr * W];
            }
        }
        int t = (row0 + cw.rbeg) % sh.T;
        const int left = M - row0 - cw.rbeg;
        R ya = y0[cw.rbeg * W];
        R yb = left > 1 ? y0[(cw.rbeg + 1) * W] : R(0);
        R yc = left > 2 ? y0[(cw.rbeg + 2) * W] : R(0);
        for (int r = cw.rbeg; r < cw.rend; ++r) {
            const R yn = row0 + r + 3 < M ? y0[(r + 3) * W] : R(0);
            R g = doq ? k_pose * (ya - gtl[r * W + col]) : R(0);
            if (doj) {
                const R jn = t + 3 < sh.T ? yn - R(3.0) * yc + R(3.0) * yb - ya : R(0);
                R acc = -jn;
                acc += R(3.0) * jm[2];
                acc += -R(3.0) * jm[1];
                acc += jm[0];
                g += k_j * acc;
                jm[0] = jm[1], jm[1] = jm[2], jm[2] = jn;
            }
            dy[(size_t)(row0 + r) * ldd + col] = g * go;
            ya = yb, yb = yc, yc = yn;
            if (++t == sh.T) t = 0;
        }
        return;
    }
    for (int r = cw.rbeg; r < cw.rend; ++r) {
        const R p = y0[r * W];
        R g = R(0);
        if (kind <= 2) {
            if ((sh.terms & TIP_LOSS_Q) && vmask[r]) g = (kind == 1 ? k_xy : k_z) * (p - gtl[r * W + col]);
        } else if ((sh.terms & TIP_LOSS_C) && cmask[r]) {
            g = k_off * (p - gtl[r * W + col] * R(5.0));
        }
        dy[(size_t)(row0 + r) * ldd + col] = g * go;
    }
}

template <typename R>
int check_shape(const R* pred, long long ldp, const R* gt, long long ldg, int B, int T, int n_pose, int n_vel, int n_sbp,
                int terms, LossShape* sh) {
    if (!pred || B < 0 || T < 0 || n_pose < 0 || n_sbp < 0 || (n_vel != 0 && n_vel != 3)) return TIP_ERR_INVALID_ARG;
    if (!terms || (terms & ~(TIP_LOSS_Q | TIP_LOSS_C | TIP_LOSS_J))) return TIP_ERR_INVALID_ARG;
    const long long W = (long long)n_pose + n_vel + 4LL * n_sbp;
    if (W > kLossMaxW) return TIP_ERR_UNSUPPORTED_CONFIG;
    if (W <= 0 || ldp < W || (long long)B * T > (1LL << 31) - 1 - kLossRows) return TIP_ERR_INVALID_ARG;
    if ((terms & (TIP_LOSS_Q | TIP_LOSS_C)) && (!gt || ldg < W)) return TIP_ERR_INVALID_ARG;
    if ((terms & TIP_LOSS_C) && n_sbp == 0) return TIP_ERR_INVALID_ARG;
    *sh = LossShape{B, T, n_pose, n_vel, 4 * n_sbp, terms};
    return TIP_OK;
}

inline int loss_blocks(int B, int T) { return (int)(((long long)B * T + kLossRows - 1) / kLossRows); }

}  // namespace
}  // namespace tip

using namespace tip;

template <typename R>
static int loss_forward_t(const R* pred, long long ld_pred, const R* gt, long long ld_gt, int B, int T, int n_pose, int n_vel,
                          int n_sbp, int terms, R* stats, void* ws, size_t ws_bytes, void* stream) {
    LossShape sh;
    const int rc = check_shape<R>(pred, ld_pred, gt, ld_gt, B, T, n_pose, n_vel, n_sbp, terms, &sh);
    if (rc != TIP_OK) return rc;
    if (!stats || !ws) return TIP_ERR_INVALID_ARG;
    const int nb = loss_blocks(B, T);
    if (ws_bytes < (size_t)(nb > 0 ? nb : 1) * kLossPart * sizeof(double) || reinterpret_cast<uintptr_t>(ws) % 8) return TIP_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    double* part = static_cast<double*>(ws);
    const int W = sh.n_pose + sh.n_vel + sh.n_sbp4;
    if (nb > 0)
        hipLaunchKernelGGL(loss_partial_kernel<R>, dim3(nb), dim3(kLossThreads), (size_t)(2 * kLossRows + 3) * W * sizeof(R), st, pred, ld_pred, gt, ld_gt, sh, part);
    hipLaunchKernelGGL(loss_final_kernel<R>, dim3(1), dim3(kLossThreads), 0, st, part, nb, sh, stats);
    return hipGetLastError() == hipSuccess ? TIP_OK : TIP_ERR_HIP;
}

template <typename R>
static int loss_backward_t(const R* pred, long long ld_pred, const R* gt, long long ld_gt, int B, int T, int n_pose, int n_vel,
                           int n_sbp, int terms, const R* stats, const R* gout, R* dpred, long long ld_dpred, void* stream) {
    LossShape sh;
    const int rc = check_shape<R>(pred, ld_pred, gt, ld_gt, B, T, n_pose, n_vel, n_sbp, terms, &sh);
    if (rc != TIP_OK) return rc;
    if (!stats || !dpred || ld_dpred < sh.n_pose + sh.n_vel + sh.n_sbp4) return TIP_ERR_INVALID_ARG;
    const int nb = loss_blocks(B, T);
    if (nb == 0) return TIP_OK;
    hipLaunchKernelGGL(loss_grad_kernel<R>, dim3(nb), dim3(kLossThreads),
                       (size_t)(2 * kLossRows + 6) * (sh.n_pose + sh.n_vel + sh.n_sbp4) * sizeof(R), static_cast<hipStream_t>(stream), pred, ld_pred, gt, ld_gt,
                       sh, stats, gout, dpred, ld_dpred);
    return hipGetLastError() == hipSuccess ? TIP_OK : TIP_ERR_HIP;
}

extern "C" {

int tip_loss_ws_bytes(int B, int T, size_t* bytes) {
    if (!bytes || B < 0 || T < 0) return TIP_ERR_INVALID_ARG;
    *bytes = (size_t)(loss_blocks(B, T) > 0 ? loss_blocks(B, T) : 1) * kLossPart * sizeof(double);
    return TIP_OK;
}

int tip_loss_forward(const float* pred, long long ld_pred, const float* gt, long long ld_gt, int B, int T, int n_pose, int n_vel,
                     int n_sbp, int terms, float* stats, void* ws, size_t ws_bytes, void* stream) {
    return loss_forward_t<float>(pred, ld_pred, gt, ld_gt, B, T, n_pose, n_vel, n_sbp, terms, stats, ws, ws_bytes, stream);
}
int tip_loss_backward(const float* pred, long long ld_pred, const float* gt, long long ld_gt, int B, int T, int n_pose, int n_vel,
                      int n_sbp, int terms, const float* stats, const float* gout, float* dpred, long long ld_dpred, void* stream) {
    return loss_backward_t<float>(pred, ld_pred, gt, ld_gt, B, T, n_pose, n_vel, n_sbp, terms, stats, gout, dpred, ld_dpred, stream);
}
int tip_loss_forward_f64(const double* pred, long long ld_pred, const double* gt, long long ld_gt, int B, int T, int n_pose, int n_vel,
                         int n_sbp, int terms, double* stats, void* ws, size_t ws_bytes, void* stream) {
    return loss_forward_t<double>(pred, ld_pred, gt, ld_gt, B, T, n_pose, n_vel, n_sbp, terms, stats, ws, ws_bytes, stream);
}
int tip_loss_backward_f64(const double* pred, long long ld_pred, const double* gt, long long ld_gt, int B, int T, int n_pose, int n_vel,
                          int n_sbp, int terms, const double* stats, const double* gout, double* dpred, long long ld_dpred, void* stream) {
    return loss_backward_t<double>(pred, ld_pred, gt, ld_gt, B, T, n_pose, n_vel, n_sbp, terms, stats, gout, dpred, ld_dpred, stream);
}

}  // extern "C"

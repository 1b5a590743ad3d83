// tip_train.hip — training step of TF_RNN_Past_State on the HIP path.
//
// Reference call site: /root/reference/train_model.py:171-196 — model.train(); y = model(x_imu, x_s + noise);
// loss.backward(); clip_grad_norm_; optimizer.step().  The loss and the optimiser stay in PyTorch; this file is the
// model's forward in train mode (activations saved, encoder dropout live) and its backward, returning the gradient
// of the 56 state-dict tensors (simple_transformer_with_state.py:20-46) in state-dict order.
//
// Forward (train) = the layer-by-layer pipeline of the general plan, reading the RAW parameter tensors (they change
// every optimiser step, so nothing is packed on the host):
//   prep (in_linear rows shuffled :88-89 / padded, b_ih + b_hh, W_hh and W_hh^T in MFMA fragment order)
//   -> prologue :63-78 -> in_linear :79 -> L x { QKV, causal SDPA (+P dropout), out-proj + dropout1 + residual, LN1,
//      linear1 + ReLU + dropout, linear2 + dropout2 + residual, LN2 }  (torch TransformerEncoderLayer, post-norm)
//   -> RNN input projection, tanh recurrence :98-99 (clustered register-resident kernel), output projection :102.
// Backward = the same chain reversed: one flexible fp32-MFMA GEMM (three operand-orientation variants) does every
// dX = dY W and, split over the 10 240 row reduction, every dW = dY^T X; LayerNorm / attention / recurrence have
// their own kernels.  The recurrence backward  delta_t = (dH_t + delta_{t+1} W_hh) * (1 - h_t^2)  has the same shape as
// the forward recurrence, so it runs on the same clustered kernel with W_hh^T fragments, time reversed.
//
// Dropout (encoder p = 0.1, torch default, live in train mode): masks are never stored.  keep(site, idx) is a pure
// function of (seed, site, element index) — splitmix64 finaliser, documented in include/tip_hip.h — evaluated in the
// forward epilogues and again in the backward.  Tests rebuild the same masks in numpy.
#include <stdlib.h>
#include <string.h>

#include "tip_internal.h"
#include "tip_layernorm.h"
#include "tip_pgemm.h"

namespace tip {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------------------
// dropout: keep(seed, site, idx) = hash >= thresh, thresh = floor(p * 2^32)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float drop_factor(const Drop d, unsigned long long idx) {
    if (d.thresh == 0) return 1.f;
    return tip_drop_hash_k(d.key, idx) >= d.thresh ? d.scale : 0.f;
}

static Drop make_drop(float p, unsigned long long seed, unsigned site) {
    Drop d;
    d.seed = seed;
    d.site = site;
    d.key = tip_drop_key(seed, site);
    if (p <= 0.f) {
        d.thresh = 0;
        d.scale = 1.f;
    } else {
        double t = (double)p * 4294967296.0;
        if (t > 4294967295.0) t = 4294967295.0;
        d.thresh = (unsigned)t;
        if (d.thresh == 0) d.thresh = 1;
        d.scale = 1.0f / (1.0f - p);
    }
    return d;
}

// the encoder dropout of a step as the other translation units see it (tip_forward_dropout: the same thresh / scale as the kernels here)
TrainDropout make_train_dropout(float p, unsigned long long seed) {
    const Drop d = make_drop(p, seed, 0);
    return TrainDropout{seed, d.thresh, d.scale};
}
bool state_mask_params(float p, unsigned long long seed, unsigned* key, unsigned* thresh) {
    if (!(p >= 0.f) || p >= 1.f) return false;
    const Drop d = make_drop(p, seed, kStateMaskSite);
    *key = d.key;
    *thresh = d.thresh;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------
// tgemm: C[i][j] = epi( sum_k A(i,k) * B(j,k) ),  128x128x16 block tile, 32x32x2 fp32 MFMA, 4 waves as 2x2.
//   operand mode 0: element (r,k) at P[r*ld + k]  (k contiguous: activations, weights as stored)
//   operand mode 1: element (r,k) at P[k*ld + r]  (r contiguous: a weight read transposed, or a [rows x features]
//                                                  activation whose ROW index is the reduction, as in dW = dY^T X)
// Rows beyond `mm`/`nn` and k beyond `kva`/`kvb` read as zero.  blockIdx.z splits the reduction (partials at
// C + z*c_zstride, no epilogue) for the dW GEMMs.
// ---------------------------------------------------------------------------------------------------------------------
struct TG {
    const float* A; long long lda;
    const float* B; long long ldb;
    float* C; long long ldc;
    int mm, nn, kk;         // extents (kk = loop extent, multiple of 16)
    int kva, kvb;           // valid k of each operand
    int c_rows;             // rows of C actually stored (<= mm)
    int klen;               // reduction length per blockIdx.z (multiple of 16)
    long long c_zstride;
    const float* bias;      // [nn] or null
    int relu;
    Drop drop;              // on the value before gate / residual
    const float* gate; long long ldgate; float gate_scale;   // v *= gate[i][j] > 0 ? gate_scale : 0
    const float* res; long long ldres;                       // v += res[i][j]
};

template <int MODE, int ROWS>
struct TileLoader {
    static constexpr int LDT = ROWS + (MODE ? 4 : 2);
    float4 r0, r1;
    __device__ __forceinline__ void load(const float* P, long long ld, int r_base, int rows, int k0, int kvalid, int tid) {
        // explicit branches: a `cond ? *p : zero` select makes the compiler pick between two ADDRESSES and park the zero
        // vector in scratch
        r0 = make_float4(0.f, 0.f, 0.f, 0.f);
        r1 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 0) {
            const int lr = tid >> 2, lk = (tid & 3) * 4;
            const int ra = r_base + lr, rb = ra + 64, k = k0 + lk;
            if (k < kvalid) {
                if (ra < rows) r0 = *reinterpret_cast<const float4*>(P + (long long)ra * ld + k);
                if (ROWS == 128 && rb < rows) r1 = *reinterpret_cast<const float4*>(P + (long long)rb * ld + k);
            }
        } else {
            const int kr = tid >> 4, ic = (tid & 15) * 4;
            const int k = k0 + kr, ia = r_base + ic, ib = ia + 64;
            if (k < kvalid) {
                if (ia < rows) r0 = *reinterpret_cast<const float4*>(P + (long long)k * ld + ia);
                if (ROWS == 128 && ib < rows) r1 = *reinterpret_cast<const float4*>(P + (long long)k * ld + ib);
            }
        }
    }
    __device__ __forceinline__ void stage(float (*S)[LDT], int tid) const {
        if (MODE == 0) {
            const int lr = tid >> 2, lk = (tid & 3) * 4;
            S[lk + 0][lr] = r0.x; S[lk + 1][lr] = r0.y; S[lk + 2][lr] = r0.z; S[lk + 3][lr] = r0.w;
            if (ROWS == 128) {
                S[lk + 0][lr + 64] = r1.x; S[lk + 1][lr + 64] = r1.y; S[lk + 2][lr + 64] = r1.z; S[lk + 3][lr + 64] = r1.w;
            }
        } else {
            const int kr = tid >> 4, ic = (tid & 15) * 4;
            *reinterpret_cast<float4*>(&S[kr][ic]) = r0;
            if (ROWS == 128) *reinterpret_cast<float4*>(&S[kr][ic + 64]) = r1;
        }
    }
};

// TI x TJ block tile (64 or 128 each), 4 waves as 2 x 2, each wave (TI/2) x (TJ/2) as 32x32 MFMA tiles.  The small tiles
// exist for the many GEMMs here whose output is only 256 wide: 10 240 x 256 is 160 tiles of 128 x 128 — fewer than the
// chip has CUs, and one 4-wave workgroup per CU cannot hide its own LDS/MFMA latency.
template <int AM, int BM, int TI, int TJ>
__global__ __launch_bounds__(256) void tgemm_kernel(TG g) {
    constexpr int LDA_T = TileLoader<AM, TI>::LDT, LDB_T = TileLoader<BM, TJ>::LDT;
    constexpr int NI = TI / 64, NJ = TJ / 64;
    __shared__ __attribute__((aligned(16))) float As[2][16][LDA_T];
    __shared__ __attribute__((aligned(16))) float Bs[2][16][LDB_T];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int i0 = blockIdx.y * TI, j0 = blockIdx.x * TJ;
    const int kbeg = blockIdx.z * g.klen;
    const int kend = min(g.kk, kbeg + g.klen);

    f32x16 acc[NI][NJ];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    TileLoader<AM, TI> la;
    TileLoader<BM, TJ> lb;
    const int l31 = lane & 31, lhi = lane >> 5;
    if (kbeg < kend) {
        la.load(g.A, g.lda, i0, g.mm, kbeg, g.kva, tid);
        lb.load(g.B, g.ldb, j0, g.nn, kbeg, g.kvb, tid);
        la.stage(As[0], tid);
        lb.stage(Bs[0], tid);
        __syncthreads();
        int cur = 0;
        for (int k0 = kbeg; k0 < kend; k0 += 16) {
            const bool more = k0 + 16 < kend;
            if (more) {
                la.load(g.A, g.lda, i0, g.mm, k0 + 16, g.kva, tid);
                lb.load(g.B, g.ldb, j0, g.nn, k0 + 16, g.kvb, tid);
            }
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int k = kk * 2 + lhi;
                float a[NI], b[NJ];
#pragma unroll
                for (int i = 0; i < NI; ++i) a[i] = As[cur][k][wm * (TI / 2) + i * 32 + l31];
#pragma unroll
                for (int j = 0; j < NJ; ++j) b[j] = Bs[cur][k][wn * (TJ / 2) + j * 32 + l31];
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            if (more) {
                la.stage(As[cur ^ 1], tid);
                lb.stage(Bs[cur ^ 1], tid);
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    // epilogue.  C/D layout of 32x32: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    float* C = g.C + (long long)blockIdx.z * g.c_zstride;
    const bool plain = gridDim.z > 1;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int col = j0 + wn * (TJ / 2) + j * 32 + l31;
            if (col >= g.nn) continue;
            const float bv = (!plain && g.bias) ? g.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * (TI / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (row >= g.c_rows) continue;
                float v = acc[i][j][r];
                if (!plain) {
                    v += bv;
                    if (g.relu) v = v > 0.f ? v : 0.f;
                    v *= drop_factor(g.drop, (unsigned long long)row * (unsigned)g.nn + (unsigned)col);
                    if (g.gate) v *= g.gate[(long long)row * g.ldgate + col] > 0.f ? g.gate_scale : 0.f;
                    if (g.res) v += g.res[(long long)row * g.ldres + col];
                }
                C[(long long)row * g.ldc + col] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// tgemm16: the (0,0) orientation — both operands k-contiguous, i.e. every forward GEMM and (with the transposed weight
// copies prep makes once per step) every dX GEMM — on v_mfma_f32_16x16x4_f32.  Both tiles go to LDS as they are in
// memory ([row][32 k], straight 16-byte copies, no transposing scatter) and one ds_read_b128 feeds four MFMA k-steps:
// lane (row l15, group lg) holds k = 4*lg .. 4*lg+3 of each 16-wide k-block for A and for B alike, so MFMA step j
// multiplies matching k's and the sum over (lg, j) covers the block (same trick as the fused inference kernels).
// TI x TJ block tile, 4 waves as 2 x 2, BK = 32.
// ---------------------------------------------------------------------------------------------------------------------
template <int TI, int TJ, int WI = 2, int WJ = 2>
__global__ __launch_bounds__(64 * WI * WJ) void tgemm16_kernel(TG g) {
    constexpr int BK = 32, LDK = BK + 4;
    constexpr int THREADS = 64 * WI * WJ;
    constexpr int RI = TI / (16 * WI), RJ = TJ / (16 * WJ);   // 16-row blocks per wave in i and j
    __shared__ __attribute__((aligned(16))) float As[2][TI][LDK];
    __shared__ __attribute__((aligned(16))) float Bs[2][TJ][LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WJ, wn = wave % WJ;
    const int l15 = lane & 15, lg = lane >> 4;
    const int i0 = blockIdx.y * TI, j0 = blockIdx.x * TJ;
    constexpr int NA = TI * (BK / 4) / THREADS, NB = TJ * (BK / 4) / THREADS;   // float4 slots per thread
    // global -> register prefetch runs TWO k-tiles ahead (two register sets), registers -> LDS one tile ahead: a load
    // has two tiles of MFMA work to land before anything waits on it (one tile is ~0.4 us, less than an L2 round trip
    // under load).
    float4 ra[2][NA], rb[2][NB];

    auto fetch = [&](int k0, float4 (&xa)[NA], float4 (&xb)[NB]) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int f = tid + u * THREADS, r = f >> 3, k = k0 + (f & 7) * 4;
            xa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 < g.kk && i0 + r < g.mm && k < g.kva) xa[u] = *reinterpret_cast<const float4*>(g.A + (long long)(i0 + r) * g.lda + k);
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int f = tid + u * THREADS, r = f >> 3, k = k0 + (f & 7) * 4;
            xb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k0 < g.kk && j0 + r < g.nn && k < g.kvb) xb[u] = *reinterpret_cast<const float4*>(g.B + (long long)(j0 + r) * g.ldb + k);
        }
    };
    auto stage = [&](int buf, const float4 (&xa)[NA], const float4 (&xb)[NB]) {
#pragma unroll
        for (int u = 0; u < NA; ++u) {
            const int f = tid + u * THREADS;
            *reinterpret_cast<float4*>(&As[buf][f >> 3][(f & 7) * 4]) = xa[u];
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int f = tid + u * THREADS;
            *reinterpret_cast<float4*>(&Bs[buf][f >> 3][(f & 7) * 4]) = xb[u];
        }
    };

    f32x4 acc[RI][RJ];
#pragma unroll
    for (int i = 0; i < RI; ++i)
#pragma unroll
        for (int j = 0; j < RJ; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int cur) {
#pragma unroll
        for (int kb = 0; kb < BK / 16; ++kb) {
            float4 a[RI], b[RJ];
#pragma unroll
            for (int i = 0; i < RI; ++i) a[i] = *reinterpret_cast<const float4*>(&As[cur][wm * (TI / WI) + i * 16 + l15][kb * 16 + lg * 4]);
#pragma unroll
            for (int j = 0; j < RJ; ++j) b[j] = *reinterpret_cast<const float4*>(&Bs[cur][wn * (TJ / WJ) + j * 16 + l15][kb * 16 + lg * 4]);
#pragma unroll
            for (int i = 0; i < RI; ++i)
#pragma unroll
                for (int j = 0; j < RJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].x, b[j].x, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RI; ++i)
#pragma unroll
                for (int j = 0; j < RJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].y, b[j].y, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RI; ++i)
#pragma unroll
                for (int j = 0; j < RJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].z, b[j].z, acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < RI; ++i)
#pragma unroll
                for (int j = 0; j < RJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i].w, b[j].w, acc[i][j], 0, 0, 0);
        }
    };

    // tiles: 0 -> LDS[0]; 1 -> registers set 1; then per pair of iterations the two register sets alternate
    fetch(0, ra[0], rb[0]);
    fetch(BK, ra[1], rb[1]);
    stage(0, ra[0], rb[0]);
    __syncthreads();
    for (int k0 = 0; k0 < g.kk; k0 += 2 * BK) {
        // even tile k0 lives in LDS[0]; set 1 holds tile k0+BK (in flight); set 0 is free
        fetch(k0 + 2 * BK, ra[0], rb[0]);
        compute(0);
        if (k0 + BK < g.kk) stage(1, ra[1], rb[1]);
        __syncthreads();
        if (k0 + BK >= g.kk) break;
        // odd tile k0+BK lives in LDS[1]; set 0 holds tile k0+2BK (in flight); set 1 is free
        fetch(k0 + 3 * BK, ra[1], rb[1]);
        compute(1);
        if (k0 + 2 * BK < g.kk) stage(0, ra[0], rb[0]);
        __syncthreads();
    }
    // epilogue.  C/D layout of 16x16: col = lane&15, row = 4*(lane>>4) + r.  The gate / residual inputs of a column block are ALL
    // requested before its first store (clamped addresses, unconditional): written element by element the loop compiles to one
    // dependent load -> wait -> store round trip per element (see pgemm_body, tip_pgemm.h).
#pragma unroll
    for (int j = 0; j < RJ; ++j) {
        const int col = j0 + wn * (TJ / WJ) + j * 16 + l15;
        if (col >= g.nn) continue;
        const float bv = g.bias ? g.bias[col] : 0.f;
        float gv[RI][4], rv[RI][4];
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + wm * (TI / WI) + i * 16 + lg * 4 + r;
                const int rc = row < g.c_rows ? row : g.c_rows - 1;
                gv[i][r] = g.gate ? g.gate[(long long)rc * g.ldgate + col] : 1.f;
                rv[i][r] = g.res ? g.res[(long long)rc * g.ldres + col] : 0.f;
            }
#pragma unroll
        for (int i = 0; i < RI; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + wm * (TI / WI) + i * 16 + lg * 4 + r;
                if (row >= g.c_rows) continue;
                float v = acc[i][j][r] + bv;
                if (g.relu) v = v > 0.f ? v : 0.f;
                v *= drop_factor(g.drop, (unsigned long long)row * (unsigned)g.nn + (unsigned)col);
                if (g.gate) v *= gv[i][r] > 0.f ? g.gate_scale : 0.f;
                if (g.res) v += rv[i][r];
                g.C[(long long)row * g.ldc + col] = v;
            }
        }
    }
}

static TG tg_base(const float* A, long long lda, const float* B, long long ldb, float* C, long long ldc, int mm, int nn, int kk) {
    TG g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.mm = mm; g.nn = nn; g.kk = round_up(kk, 16);
    g.kva = kk; g.kvb = kk;
    g.c_rows = mm;
    g.klen = g.kk;
    g.drop.thresh = 0; g.drop.scale = 1.f;
    g.gate_scale = 1.f;
    return g;
}

static thread_local int g_tgemm_cus = 256;   // set from the handle at every entry point (per thread: handles of GPUs with different CU counts may run on different threads)

// (0,0) orientation, no split: 16x16x4 kernel.  kk is padded to the 32-wide k tile (k beyond kva/kvb reads as zero).
// Tile: 64 x 64 always — measured on MI355X at M = 10 240 it beats 64 x 128 and 128 x 128 for every shape of this model
// (N = 256 .. 1024, K = 144 .. 1024): 640+ small workgroups, four resident per CU, hide each other's barrier and
// LDS latency better than fewer big ones.  TIP_TGEMM16_TILE = 1 / 2 / 3 forces 128x128 / 64x128 / 128x128 on 8 waves
// for measurement.
static hipError_t tgemm16_launch(TG g, hipStream_t s) {
    g.kk = round_up(g.kk, 32);
    static int force = -1;
    if (force < 0) force = tip_env("TIP_TGEMM16_TILE") ? atoi(tip_env("TIP_TGEMM16_TILE")) : 0;
    if (force == 3) {   // 8-wave 128 x 128 (measured: no gain over 64 x 64 even at N, K in the thousands: 112 vs 110 TFLOP/s)
        hipLaunchKernelGGL((tgemm16_kernel<128, 128, 4, 2>), dim3((g.nn + 127) / 128, (g.mm + 127) / 128), dim3(512), 0, s, g);
    } else if (force == 1) {
        hipLaunchKernelGGL((tgemm16_kernel<128, 128>), dim3((g.nn + 127) / 128, (g.mm + 127) / 128), dim3(256), 0, s, g);
    } else if (force == 2) {
        hipLaunchKernelGGL((tgemm16_kernel<64, 128>), dim3((g.nn + 127) / 128, (g.mm + 63) / 64), dim3(256), 0, s, g);
    } else {
        hipLaunchKernelGGL((tgemm16_kernel<64, 64>), dim3((g.nn + 63) / 64, (g.mm + 63) / 64), dim3(256), 0, s, g);
    }
    return hipGetLastError();
}

// The same GEMM contract on the panel kernel (tip_pgemm.h) when the weight is available in fragment order and the shape
// fits (N % 512 == 0, K % 128 == 0): the scaled configuration's linears, forward and dX alike.
struct TgEpi {
    TG g;
    typedef f32x4 Col;
    struct Aux {
        f32x4 gate, res;
    };
    __device__ __forceinline__ Col begin(int col) const {
        return g.bias ? *reinterpret_cast<const f32x4*>(g.bias + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ Aux load(int row, int col) const {
        Aux a;
        a.gate = g.gate ? *reinterpret_cast<const f32x4*>(g.gate + (long long)row * g.ldgate + col) : (f32x4){1.f, 1.f, 1.f, 1.f};
        a.res = g.res ? *reinterpret_cast<const f32x4*>(g.res + (long long)row * g.ldres + col) : (f32x4){0.f, 0.f, 0.f, 0.f};
        return a;
    }
    __device__ __forceinline__ void store(int row, int col, f32x4 v, const Col& b, const Aux& a) const {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = v[e];
            if (g.bias) x += b[e];
            if (g.relu) x = x > 0.f ? x : 0.f;
            x *= drop_factor(g.drop, (unsigned long long)row * (unsigned)g.nn + (unsigned)(col + e));
            if (g.gate) x *= a.gate[e] > 0.f ? g.gate_scale : 0.f;
            if (g.res) x += a.res[e];
            v[e] = x;
        }
        *reinterpret_cast<f32x4*>(g.C + (long long)row * g.ldc + col) = v;
    }
};

__global__ __launch_bounds__(pg::THREADS) void pgemm_tg_kernel(TG g, const float* __restrict__ wfrag, int wbytes) {
    const TgEpi epi{g};
    pgemm_body(g.A, (int)g.lda, wfrag, wbytes, g.mm, g.nn, g.kva, epi);
}

// shape fits AND the grid fills the chip: an 80 x 512 panel per workgroup needs M*N/40960 >= 2 x CUs of them to beat the
// 64 x 64-tile kernel (scaled model: B >= 128 windows of 80 frames; measured B=64: slower, B=256: +6 %)
static bool panel_ok(int M, int N, int K) {
    return pgemm_ok(M, N, K) && (long long)(N / pg::COLS) * ((M + pg::ROWS - 1) / pg::ROWS) >= 2LL * g_tgemm_cus;
}

static hipError_t lin_launch(const TG& g, const float* wfrag, hipStream_t s) {
    static int use_pg = -1;   // TIP_TRAIN_PGEMM=0: LDS-tiled kernel everywhere (measurement)
    if (use_pg < 0) use_pg = (tip_env("TIP_TRAIN_PGEMM") && tip_env("TIP_TRAIN_PGEMM")[0] == '0') ? 0 : 1;
    auto al16 = [](const void* q, long long ld) { return !q || ((reinterpret_cast<uintptr_t>(q) & 15) == 0 && (ld & 3) == 0); };
    if (!use_pg || !wfrag || !panel_ok(g.mm, g.nn, g.kva) || g.kva != g.kvb || (long long)g.nn * g.kva * 4 > 0x7fffffffLL ||
        !al16(g.bias, 0) || !al16(g.C, g.ldc) || !al16(g.res, g.ldres) || !al16(g.gate, g.ldgate))   // 16-byte epilogue accesses
        return tgemm16_launch(g, s);
    static PerDeviceFlag attr_flag; bool& attr_set = attr_flag.cur();
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pgemm_tg_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           pg::LDS_BYTES);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(pgemm_tg_kernel, dim3(g.nn / pg::COLS, (g.mm + pg::ROWS - 1) / pg::ROWS), dim3(pg::THREADS), pg::LDS_BYTES, s, g,
                       wfrag, (int)((long long)g.nn * g.kva * 4));
    return hipGetLastError();
}

template <int AM, int BM>
static hipError_t tgemm_launch(const TG& g, int splits, hipStream_t s) {
    auto blocks = [&](int ti, int tj) { return (long long)((g.mm + ti - 1) / ti) * ((g.nn + tj - 1) / tj) * splits; };
    const long long want = 2LL * g_tgemm_cus;
    static int force = -1;   // measurement only: TIP_TGEMM_TILE = 1 (128x128) / 2 (64x128) / 3 (64x64)
    if (force < 0) force = tip_env("TIP_TGEMM_TILE") ? atoi(tip_env("TIP_TGEMM_TILE")) : 0;
    if (force == 1 || (force == 0 && blocks(128, 128) >= want)) {
        hipLaunchKernelGGL((tgemm_kernel<AM, BM, 128, 128>), dim3((g.nn + 127) / 128, (g.mm + 127) / 128, splits), dim3(256), 0, s, g);
    } else if (force == 2 || (force == 0 && blocks(64, 128) >= want)) {
        hipLaunchKernelGGL((tgemm_kernel<AM, BM, 64, 128>), dim3((g.nn + 127) / 128, (g.mm + 63) / 64, splits), dim3(256), 0, s, g);
    } else {
        hipLaunchKernelGGL((tgemm_kernel<AM, BM, 64, 64>), dim3((g.nn + 63) / 64, (g.mm + 63) / 64, splits), dim3(256), 0, s, g);
    }
    return hipGetLastError();
}

// out[i] = sum_z part[z*stride + i]   (z summed in order; four partial chains only to keep loads in flight)
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, long long stride, int Z,
                                                            float* __restrict__ out, long long n) {
    const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i4 >= n) return;
    if (i4 + 3 < n) {
        float4 a[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        int z = 0;
        for (; z + 3 < Z; z += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 b = *reinterpret_cast<const float4*>(part + (z + u) * stride + i4);
                a[u].x += b.x; a[u].y += b.y; a[u].z += b.z; a[u].w += b.w;
            }
        }
        for (; z < Z; ++z) {
            const float4 b = *reinterpret_cast<const float4*>(part + z * stride + i4);
            a[0].x += b.x; a[0].y += b.y; a[0].z += b.z; a[0].w += b.w;
        }
        float4 r;
        r.x = (a[0].x + a[1].x) + (a[2].x + a[3].x);
        r.y = (a[0].y + a[1].y) + (a[2].y + a[3].y);
        r.z = (a[0].z + a[1].z) + (a[2].z + a[3].z);
        r.w = (a[0].w + a[1].w) + (a[2].w + a[3].w);
        if ((reinterpret_cast<uintptr_t>(out + i4) & 15) == 0) {
            *reinterpret_cast<float4*>(out + i4) = r;
        } else {
            out[i4] = r.x; out[i4 + 1] = r.y; out[i4 + 2] = r.z; out[i4 + 3] = r.w;
        }
    } else {
        for (long long i = i4; i < n; ++i) {
            float a = 0.f;
            for (int z = 0; z < Z; ++z) a += part[z * stride + i];
            out[i] = a;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// dwgemm: dW[n][k] = sum_m dY[m][n] * X[m][k] straight from the two row-major activations, no LDS staging, no transposes.
// The reduction index m is the ROW of both operands.  v_mfma_f32_16x16x4 takes from lane (l15, lg) the A element
// (i = l15, kk = lg) and the B element (j = l15, kk = lg); which feature plays "i" (or "j") in which MFMA is ours to choose.
// Lane (l15, lg) loads ONE float4 per operand from row m0 + lg: features 4*l15 .. 4*l15+3.  Component c of that float4 is
// the operand of the "virtual tile" c, made of the features {4*i + c}: 4 interleaved 16-feature tiles per 64-feature block,
// so a pair of 16-byte loads (4 rows x 256 contiguous bytes per wave instruction) feeds 16 MFMAs.  The accumulator of virtual
// tiles (ca, cb) holds dW[64*ia + 4*(4*lg + e) + ca][4*l15 + cb]: the four cb tiles of a lane are 4 consecutive k -> 16-byte stores.
// Per wave: 128 out-features x 64 in-features (3 loads, 32 MFMAs per 4 rows; 128 accumulator registers).  The 4 waves of a
// workgroup share ONE output tile and take every 4th group of rows (in-workgroup split of the reduction, summed through LDS
// in a fixed order), so the split-K partial volume that goes through HBM is a quarter of one-wave-per-tile's.
// ---------------------------------------------------------------------------------------------------------------------
struct DwArgs {
    const float* dY; long long ldy;
    const float* X; long long ldx;
    float* out; long long zstride;     // partial z at out + z*zstride, [n_out][k_in]
    int n_out, k_in, M, klen, splits;  // klen: rows per split, multiple of 16
};

constexpr int kDwDepth = 6;            // row groups in flight per wave (register stages)
template <int NB> constexpr int dw_lds_bytes() { return 2 * 32 * NB * 64 * 4 * (int)sizeof(float); }   // two accumulator images

// NB = 64-feature blocks of X per wave: 1 -> 128 x 64 per wave, 2 waves per SIMD; 2 -> 128 x 128 per wave (256 accumulator
// registers, one wave per SIMD, 1.5x fewer operand bytes per MFMA)
template <int NB>
__device__ __forceinline__ void dwgemm_body(const DwArgs& g, int tile, int z) {
    extern __shared__ float dw_red[];
    constexpr int NT = 32 * NB;        // 16x16 accumulator tiles per wave
    const int lane = threadIdx.x & 63, l15 = lane & 15, lg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_b = g.k_in / (64 * NB);
    const int ta = tile / tiles_b, tb = tile - ta * tiles_b;
    const int kbeg = z * g.klen;
    const int rows = min(g.klen, g.M - kbeg);
    const int ns = ((rows >> 2) - wave + 3) >> 2;          // this wave's row groups: wave, wave + 4, ...
    const float* pa = g.dY + (long long)(kbeg + 4 * wave + lg) * g.ldy + ta * 128 + 4 * l15;
    const float* pb = g.X + (long long)(kbeg + 4 * wave + lg) * g.ldx + tb * (64 * NB) + 4 * l15;
    const long long sa = 16 * g.ldy, sb = 16 * g.ldx;      // one own step = 16 rows further
    f32x4 acc[2][4][NB][4];
#pragma unroll
    for (int ia = 0; ia < 2; ++ia)
#pragma unroll
        for (int ca = 0; ca < 4; ++ca)
#pragma unroll
            for (int ib = 0; ib < NB; ++ib)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) acc[ia][ca][ib][cb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float4 Av[kDwDepth][2], Bv[kDwDepth][NB];
    auto fetch = [&](int d, int st) {
#pragma unroll
        for (int ia = 0; ia < 2; ++ia) Av[d][ia] = *reinterpret_cast<const float4*>(pa + st * sa + 64 * ia);
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) Bv[d][ib] = *reinterpret_cast<const float4*>(pb + st * sb + 64 * ib);
    };
    auto mfmas = [&](int d) {
        float av[2][4], bv[NB][4];
#pragma unroll
        for (int ia = 0; ia < 2; ++ia) av[ia][0] = Av[d][ia].x, av[ia][1] = Av[d][ia].y, av[ia][2] = Av[d][ia].z, av[ia][3] = Av[d][ia].w;
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) bv[ib][0] = Bv[d][ib].x, bv[ib][1] = Bv[d][ib].y, bv[ib][2] = Bv[d][ib].z, bv[ib][3] = Bv[d][ib].w;
#pragma unroll
        for (int ia = 0; ia < 2; ++ia)
#pragma unroll
            for (int ca = 0; ca < 4; ++ca)
#pragma unroll
                for (int ib = 0; ib < NB; ++ib)
#pragma unroll
                    for (int cb = 0; cb < 4; ++cb)
                        acc[ia][ca][ib][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ia][ca], bv[ib][cb], acc[ia][ca][ib][cb], 0, 0, 0);
    };
    if (ns > 0) {
        // The steady-state loop is branch-free: stage d is consumed, then refilled with row group s0 + d + kDwDepth (index
        // clamped to the last group: a harmless reload near the end).  With per-stage `if (step < ns)` guards the compiler
        // shuffled the ring through register copies and waited for each freshly issued load — the prefetch was void.
#pragma unroll
        for (int d = 0; d < kDwDepth; ++d) fetch(d, d < ns ? d : ns - 1);
        int s0 = 0;
        for (; s0 + kDwDepth <= ns; s0 += kDwDepth) {
#pragma unroll
            for (int d = 0; d < kDwDepth; ++d) {
                mfmas(d);
                const int nx = s0 + d + kDwDepth;
                fetch(d, nx < ns ? nx : ns - 1);
                __builtin_amdgcn_sched_barrier(0);   // keep each refill where it is (the scheduler otherwise sinks all of them
                                                     // to the end of the body: stage 0 would wait a full memory round trip)
            }
        }
        // remaining ns - s0 (< kDwDepth) row groups are already in stages 0 .. ns - s0 - 1
#pragma unroll
        for (int d = 0; d < kDwDepth - 1; ++d)
            if (s0 + d < ns) mfmas(d);
    }
    // (w0 + w2) + (w1 + w3) through two accumulator images [tile][lane][4]
    auto tile_of = [&](int t) -> f32x4& { return acc[t / (16 * NB)][(t / (4 * NB)) & 3][(t >> 2) % NB][t & 3]; };
    auto put = [&](float* img) {
#pragma unroll
        for (int t = 0; t < NT; ++t) *reinterpret_cast<f32x4*>(img + (t * 64 + lane) * 4) = tile_of(t);
    };
    auto add = [&](const float* img) {
#pragma unroll
        for (int t = 0; t < NT; ++t) tile_of(t) += *reinterpret_cast<const f32x4*>(img + (t * 64 + lane) * 4);
    };
    float* img0 = dw_red;
    float* img1 = dw_red + NT * 64 * 4;
    if (wave == 2) put(img0);
    if (wave == 3) put(img1);
    __syncthreads();
    if (wave == 0) add(img0);
    if (wave == 1) add(img1);
    __syncthreads();
    if (wave == 1) put(img0);
    __syncthreads();
    if (wave != 0) return;
    add(img0);
    float* o = g.out + (long long)z * g.zstride + (long long)(ta * 128) * g.k_in + tb * (64 * NB) + 4 * l15;
#pragma unroll
    for (int ia = 0; ia < 2; ++ia)
#pragma unroll
        for (int ca = 0; ca < 4; ++ca)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int n = 64 * ia + 4 * (4 * lg + e) + ca;
#pragma unroll
                for (int ib = 0; ib < NB; ++ib)
                    *reinterpret_cast<float4*>(o + (long long)n * g.k_in + 64 * ib) =
                        make_float4(acc[ia][ca][ib][0][e], acc[ia][ca][ib][1][e], acc[ia][ca][ib][2][e], acc[ia][ca][ib][3][e]);
            }
}

// Workgroup -> (tile, split).  Hardware hands consecutive workgroup ids to the 8 XCDs round-robin; id L becomes virtual id
// (L % 8) * ceil(total / 8) + L / 8, so each XCD works through one CONTIGUOUS run of (split, tile) pairs: neighbours in that
// run share their dY / X row slabs, which then sit in that XCD's L2 instead of being fetched into all eight.
__device__ __forceinline__ bool dw_work(int tiles, int splits, int* tile, int* z) {
    const int total = tiles * splits, per_xcd = (total + 7) >> 3;
    const int v = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if ((int)(blockIdx.x >> 3) >= per_xcd || v >= total) return false;
    *z = v / tiles;
    *tile = v - *z * tiles;
    return true;
}
static unsigned dw_grid(int tiles, int splits) { return (unsigned)((tiles * splits + 7) / 8 * 8); }

template <int NB>
__global__ __launch_bounds__(256, NB == 1 ? 2 : 1) void dwgemm_kernel(DwArgs g) {
    int tile, z;
    if (!dw_work((g.n_out >> 7) * (g.k_in / (64 * NB)), g.splits, &tile, &z)) return;
    dwgemm_body<NB>(g, tile, z);
}

// several weight gradients that reduce over the same rows in ONE launch (the four of an encoder layer): their tiles fill the
// chip together, so each workgroup walks 4x more rows than it would in four launches of their own — the pipeline fill, the
// LDS reduction and the tail are paid once, and the split-K partial volume per gradient shrinks accordingly
constexpr int kDwBatchMax = 4;
struct DwBatch {
    DwArgs p[kDwBatchMax];
    int tile0[kDwBatchMax + 1];
    int n;
};
template <int NB>
__global__ __launch_bounds__(256, NB == 1 ? 2 : 1) void dwgemm_batch_kernel(DwBatch b) {
    int tile, z;
    if (!dw_work(b.tile0[b.n], b.p[0].splits, &tile, &z)) return;
    int q = 0;
#pragma unroll
    for (int i = 1; i < kDwBatchMax; ++i)
        if (i < b.n && tile >= b.tile0[i]) q = i;
    // select by value (uniform branch) rather than by pointer into the kernel argument: keeps the descriptor in SGPRs
    if (q == 0) dwgemm_body<NB>(b.p[0], tile - b.tile0[0], z);
    else if (q == 1) dwgemm_body<NB>(b.p[1], tile - b.tile0[1], z);
    else if (q == 2) dwgemm_body<NB>(b.p[2], tile - b.tile0[2], z);
    else dwgemm_body<NB>(b.p[3], tile - b.tile0[3], z);
}

struct ReduceBatch {
    const float* part[kDwBatchMax];
    float* out[kDwBatchMax];
    long long per[kDwBatchMax];
    int splits;
};
// out_q[i] = sum_z part_q[z*per_q + i], z in order; blockIdx.y = q
__global__ __launch_bounds__(256) void splitk_reduce_batch_kernel(ReduceBatch r) {
    const int q = blockIdx.y;
    const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const long long per = r.per[q];
    if (i4 >= per) return;
    const float* p = r.part[q] + i4;
    float4 a = *reinterpret_cast<const float4*>(p), b = make_float4(0.f, 0.f, 0.f, 0.f);
    int z = 1;
    for (; z + 1 < r.splits; z += 2) {   // two chains only to keep loads in flight; fixed order
        const float4 u = *reinterpret_cast<const float4*>(p + z * per);
        const float4 v = *reinterpret_cast<const float4*>(p + (z + 1) * per);
        a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
    }
    if (z < r.splits) {
        const float4 u = *reinterpret_cast<const float4*>(p + z * per);
        a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
    }
    float* o = r.out[q] + i4;
    const float4 t = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    if ((reinterpret_cast<uintptr_t>(o) & 15) == 0) {
        *reinterpret_cast<float4*>(o) = t;
    } else {
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = t.w;
    }
}

static bool dwgemm_ok(long long ldy, long long ldx, int n_rows_pad, int n_store, int K, int M) {
    static int off = -1;   // TIP_DW_KERNEL=tgemm: the LDS-tiled 32x32x2 kernel for every shape (measurement)
    if (off < 0) off = tip_env("TIP_DW_KERNEL") && !strcmp(tip_env("TIP_DW_KERNEL"), "tgemm");
    return !off && n_store == n_rows_pad && n_store % 128 == 0 && K % 64 == 0 && M % 4 == 0 && ldy % 4 == 0 && ldx % 4 == 0;
}

// split count of a dwgemm launch: fewest rounds of resident workgroups times (rows per split + a fixed per-workgroup cost —
// pipeline fill, LDS reduction, 32-KB store ~ the time of 96 rows), plus the second-stage reduce when the reduction is split
static int dw_choose_splits(int tiles, int slots, int M, long long per_total, size_t part_floats, int tile_floats) {
    static int force = -1;   // TIP_DW_SPLITS: fixed split count (measurement)
    if (force < 0) force = tip_env("TIP_DW_SPLITS") ? atoi(tip_env("TIP_DW_SPLITS")) : 0;
    const int smax = std::max(1, (int)std::min<long long>(std::max(1, M / 128), (long long)part_floats / per_total));
    if (force > 0) return std::min(force, smax);
    int best = 1;
    double best_cost = 1e30;
    for (int sp = 1; sp <= smax; ++sp) {
        const int klen_s = round_up((M + sp - 1) / sp, 16);
        const int sp_eff = (M + klen_s - 1) / klen_s;
        const double rounds = (double)((tiles * sp_eff + slots - 1) / slots);
        const double cost = rounds * (klen_s + 96.0) + (sp_eff > 1 ? 24.0 + 0.02 * sp_eff * (double)per_total / tile_floats : 0.0);
        if (cost < best_cost - 1e-9) best_cost = cost, best = sp_eff;
    }
    return best;
}

static hipError_t dw_set_attrs() {
    static PerDeviceFlag done_flag;
    bool& done = done_flag.cur();
    if (done) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(dwgemm_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, dw_lds_bytes<1>());
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(dwgemm_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, dw_lds_bytes<2>());
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(dwgemm_batch_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, dw_lds_bytes<1>());
    if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(dwgemm_batch_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, dw_lds_bytes<2>());
    done = e == hipSuccess;
    return e;
}

struct DwProb {
    const float* dY; long long ldy; int n_out;
    const float* X; long long ldx; int K;
    float* out;
};

// n <= 4 weight gradients over the same M rows in one dwgemm launch + one reduce launch.  False when a shape is not
// eligible (caller then issues them one by one).
static bool grad_weight_batch(const DwProb* pr, int n, int M, float* part, size_t part_floats, int num_cus, hipStream_t s, hipError_t* err) {
    static int on = -1;   // TIP_DW_GROUP=0: one launch per gradient (measurement)
    if (on < 0) on = !(tip_env("TIP_DW_GROUP") && atoi(tip_env("TIP_DW_GROUP")) == 0);
    if (!on || n < 1 || n > kDwBatchMax) return false;
    long long per_total = 0;
    DwBatch b;
    b.n = n;
    b.tile0[0] = 0;
    static int nbsel = -1;   // TIP_DW_NB=2: 128 x 128 per wave in the batched launch too (measurement)
    if (nbsel < 0) nbsel = tip_env("TIP_DW_NB") ? atoi(tip_env("TIP_DW_NB")) : 0;
    int NB = nbsel == 2 ? 2 : 1;
    for (int i = 0; i < n; ++i) {
        if (!dwgemm_ok(pr[i].ldy, pr[i].ldx, pr[i].n_out, pr[i].n_out, pr[i].K, M)) return false;
        if (pr[i].K % 128) NB = 1;
    }
    for (int i = 0; i < n; ++i) {
        b.tile0[i + 1] = b.tile0[i] + (pr[i].n_out / 128) * (pr[i].K / (64 * NB));
        per_total += (long long)pr[i].n_out * pr[i].K;
    }
    for (int i = n; i < kDwBatchMax; ++i) b.tile0[i + 1] = b.tile0[n], b.p[i] = DwArgs{};
    const int tiles = b.tile0[n];
    int splits = dw_choose_splits(tiles, (NB == 1 ? 2 : 1) * num_cus, M, per_total, part_floats, 128 * 64 * NB);
    const int klen = round_up((M + splits - 1) / splits, 16);
    splits = (M + klen - 1) / klen;
    ReduceBatch r;
    r.splits = splits;
    long long off = 0, per_max = 0;
    for (int i = 0; i < kDwBatchMax; ++i) r.part[i] = nullptr, r.out[i] = nullptr, r.per[i] = 0;
    for (int i = 0; i < n; ++i) {
        const long long per = (long long)pr[i].n_out * pr[i].K;
        b.p[i] = DwArgs{pr[i].dY, pr[i].ldy, pr[i].X, pr[i].ldx, splits == 1 ? pr[i].out : part + off, per, pr[i].n_out, pr[i].K, M, klen, splits};
        r.part[i] = part + off; r.out[i] = pr[i].out; r.per[i] = per;
        off += (long long)splits * per;
        per_max = std::max(per_max, per);
    }
    *err = dw_set_attrs();
    if (*err != hipSuccess) return true;
    if (NB == 1) hipLaunchKernelGGL(dwgemm_batch_kernel<1>, dim3(dw_grid(tiles, splits)), dim3(256), dw_lds_bytes<1>(), s, b);
    else hipLaunchKernelGGL(dwgemm_batch_kernel<2>, dim3(dw_grid(tiles, splits)), dim3(256), dw_lds_bytes<2>(), s, b);
    if (splits > 1)
        hipLaunchKernelGGL(splitk_reduce_batch_kernel, dim3((unsigned)((per_max / 4 + 255) / 256), n), dim3(256), 0, s, r);
    *err = hipGetLastError();
    return true;
}

// dW[N x K] = dY^T X with the reduction over the M rows split across the grid; deterministic two-stage sum.
// A(i = n, k = m) = dY[m*ldy + n], B(j = k', k = m) = X[m*ldx + k'].
static hipError_t grad_weight(const float* dY, long long ldy, int n_rows_pad, int n_store, const float* X, long long ldx, int K,
                              int M, float* part, size_t part_floats, float* out, int num_cus, hipStream_t s) {
    if (dwgemm_ok(ldy, ldx, n_rows_pad, n_store, K, M)) {
        static int nbsel = -1;   // TIP_DW_NB: 1 / 2 = X blocks per wave for every shape (measurement)
        if (nbsel < 0) nbsel = tip_env("TIP_DW_NB") ? atoi(tip_env("TIP_DW_NB")) : 0;
        // 128 x 128 per wave once such tiles alone fill the chip (big layers: operand traffic is what binds there, +7 %);
        // 128 x 64 with two waves per SIMD otherwise (more, smaller workgroups; better at the paper model's shapes)
        const bool big = K % 128 == 0 && (n_store / 128) * (K / 128) >= num_cus;
        const int NB = nbsel ? ((nbsel == 2 && K % 128 == 0) ? 2 : 1) : (big ? 2 : 1);
        const int tiles = (n_store / 128) * (K / (64 * NB));
        const int slots = num_cus * (NB == 1 ? 2 : 1);       // workgroups resident at once
        const long long per = (long long)n_store * K;       // multiple of 4
        int splits = dw_choose_splits(tiles, slots, M, per, part_floats, 128 * 64 * NB);
        const int klen = round_up((M + splits - 1) / splits, 16);
        splits = (M + klen - 1) / klen;
        DwArgs a{dY, ldy, X, ldx, splits == 1 ? out : part, per, n_store, K, M, klen, splits};
        hipError_t e = dw_set_attrs();
        if (e != hipSuccess) return e;
        if (NB == 1) hipLaunchKernelGGL(dwgemm_kernel<1>, dim3(dw_grid(tiles, splits)), dim3(256), dw_lds_bytes<1>(), s, a);
        else hipLaunchKernelGGL(dwgemm_kernel<2>, dim3(dw_grid(tiles, splits)), dim3(256), dw_lds_bytes<2>(), s, a);
        if (splits > 1)
            hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((per / 4 + 255) / 256)), dim3(256), 0, s, part, per, splits, out, per);
        return hipGetLastError();
    }
    TG g = tg_base(dY, ldy, X, ldx, part, K, n_rows_pad, K, M);
    g.c_rows = n_store;
    const int tiles = ((K + 127) / 128) * ((n_rows_pad + 127) / 128);
    static int sdiv = -1;   // TIP_DW_SPLITDIV: fewer splits (smaller partial volume), tgemm_launch then picks smaller tiles
    if (sdiv < 0) sdiv = tip_env("TIP_DW_SPLITDIV") ? atoi(tip_env("TIP_DW_SPLITDIV")) : 2;   // measured: 2 is the sweet spot (1: more reduce traffic, 4+: tiles too small)
    int splits = ((2 * num_cus + tiles - 1) / tiles + sdiv - 1) / sdiv;
    const long long per = (long long)n_store * K;
    const long long stride = (per + 3) / 4 * 4;
    if ((long long)splits * stride > (long long)part_floats) splits = (int)((long long)part_floats / stride);
    if (splits < 1) return hipErrorInvalidValue;
    int klen = round_up((g.kk + splits - 1) / splits, 16);
    splits = (g.kk + klen - 1) / klen;
    g.klen = klen;
    g.c_zstride = stride;
    hipError_t e;
    if (splits == 1) {
        g.C = out;
        return tgemm_launch<1, 1>(g, 1, s);
    }
    e = tgemm_launch<1, 1>(g, splits, s);
    if (e != hipSuccess) return e;
    const long long n4 = (per + 3) / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, part, stride, splits, out, per);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// column sums (bias gradients): out[n] = sum_m X[m*ld + n]; two deterministic stages
// ---------------------------------------------------------------------------------------------------------------------
// A data-movement side task that rides in a column-sum launch (grid-stride over all its blocks, after their reduction): the
// backward's pad-rows and shift-rows copies read arrays the neighbouring column sum reads or writes next to, and as launches
// of their own they cost more in launch overhead than in bytes.
struct ColSide {
    int kind = 0;                 // 0 none; 1: dst[r][c] = c < n ? src[r][n-wide row][c] : 0 (row padding n -> npad);
                                  // 2: dst[b,t] = src[b,t-1], dst[b,0] = 0 over float4s (rows of R4 float4s, T rows per window)
    const float* src = nullptr;
    float* dst = nullptr;
    int n = 0, npad = 0;          // kind 1
    int T = 0, R4 = 0;            // kind 2
    long long total = 0;          // elements (kind 1) / float4s (kind 2)
};

// block = 64 columns x 4 row lanes; grid (ceil(N/64), Z)
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ X, long long ld, int M, int N,
                                                          int rows_per, float* __restrict__ part, ColSide side) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int m0 = blockIdx.y * rows_per, m1 = min(M, m0 + rows_per);
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (n < N) {
        int m = m0 + rl;
        for (; m + 12 < m1; m += 16) {
            a0 += X[(long long)m * ld + n];
            a1 += X[(long long)(m + 4) * ld + n];
            a2 += X[(long long)(m + 8) * ld + n];
            a3 += X[(long long)(m + 12) * ld + n];
        }
        for (; m < m1; m += 4) a0 += X[(long long)m * ld + n];
    }
    red[rl][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (rl == 0 && n < N) part[(long long)blockIdx.y * N + n] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
    if (side.kind) {
        // Four elements per trip, all four loads before the first store, through restrict-qualified local pointers: a plain
        // `dst[i] = f(src[..])` grid-stride loop is one dependent load -> wait -> store round trip per element (the optimiser may not
        // move a load across a store it cannot prove disjoint) behind a 64-bit division — 17 us for the 5.9 MB pad copy.  32-bit
        // index arithmetic (the launcher checks the sizes).
        const unsigned nthreads = gridDim.x * gridDim.y * 256u;
        const unsigned t0 = (blockIdx.y * gridDim.x + blockIdx.x) * 256u + threadIdx.x;
        const unsigned total = (unsigned)side.total;
        if (side.kind == 1) {
            const float* __restrict__ src = side.src;
            float* __restrict__ dst = side.dst;
            const unsigned npad = (unsigned)side.npad, nn = (unsigned)side.n;
            for (unsigned i = t0; i < total; i += 4u * nthreads) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned iu = i + u * nthreads;
                    const unsigned ic = iu < total ? iu : total - 1u;
                    const unsigned r = ic / npad, cc = ic - r * npad;
                    v[u] = src[(size_t)r * nn + (cc < nn ? cc : nn - 1u)];
                    if (cc >= nn) v[u] = 0.f;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned iu = i + u * nthreads;
                    if (iu < total) dst[iu] = v[u];
                }
            }
        } else {
            const float4* __restrict__ src = reinterpret_cast<const float4*>(side.src);
            float4* __restrict__ dst = reinterpret_cast<float4*>(side.dst);
            const unsigned R4 = (unsigned)side.R4, TT = (unsigned)side.T;
            for (unsigned i = t0; i < total; i += 4u * nthreads) {
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned iu = i + u * nthreads;
                    const unsigned ic = iu < total ? iu : total - 1u;
                    const unsigned row = ic / R4, t = row % TT;
                    v[u] = src[t ? ic - R4 : ic];
                    if (!t) v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const unsigned iu = i + u * nthreads;
                    if (iu < total) dst[iu] = v[u];
                }
            }
        }
    }
}

// out[n] = sum_z part[z][n]; block = 64 columns x 4 z lanes
// columns n >= nsplit go to out_hi[n - nsplit], columns n >= nsplit2 to out_hi2[n - nsplit2] (a split at N or beyond is
// inactive); `out2` duplicates the low part when given
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ part, int Z, int N, float* __restrict__ out,
                                                        float* __restrict__ out2, int nsplit, float* __restrict__ out_hi,
                                                        int nsplit2 = 0x7fffffff, float* __restrict__ out_hi2 = nullptr) {
    __shared__ float red[4][64];
    const int c = threadIdx.x & 63, zl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (n < N) {
        int z = zl;
        for (; z + 12 < Z; z += 16) {
            a0 += part[(long long)z * N + n];
            a1 += part[(long long)(z + 4) * N + n];
            a2 += part[(long long)(z + 8) * N + n];
            a3 += part[(long long)(z + 12) * N + n];
        }
        for (; z < Z; z += 4) a0 += part[(long long)z * N + n];
    }
    red[zl][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (zl == 0 && n < N) {
        const float v = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        if (n >= nsplit2) {
            out_hi2[n - nsplit2] = v;
        } else if (n >= nsplit) {
            out_hi[n - nsplit] = v;
        } else {
            out[n] = v;
            if (out2) out2[n] = v;
        }
    }
}

// the same reduction for several partial arrays in one launch (blockIdx.y = source): the fused backward leaves every layer's
// per-window partials in place and reduces them all at the end instead of with two small launches per layer
constexpr int kColMulti = 16;
struct ColSrc {
    const float* part;   // [Z][N]
    float* out;          // columns [0, nsplit)
    float* out_hi;       // columns [nsplit, nsplit2)
    float* out_hi2;      // columns [nsplit2, N)
    int N, nsplit, nsplit2;
    int Z = 0;           // partial rows of THIS source (0: the batch-wide ColMulti::Z)
    float* out_dup = nullptr;   // second copy of the low part (b_ih and b_hh receive the same gradient)
};
struct ColMulti {
    ColSrc src[kColMulti];
    int Z;
};
__global__ __launch_bounds__(256) void colreduce_multi_kernel(ColMulti m) {
    __shared__ float red[4][64];
    const ColSrc& q = m.src[blockIdx.y];
    const int N = q.N, Z = q.Z ? q.Z : m.Z;
    if ((int)blockIdx.x * 64 >= N) return;
    const float* __restrict__ part = q.part;
    const int c = threadIdx.x & 63, zl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (n < N) {
        int z = zl;
        for (; z + 12 < Z; z += 16) {
            a0 += part[(long long)z * N + n];
            a1 += part[(long long)(z + 4) * N + n];
            a2 += part[(long long)(z + 8) * N + n];
            a3 += part[(long long)(z + 12) * N + n];
        }
        for (; z < Z; z += 4) a0 += part[(long long)z * N + n];
    }
    red[zl][c] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (zl == 0 && n < N) {
        const float v = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        if (n >= q.nsplit2) q.out_hi2[n - q.nsplit2] = v;
        else if (n >= q.nsplit) q.out_hi[n - q.nsplit] = v;
        else {
            q.out[n] = v;
            if (q.out_dup) q.out_dup[n] = v;
        }
    }
}

constexpr int kColZ = 64;

// `defer` (fused backward): only the partial pass runs here; the source is appended to the one multi-source reduction launched at
// the end of the backward (its partial array must then stay untouched until that launch)
static hipError_t colsum(const float* X, long long ld, int M, int N, float* part, float* out, float* out2, hipStream_t s,
                         ColSrc* defer = nullptr, const ColSide& side = ColSide()) {
    const int rows_per = (M + kColZ - 1) / kColZ;
    const int Z = (M + rows_per - 1) / rows_per;
    if (side.kind && side.total >= (1LL << 30)) return hipErrorInvalidValue;   // the side task indexes with 32 bits
    hipLaunchKernelGGL(colsum_part_kernel, dim3((N + 63) / 64, Z), dim3(256), 0, s, X, ld, M, N, rows_per, part, side);
    if (defer) {
        *defer = ColSrc{part, out, nullptr, nullptr, N, N, N};
        defer->Z = Z;
        defer->out_dup = out2;
        return hipGetLastError();
    }
    hipLaunchKernelGGL(colreduce_kernel, dim3((N + 63) / 64), dim3(256), 0, s, part, Z, N, out, out2, N, nullptr);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// LayerNorm (torch LayerNorm: biased variance, eps 1e-5, affine): forward saving (mean, rstd); backward
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum64(float v) { return wave64_sum(v); }   // DPP + permlane swaps (tip_layernorm.h)

template <int NV>
__global__ __launch_bounds__(256) void tln_fwd_kernel(const float* __restrict__ z, const float* __restrict__ g,
                                                      const float* __restrict__ be, float* __restrict__ x,
                                                      float* __restrict__ stats, int M) {
    constexpr int D = NV * 256;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* zr = z + (size_t)row * D;
    float4 v[NV];
    float sm = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = *reinterpret_cast<const float4*>(zr + (i * 64 + lane) * 4);
        sm += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mean = wave_sum64(sm) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float a = v[i].x - mean, b = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (c * c + d * d);
    }
    const float rstd = 1.0f / sqrtf(wave_sum64(q) / (float)D + 1e-5f);
    if (lane == 0) {
        stats[(size_t)row * 2] = mean;
        stats[(size_t)row * 2 + 1] = rstd;
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        const float4 gg = *reinterpret_cast<const float4*>(g + c);
        const float4 bb = *reinterpret_cast<const float4*>(be + c);
        float4 o;
        o.x = (v[i].x - mean) * rstd * gg.x + bb.x;
        o.y = (v[i].y - mean) * rstd * gg.y + bb.y;
        o.z = (v[i].z - mean) * rstd * gg.z + bb.z;
        o.w = (v[i].w - mean) * rstd * gg.w + bb.w;
        *reinterpret_cast<float4*>(x + (size_t)row * D + c) = o;
    }
}

constexpr int kLnRows = 16;   // rows per workgroup in the backward (4 per wave): 640 workgroups at M = 10 240 keep every SIMD loaded

// dz = rstd * (dyg - mean(dyg) - xhat * mean(dyg * xhat)), dyg = dy * gamma;  partial dgamma = sum dy * xhat, dbeta = sum dy.
// Also writes dzm = dz * keep(site, idx), the gradient that flows into the dropped branch of the residual, and its column
// sums (third partial): that IS the bias gradient of the linear layer feeding the branch (out_proj / linear2).
template <int NV>
__global__ __launch_bounds__(256) void tln_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                      const float* __restrict__ stats, const float* __restrict__ g,
                                                      float* __restrict__ dz, float* __restrict__ dzm, Drop drop,
                                                      float* __restrict__ part, int M) {
    constexpr int D = NV * 256;
    __shared__ float red[4][3][D];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4 gg[NV], dg[NV], db[NV], dm[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        gg[i] = *reinterpret_cast<const float4*>(g + (i * 64 + lane) * 4);
        dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        dm[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int r0 = blockIdx.x * kLnRows;
    for (int rr = wave; rr < kLnRows; rr += 4) {
        const int row = r0 + rr;
        if (row >= M) break;
        const float mean = stats[(size_t)row * 2], rstd = stats[(size_t)row * 2 + 1];
        float4 xh[NV], dyv[NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            const float4 zv = *reinterpret_cast<const float4*>(z + (size_t)row * D + c);
            dyv[i] = *reinterpret_cast<const float4*>(dy + (size_t)row * D + c);
            xh[i].x = (zv.x - mean) * rstd; xh[i].y = (zv.y - mean) * rstd;
            xh[i].z = (zv.z - mean) * rstd; xh[i].w = (zv.w - mean) * rstd;
            const float a = dyv[i].x * gg[i].x, b = dyv[i].y * gg[i].y, cc = dyv[i].z * gg[i].z, d = dyv[i].w * gg[i].w;
            s1 += (a + b) + (cc + d);
            s2 += (a * xh[i].x + b * xh[i].y) + (cc * xh[i].z + d * xh[i].w);
            dg[i].x += dyv[i].x * xh[i].x; dg[i].y += dyv[i].y * xh[i].y; dg[i].z += dyv[i].z * xh[i].z; dg[i].w += dyv[i].w * xh[i].w;
            db[i].x += dyv[i].x; db[i].y += dyv[i].y; db[i].z += dyv[i].z; db[i].w += dyv[i].w;
        }
        const float m1 = wave_sum64(s1) / (float)D, m2 = wave_sum64(s2) / (float)D;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int c = (i * 64 + lane) * 4;
            float4 o;
            o.x = rstd * (dyv[i].x * gg[i].x - m1 - xh[i].x * m2);
            o.y = rstd * (dyv[i].y * gg[i].y - m1 - xh[i].y * m2);
            o.z = rstd * (dyv[i].z * gg[i].z - m1 - xh[i].z * m2);
            o.w = rstd * (dyv[i].w * gg[i].w - m1 - xh[i].w * m2);
            *reinterpret_cast<float4*>(dz + (size_t)row * D + c) = o;
            {
                const unsigned long long idx = (unsigned long long)row * D + c;
                float4 m;
                m.x = o.x * drop_factor(drop, idx);
                m.y = o.y * drop_factor(drop, idx + 1);
                m.z = o.z * drop_factor(drop, idx + 2);
                m.w = o.w * drop_factor(drop, idx + 3);
                *reinterpret_cast<float4*>(dzm + (size_t)row * D + c) = m;
                dm[i].x += m.x; dm[i].y += m.y; dm[i].z += m.z; dm[i].w += m.w;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = (i * 64 + lane) * 4;
        *reinterpret_cast<float4*>(&red[wave][0][c]) = dg[i];
        *reinterpret_cast<float4*>(&red[wave][1][c]) = db[i];
        *reinterpret_cast<float4*>(&red[wave][2][c]) = dm[i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * D; i += 256) {
        const int w = i / D, c = i - w * D;
        part[(size_t)blockIdx.x * 3 * D + i] = (red[0][w][c] + red[1][w][c]) + (red[2][w][c] + red[3][w][c]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// small data-movement kernels
// ---------------------------------------------------------------------------------------------------------------------
// in_linear with the channel shuffle (:88-89) folded into its rows (new row a*H + b <- old row b*dh + a), K padded to InPad,
// and the root-velocity zeroing (:75) folded into its columns
__global__ __launch_bounds__(256) void prep_in_kernel(const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ Wp,
                                                      float* __restrict__ bp, int D, int H, int In, int InPad, int z0, int z1) {
    const int dh = D / H;
    const int total = D * InPad;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int nn = i / InPad, c = i - nn * InPad;
        const int old = (nn % H) * dh + nn / H;
        Wp[i] = (c < In && !(c >= z0 && c < z1)) ? W[(size_t)old * In + c] : 0.f;   // root-velocity columns (:75) read as zero
        if (c == 0) bp[nn] = bias[old];
    }
}

// b_ih + b_hh; W_hh in 16x16x4 B-fragment order for the forward (h W_hh^T) and W_hh^T for the backward (delta W_hh)
__global__ __launch_bounds__(256) void prep_rnn_kernel(const float* __restrict__ Whh, const float* __restrict__ bih,
                                                       const float* __restrict__ bhh, float* __restrict__ ff, float* __restrict__ fb,
                                                       float* __restrict__ bsum, int R) {
    const int KB = R / 16;
    const int total = R * R;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int sidx = i & 3, lane = (i >> 2) & 63, blk = i >> 8;
        const int kb = blk % KB, nb = blk / KB;
        const int n = nb * 16 + (lane & 15), k = kb * 16 + 4 * (lane >> 4) + sidx;
        ff[i] = Whh[(size_t)n * R + k];
        fb[i] = Whh[(size_t)k * R + n];
        if (i < R) bsum[i] = bih[i] + bhh[i];
    }
}

// dU [M][InPad] -> d x_imu [M][NI], d x_s [M][S] (see tip_train_input_grads)
__global__ __launch_bounds__(256) void input_grads_kernel(const float* __restrict__ dU, const float* __restrict__ x_s,
                                                          const float* __restrict__ mask, float scale, float* __restrict__ dx_imu,
                                                          float* __restrict__ dx_s, long long M, int NI, int S, int InPad) {
    const int In = NI + S;
    const long long total = M * In;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long m = i / In;
        const int c = (int)(i - m * In);
        const float g = dU[m * InPad + c];
        if (c < NI) {
            if (dx_imu) dx_imu[m * NI + c] = g;
        } else if (dx_s) {
            const long long j = m * S + (c - NI);
            const float xv = x_s[j];
            dx_s[j] = (xv != xv) ? 0.f : (mask ? g * mask[j] * scale : g);
        }
    }
}

__global__ __launch_bounds__(256) void pad_rows_kernel(const float* __restrict__ src, int n, float* __restrict__ dst, int npad, long long rows) {
    const long long total = rows * npad;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / npad;
        const int c = (int)(i - r * npad);
        dst[i] = c < n ? src[r * n + c] : 0.f;
    }
}

// hprev[b,t] = h[b,t-1], h[b,-1] = 0
__global__ __launch_bounds__(256) void shift_rows_kernel(const float* __restrict__ h, float* __restrict__ hp, int T, int R4, long long total4) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long long)gridDim.x * 256) {
        const long long row = i / R4;
        const int t = (int)(row % T);
        reinterpret_cast<float4*>(hp)[i] = t ? reinterpret_cast<const float4*>(h)[i - R4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// dW_in in state-dict layout: undo the row shuffle, crop the K padding, zero the root-velocity columns (:75)
__global__ __launch_bounds__(256) void finish_in_kernel(const float* __restrict__ dWp, const float* __restrict__ dbp, float* __restrict__ dW,
                                                        float* __restrict__ db, int D, int H, int In, int InPad, int z0, int z1) {
    const int dh = D / H;
    const int total = D * In;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int old = i / In, c = i - old * In;
        const int nn = (old % dh) * H + old / dh;
        dW[i] = (c >= z0 && c < z1) ? 0.f : dWp[(size_t)nn * InPad + c];
        if (c == 0) db[old] = dbp[nn];
    }
}

// transposed weight copies for the dX GEMMs (dX = dY W reads W "the other way"): one launch for all of them.
// dst[c][r] = src[r][c] for r < rows, 0 for rows <= r < ldd (k padding), dst row stride ldd.
constexpr int kMaxTr = 64;
struct TrDesc {
    const float* src;
    float* dst;
    int rows, cols, ldd, tile0, tiles_c;
};
struct TrBatch {
    TrDesc d[kMaxTr];
    int n;
};

__global__ __launch_bounds__(256) void transpose_batch_kernel(TrBatch tb) {
    __shared__ float tile[32][33];
    int di = 0;
    while (di + 1 < tb.n && (int)blockIdx.x >= tb.d[di + 1].tile0) ++di;
    const TrDesc& t = tb.d[di];
    const int lt = blockIdx.x - t.tile0;
    const int tr = lt / t.tiles_c, tc = lt % t.tiles_c;       // tile row (over padded rows) / tile col
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int y = ty; y < 32; y += 8) {
        const int r = tr * 32 + y, c = tc * 32 + tx;
        tile[y][tx] = (r < t.rows && c < t.cols) ? t.src[(size_t)r * t.cols + c] : 0.f;
    }
    __syncthreads();
    for (int y = ty; y < 32; y += 8) {
        const int c = tc * 32 + y, r = tr * 32 + tx;
        if (c < t.cols && r < t.ldd) t.dst[(size_t)c * t.ldd + r] = tile[tx][y];
    }
}

static void tr_add(TrBatch& tb, int& tiles, const float* src, float* dst, int rows, int cols, int ldd) {
    TrDesc& t = tb.d[tb.n++];
    t.src = src; t.dst = dst; t.rows = rows; t.cols = cols; t.ldd = ldd; t.tile0 = tiles;
    t.tiles_c = (cols + 31) / 32;
    tiles += ((ldd + 31) / 32) * t.tiles_c;
}

static int grid_for(long long n) {
    long long b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

// ---------------------------------------------------------------------------------------------------------------------
// buffers
// ---------------------------------------------------------------------------------------------------------------------
struct TrainLayer {
    size_t qkv, ast, att, z1, st1, x1, hid, z2, st2, xo;
    size_t wqkv_t, wo_t, w1_t, w2_t;   // transposed weight copies (prep) for the dX GEMMs
    // fragment-order copies for the panel GEMM (0 = shape not eligible / fused path): forward W, and W^T for the dX products
    size_t qkv_f = 0, wo_f = 0, w1_f = 0, w2_f = 0, qkv_tf = 0, wo_tf = 0, w1_tf = 0, w2_tf = 0;
};
struct TrainSaved {
    size_t win_p, bin_p, bsum, whh_f, whh_b, U, x0, ih, hall, flags;
    size_t wout_t, wih_t;
    size_t wih_f = 0, wih_tf = 0;
    size_t wout_f = 0, wout_tf = 0;   // fused path: W_out fragments [144][R] (forward projection) and W_out^T fragments [R][160] (dH)
    size_t fused_img;   // fused-plan weight image (fragment order), packed on the GPU every step; 0 floats when unsupported
    std::vector<TrainLayer> layers;
    size_t total;
};
struct TrainScratch {
    size_t dyp, dh, delta, hprev, ga, gb, gc, gbig, datt, part, colpart, dwin_p, dbin_p;
    size_t colpart_out, colpart_rnn, colpart_in;   // partial column sums reduced by the backward's last launch (fused path)
    size_t gbig2;          // fused backward: dqkv of the attention half (dff2 / dpre of the FFN half stay alive for the layer's one dW launch)
    size_t bwimg, lnwin;   // fused backward: transposed-weight fragment image, per-window LayerNorm partials [B][3D]
    size_t part_floats;
    size_t total;
};

// shapes win_gemm_kernel / head_ksplit_kernel are instantiated for: the paper configuration's widths
static bool win_gemm_shapes(const Dims& d) { return d.D == 256 && d.R == 512 && d.S > 128 && d.S <= 144; }

static size_t take(size_t& off, size_t n) {
    const size_t o = off;
    off += (n + 63) / 64 * 64;
    return o;
}

// the encoder of the training forward runs as the one fused kernel (paper configuration with the RNN; tip_train_forward), which also
// writes the FFN gates as bits for ffn_bwd_kernel; tip_train_backward asks the same question
static bool train_forward_is_fused(const Dims& d, int T) {
    static int use_fused = -1;   // TIP_TRAIN_FUSED=0: layer-by-layer forward (measurement)
    if (use_fused < 0) use_fused = (tip_env("TIP_TRAIN_FUSED") && tip_env("TIP_TRAIN_FUSED")[0] == '0') ? 0 : 1;
    return use_fused && fused_supported(d, T) && fused_has_rnn_ih(d) && fused_packed_floats(d) > 0;
}

static TrainSaved saved_layout(const Dims& d, int B, int T) {
    TrainSaved L;
    const size_t M = (size_t)B * T;
    size_t off = 0;
    L.win_p = take(off, (size_t)d.D * d.InPad);
    L.bin_p = take(off, d.D);
    L.bsum = take(off, d.R);
    L.whh_f = take(off, (size_t)d.R * d.R);
    L.whh_b = take(off, (size_t)d.R * d.R);
    L.wout_t = take(off, (size_t)(d.with_rnn ? d.R : d.D) * round_up(d.S, 16));   // W_out^T [K_out][S padded]: K_out = R, or D without an RNN
    L.wih_t = take(off, (size_t)d.D * d.R);
    L.U = take(off, M * d.InPad);
    L.x0 = take(off, M * d.D);
    // the panel GEMM serves the layer-by-layer path; the paper configuration's fused kernels have their own images
    const bool frags = !(fused_supported(d, T) && fused_has_rnn_ih(d));
    if (frags && d.with_rnn && panel_ok((int)M, d.R, d.D)) L.wih_f = take(off, (size_t)d.R * d.D);
    if (frags && d.with_rnn && panel_ok((int)M, d.D, d.R)) L.wih_tf = take(off, (size_t)d.R * d.D);
    if (!frags && win_gemm_shapes(d)) {   // the per-window kernels of the fused path (win_gemm_kernel, head_ksplit_kernel)
        L.wih_tf = take(off, (size_t)d.R * d.D);
        L.wout_tf = take(off, (size_t)d.R * 160);
        L.wout_f = take(off, (size_t)144 * d.R);
    }
    for (int l = 0; l < d.L; ++l) {
        TrainLayer t;
        t.qkv = take(off, M * 3 * d.D);
        t.ast = take(off, (size_t)B * d.H * T * 2);
        t.att = take(off, M * d.D);
        t.z1 = take(off, M * d.D);
        t.st1 = take(off, M * 2);
        t.x1 = take(off, M * d.D);
        // + the fused FFN backward's gate bits, one bit per hidden element, behind the rows (tip_fused.hip: hidden_to_hbm)
        t.hid = take(off, M * d.F + (fused_supported(d, T) ? M * (d.F / 32) : 0));
        t.z2 = take(off, M * d.D);
        t.st2 = take(off, M * 2);
        t.xo = take(off, M * d.D);
        t.wqkv_t = take(off, (size_t)3 * d.D * d.D);
        t.wo_t = take(off, (size_t)d.D * d.D);
        t.w1_t = take(off, (size_t)d.F * d.D);
        t.w2_t = take(off, (size_t)d.F * d.D);
        if (frags) {
            if (panel_ok((int)M, 3 * d.D, d.D)) t.qkv_f = take(off, (size_t)3 * d.D * d.D);
            if (panel_ok((int)M, d.D, d.D)) t.wo_f = take(off, (size_t)d.D * d.D);
            if (panel_ok((int)M, d.F, d.D)) t.w1_f = take(off, (size_t)d.F * d.D);
            if (panel_ok((int)M, d.D, d.F)) t.w2_f = take(off, (size_t)d.F * d.D);
            if (panel_ok((int)M, d.D, 3 * d.D)) t.qkv_tf = take(off, (size_t)3 * d.D * d.D);   // dx_in = dqkv Wqkv: N = D, K = 3D
            if (panel_ok((int)M, d.D, d.D)) t.wo_tf = take(off, (size_t)d.D * d.D);
            if (panel_ok((int)M, d.D, d.F)) t.w1_tf = take(off, (size_t)d.F * d.D);            // dx1 = dpre W1:   N = D, K = F
            if (panel_ok((int)M, d.F, d.D)) t.w2_tf = take(off, (size_t)d.F * d.D);            // dhid = dff2 W2:  N = F, K = D
        }
        L.layers.push_back(t);
    }
    L.ih = take(off, M * d.R);
    L.hall = take(off, M * d.R);
    L.flags = take(off, rnn_flag_words(B, T));
    L.fused_img = take(off, fused_packed_floats(d));
    L.total = off;
    return L;
}

static TrainScratch scratch_layout(const Dims& d, int B, int T) {
    TrainScratch S;
    const size_t M = (size_t)B * T;
    const int Sp = round_up(d.S, 16);
    size_t off = 0;
    S.dyp = take(off, M * Sp);
    S.dh = take(off, M * d.R);
    S.delta = take(off, M * d.R);
    S.hprev = take(off, M * d.R);
    S.ga = take(off, M * d.D);
    S.gb = take(off, M * d.D);
    S.gc = take(off, M * d.D);
    S.gbig = take(off, M * (size_t)std::max(std::max(3 * d.D, d.F), d.InPad));   // (InPad: tip_train_input_grads writes dU [M][InPad] here)
    S.datt = take(off, M * d.D);
    S.gbig2 = take(off, M * 3 * d.D);
    size_t wmax = (size_t)d.F * d.D;
    if ((size_t)3 * d.D * d.D > wmax) wmax = (size_t)3 * d.D * d.D;
    if ((size_t)d.R * d.R > wmax) wmax = (size_t)d.R * d.R;
    S.part_floats = wmax * 32;
    S.part = take(off, S.part_floats);
    const size_t ln_parts = (M + kLnRows - 1) / kLnRows * 3 * d.D;
    size_t cmax = (size_t)kColZ * (size_t)(3 * d.D > d.F ? 3 * d.D : d.F);
    if (ln_parts > cmax) cmax = ln_parts;
    S.colpart = take(off, cmax);
    // partials whose reduction is deferred to the end of the backward (output bias, recurrence biases, in_linear bias)
    S.colpart_out = take(off, (size_t)kColZ * round_up(d.S, 64));
    S.colpart_rnn = take(off, (size_t)kColZ * d.R);
    S.colpart_in = take(off, (size_t)kColZ * d.D);
    S.dwin_p = take(off, (size_t)d.D * d.InPad);
    S.dbin_p = take(off, d.D);
    S.bwimg = take(off, fused_bwd_image_floats(d));
    // per layer: [B][3D + F] of the FFN half, then [B][6D] of the attention half (reduced by one launch after the last layer)
    S.lnwin = take(off, (size_t)d.L * B * (size_t)(9 * d.D + d.F));
    S.total = off;
    return S;
}

static bool train_supported(const Dims& d, int B, int T) {
    // with_rnn = False (simple_transformer_with_state.py:43-46: the output projection reads the encoder) and any rnn_hidden the
    // inference path serves (multiples of 64 up to 512; 512 runs on the register-resident cluster kernels, the rest on the streaming one)
    if (d.with_rnn && (d.R % 64 != 0 || d.R > 512)) return false;
    if (d.D % 256 != 0 || d.D > 1024 || (d.D / 256 == 3)) return false;
    if (d.dh != 16 && d.dh != 32 && d.dh != 64) return false;
    if (d.F % 4 != 0 || T < 1 || T > 128 || B < 1 || 4 * d.L + 2 > kMaxTr) return false;
    if ((long long)B * T * (d.with_rnn ? d.R : d.D) * 4 > 0x7fffffffLL) return false;
    return true;
}

template <typename K>
static hipError_t ln_dispatch(int D, K&& k) {
    switch (D / 256) {
        case 1: return k(std::integral_constant<int, 1>());
        case 2: return k(std::integral_constant<int, 2>());
        case 4: return k(std::integral_constant<int, 4>());
        default: return hipErrorInvalidValue;
    }
}

// C[M,N] = epi(A[M,K(lda)] * W[N(+pad),K(ldw)]^T + bias (+ res)); flags: 1 = relu, 2 = residual — the general plan's GEMM
// contract (tip_general.hip launch_gemm) on the 16x16x4 straight-copy kernel.
hipError_t launch_gemm16(const float* A, int lda, const float* W, int ldw, const float* bias, const float* res, int ldres,
                         float* C, int ldc, int M, int N, int K, int flags, hipStream_t s) {
    if (M <= 0) return hipSuccess;
    TG g = tg_base(A, lda, W, ldw, C, ldc, M, N, K);
    g.bias = bias;
    g.relu = flags & 1;
    if (flags & 2) {
        g.res = res;
        g.ldres = ldres;
    }
    return tgemm16_launch(g, s);
}

// workgroups per window tile of the two recurrences.  demoted (TIP_OPT_DEMOTED, after a lost hand-off): 1 — no inter-workgroup
// hand-off in either direction (streaming kernel).  rnn_hidden 512: the register-resident clusters; other widths: as many
// workgroups per 16-window tile of the streaming kernel as the tile count leaves CUs for.
static int auto_cluster(const Dims& d, int B, int num_cus, bool demoted) {
    if (demoted) return 1;
    const int ntiles = (B + kRnnTile - 1) / kRnnTile;
    if (d.R != 512) {
        int c = 8;
        while (c > 1 && (ntiles * c > num_cus || (d.R / 16) % (4 * c))) c >>= 1;
        return c;
    }
    static const bool rows4 = !(tip_env("TIP_RNN_ROWS4") && tip_env("TIP_RNN_ROWS4")[0] == '0');
    if (rows4) return kRnnRows4;
    int c = 16;
    while (c > 4 && ntiles * c > num_cus) c >>= 1;
    return c;   // the resident kernel exists for 4 / 8 / 16
}

}  // namespace tip

using namespace tip;

// state-dict order (simple_transformer_with_state.py:20-46)
enum { P_IN_W = 0, P_IN_B = 1, P_LAYER0 = 2 };
enum { PL_QKV_W = 0, PL_QKV_B, PL_OUT_W, PL_OUT_B, PL_L1_W, PL_L1_B, PL_L2_W, PL_L2_B, PL_N1_W, PL_N1_B, PL_N2_W, PL_N2_B, PL_COUNT };
enum { PR_WIH = 0, PR_WHH, PR_BIH, PR_BHH, PR_LIN_W, PR_LIN_B };

static int train_fail(tip_handle* h, hipError_t e, const char* what) {
    h->last_hip_error = std::string(what) + ": " + hipGetErrorString(e);
    return TIP_ERR_HIP;
}

// true when the forward packs its weight layouts through the fused image's pack launches (paper configuration, both fused halves on):
// only then do the fragment copies the per-window kernels of the backward read (wih_tf, wout_tf) exist in `saved`
static bool train_fused_prep(const Dims& d, int T) {
    static int use_fused_prep = -1;
    if (use_fused_prep < 0) use_fused_prep = (tip_env("TIP_TRAIN_FUSED") && tip_env("TIP_TRAIN_FUSED")[0] == '0') ? 0 : 1;
    return use_fused_prep && fused_supported(d, T) && fused_has_rnn_ih(d) && fused_packed_floats(d) > 0 &&
           fused_bwd_image_floats(d) > 0 && !(tip_env("TIP_TRAIN_FUSED_BWD") && tip_env("TIP_TRAIN_FUSED_BWD")[0] == '0');
}

#define TT(expr, what)                                   \
    do {                                                 \
        hipError_t e_ = (expr);                          \
        if (e_ != hipSuccess) return train_fail(h, e_, what); \
    } while (0)

extern "C" {

int tip_train_bytes(const tip_handle* h, int B, int T, size_t* saved_bytes, size_t* scratch_bytes) {
    if (!h || B < 0 || T < 0) return TIP_ERR_INVALID_ARG;
    if (B == 0 || T == 0) {
        if (saved_bytes) *saved_bytes = 256;
        if (scratch_bytes) *scratch_bytes = 256;
        return TIP_OK;
    }
    if (!train_supported(h->d, B, T)) return TIP_ERR_UNSUPPORTED_CONFIG;
    g_tgemm_cus = h->num_cus;   // the layouts below depend on it (panel-GEMM eligibility)
    if (saved_bytes) *saved_bytes = saved_layout(h->d, B, T).total * sizeof(float);
    if (scratch_bytes) *scratch_bytes = scratch_layout(h->d, B, T).total * sizeof(float);
    return TIP_OK;
}

int tip_train_saved_view(const tip_handle* h, int B, int T, int what, int layer, size_t* float_offset, size_t* floats) {
    if (!h || !float_offset || !floats) return TIP_ERR_INVALID_ARG;
    const Dims& d = h->d;
    if (!train_supported(d, B, T)) return TIP_ERR_UNSUPPORTED_CONFIG;
    if (layer < 0 || layer >= d.L) return TIP_ERR_INVALID_ARG;
    g_tgemm_cus = h->num_cus;
    const TrainSaved L = saved_layout(d, B, T);
    const TrainLayer& t = L.layers[layer];
    const size_t M = (size_t)B * T;
    switch (what) {
        case TIP_SAVED_QKV: *float_offset = t.qkv; *floats = M * 3 * d.D; break;
        case TIP_SAVED_ATT: *float_offset = t.att; *floats = M * d.D; break;
        case TIP_SAVED_X1: *float_offset = t.x1; *floats = M * d.D; break;
        case TIP_SAVED_HID: *float_offset = t.hid; *floats = M * d.F; break;
        case TIP_SAVED_XOUT: *float_offset = t.xo; *floats = M * d.D; break;
        case TIP_SAVED_HALL: *float_offset = L.hall; *floats = M * d.R; break;
        default: return TIP_ERR_INVALID_ARG;
    }
    return TIP_OK;
}

int tip_train_forward(tip_handle* h, const float* const* params, int n_params, const float* x_imu, const float* x_s,
                      const float* keep_mask, float keep_scale, float p_drop, unsigned long long seed, float* y, void* saved,
                      size_t saved_bytes, int B, int T, void* stream) {
    if (!h || !params || !x_imu || !x_s || !y || !saved) return TIP_ERR_INVALID_ARG;
    if (tip_check(h, 0) != TIP_OK) return TIP_ERR_HANDOFF;   // an earlier launch lost a hand-off (sticky; tip_check(h, 1) clears)
    if (n_params != (int)h->tensor_names.size()) return TIP_ERR_INVALID_ARG;
    if (p_drop < 0.f || p_drop >= 1.f) return TIP_ERR_INVALID_ARG;
    const Dims& d = h->d;
    if (!train_supported(d, B, T)) return TIP_ERR_UNSUPPORTED_CONFIG;
    g_tgemm_cus = h->num_cus;
    const TrainSaved L = saved_layout(d, B, T);
    if (reinterpret_cast<uintptr_t>(saved) % 256 || saved_bytes < L.total * sizeof(float)) return TIP_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int ecus = effective_cus(h->num_cus, s);   // the stream's CU mask counts (co-residency of the recurrence's clusters)
    CoopSerial serial(h->device, s);   // see tip_internal.h: forwards of different streams do not overlap on the device
    if (serial.status != hipSuccess) return train_fail(h, serial.status, "stream serialisation");
    float* W = static_cast<float*>(saved);
    const int M = B * T;
    const float* const* rp = params + P_LAYER0 + PL_COUNT * d.L;
    // without an RNN (:43-46) the tail of the state dict is just linear.weight [S, D], linear.bias
    const float* lin_w = d.with_rnn ? rp[PR_LIN_W] : rp[0];
    const float* lin_b = d.with_rnn ? rp[PR_LIN_B] : rp[1];
    const int Kout = d.with_rnn ? d.R : d.D;

    // Paper configuration: the encoder runs as ONE kernel (below) on a fused weight image, and the only other layouts the step
    // needs are W_hh in fragment order (forward / transposed for the backward) and the transposed W_out / W_ih of the two dX GEMMs
    // that stay batch-wide: they ride in the fused image's pack launches (round 3: prep_in, prep_rnn and the batched transpose of
    // every layer's weights were three more launches, ~20 us, most of it for copies only the layer-by-layer path reads).
    const bool fused_prep = train_fused_prep(d, T);
    if (!fused_prep) {
        hipLaunchKernelGGL(prep_in_kernel, dim3(grid_for((long long)d.D * d.InPad)), dim3(256), 0, s, params[P_IN_W], params[P_IN_B],
                           W + L.win_p, W + L.bin_p, d.D, d.H, d.In, d.InPad, d.n_imu_total + d.rootv0, d.n_imu_total + d.rootv1);
        if (d.with_rnn)
            hipLaunchKernelGGL(prep_rnn_kernel, dim3(grid_for((long long)d.R * d.R)), dim3(256), 0, s, rp[PR_WHH], rp[PR_BIH], rp[PR_BHH],
                               W + L.whh_f, W + L.whh_b, W + L.bsum, d.R);
        {
            TrBatch tb;
            tb.n = 0;
            int tiles = 0;
            tr_add(tb, tiles, lin_w, W + L.wout_t, d.S, Kout, round_up(d.S, 16));
            if (d.with_rnn) tr_add(tb, tiles, rp[PR_WIH], W + L.wih_t, d.R, d.D, d.R);
            for (int l = 0; l < d.L; ++l) {
                const float* const* lp = params + P_LAYER0 + PL_COUNT * l;
                const TrainLayer& t = L.layers[l];
                tr_add(tb, tiles, lp[PL_QKV_W], W + t.wqkv_t, 3 * d.D, d.D, 3 * d.D);
                tr_add(tb, tiles, lp[PL_OUT_W], W + t.wo_t, d.D, d.D, d.D);
                tr_add(tb, tiles, lp[PL_L1_W], W + t.w1_t, d.F, d.D, d.F);
                tr_add(tb, tiles, lp[PL_L2_W], W + t.w2_t, d.D, d.F, d.D);
            }
            hipLaunchKernelGGL(transpose_batch_kernel, dim3(tiles), dim3(256), 0, s, tb);
        }
    }
    TT(hipGetLastError(), "train_prep");
    // (the hybrid fused forward stashes U itself: FusedTrain::u)
    static const bool padded_fwd = tip_env("TIP_TRAIN_FWD_PADDED") != nullptr;
    const bool u_in_kernel = fused_prep && !padded_fwd && d.InPad == 224;
    if (!u_in_kernel) TT(launch_prologue(d, x_imu, x_s, keep_mask, keep_mask ? keep_scale : 1.f, W + L.U, M, s), "train_prologue");
    // Paper configuration: the encoder runs as ONE kernel — the fused inference kernel with its activations stashed and the
    // dropout sites live (tip_fused.hip, fused_encoder_kernel<8>) — on a weight image packed on the GPU from the live
    // parameters.  Any other supported configuration takes the layer-by-layer path below.
    const bool fused = train_forward_is_fused(d, T);
    bool hall_armed = false;
    if (fused) {
        std::vector<PackOp> ops;
        fused_pack_ops(d, params, L.fused_img, ops);
        if (fused_prep) {
            auto op = [&](const float* src, size_t dst, int N, int K, int src_rows, int src_cols, int frag, int transpose) {
                PackOp o;
                o.src = src; o.src2 = nullptr; o.dst_off = dst; o.N = N; o.K = K; o.src_rows = src_rows; o.src_cols = src_cols; o.frag = frag;
                o.shuffle_h = 0; o.shuffle_dh = 0; o.z0 = 0; o.z1 = 0; o.scale_rows = 0; o.scale = 1.f; o.transpose = transpose;
                ops.push_back(o);
            };
            op(rp[PR_WHH], L.whh_f, d.R, d.R, d.R, d.R, 1, 0);                              // W_hh fragments (forward recurrence)
            op(rp[PR_WHH], L.whh_b, d.R, d.R, d.R, d.R, 1, 1);                              // W_hh^T fragments (backward recurrence)
            op(rp[PR_LIN_W], L.wout_t, d.R, round_up(d.S, 16), d.R, d.S, 0, 1);             // W_out^T [R][S padded to 16] (dH = dy W_out)
            op(rp[PR_WIH], L.wih_t, d.D, d.R, d.D, d.R, 0, 1);                              // W_ih^T [D][R] (d_enc = delta W_ih)
            if (L.wout_f) {
                op(rp[PR_WIH], L.wih_tf, d.D, d.R, d.D, d.R, 1, 1);                         // ... and in fragment order (win_gemm_kernel)
                op(rp[PR_LIN_W], L.wout_tf, d.R, 160, d.R, d.S, 1, 1);                      // W_out^T fragments [R][160], zero padded (dH)
                op(rp[PR_LIN_W], L.wout_f, 144, d.R, d.S, d.R, 1, 0);                       // W_out fragments [144][R] (forward projection)
            }
        }
        TT(hipMemsetAsync(W + L.fused_img, 0, fused_packed_floats(d) * sizeof(float), s), "train_fused_pack");
        TT(run_pack_ops(ops, W, s), "train_fused_pack");
        FusedTrain tr;
        tr.sv = W;
        const TrainLayer& t0 = L.layers[0];
        auto u64 = [](size_t off) { return (unsigned)(off / 64); };   // take() aligns every array to 64 floats
        tr.x0 = u64(L.x0); tr.qkv = u64(t0.qkv); tr.ast = u64(t0.ast); tr.att = u64(t0.att); tr.z1 = u64(t0.z1);
        tr.st1 = u64(t0.st1); tr.x1 = u64(t0.x1); tr.hid = u64(t0.hid); tr.z2 = u64(t0.z2); tr.st2 = u64(t0.st2); tr.xo = u64(t0.xo);
        tr.layer_stride = d.L > 1 ? u64(L.layers[1].qkv - t0.qkv) : 0;
        tr.u = u_in_kernel ? u64(L.U) : 0;
        const Drop dr = make_drop(p_drop, seed, 0);
        tr.seed = seed; tr.thresh = dr.thresh; tr.scale = dr.scale;
        // the encoder also pre-fills its windows' HALL rows with the recurrence's hand-off sentinel (saves a 21-MB memset)
        hall_armed = rnn_uses_sentinel(d, B, T, auto_cluster(d, B, ecus, h->demoted != 0));
        static const bool padded = tip_env("TIP_TRAIN_FWD_PADDED") != nullptr;   // A/B runs only (tools/train_bench.py)
        TT((padded ? launch_fused_train : launch_fused_train_h)(d, W + L.fused_img, x_imu, x_s, keep_mask, keep_mask ? keep_scale : 1.f,
                                                                 W + L.ih, hall_armed ? W + L.hall : nullptr, tr, B, T, ecus, s),
           "train_fused_encoder");
    } else {
    {
        // fragment-order weight copies (forward and transposed) for the panel GEMM, packed from the live parameters
        std::vector<PackOp> ops;
        auto frag = [&](const float* src, size_t dst, int N, int K, bool tr) {
            if (!dst) return;
            PackOp o;
            o.src = src; o.src2 = nullptr; o.dst_off = dst; o.N = N; o.K = K; o.src_rows = N; o.src_cols = K; o.frag = 1;
            o.shuffle_h = 0; o.shuffle_dh = 0; o.z0 = 0; o.z1 = 0; o.scale_rows = 0; o.scale = 1.f; o.transpose = tr ? 1 : 0;
            ops.push_back(o);
        };
        if (d.with_rnn) {
            frag(rp[PR_WIH], L.wih_f, d.R, d.D, false);
            frag(rp[PR_WIH], L.wih_tf, d.D, d.R, true);
        }
        for (int l = 0; l < d.L; ++l) {
            const float* const* lp = params + P_LAYER0 + PL_COUNT * l;
            const TrainLayer& t = L.layers[l];
            frag(lp[PL_QKV_W], t.qkv_f, 3 * d.D, d.D, false);
            frag(lp[PL_OUT_W], t.wo_f, d.D, d.D, false);
            frag(lp[PL_L1_W], t.w1_f, d.F, d.D, false);
            frag(lp[PL_L2_W], t.w2_f, d.D, d.F, false);
            frag(lp[PL_QKV_W], t.qkv_tf, d.D, 3 * d.D, true);
            frag(lp[PL_OUT_W], t.wo_tf, d.D, d.D, true);
            frag(lp[PL_L1_W], t.w1_tf, d.D, d.F, true);
            frag(lp[PL_L2_W], t.w2_tf, d.F, d.D, true);
        }
        if (!ops.empty()) TT(run_pack_ops(ops, W, s), "train_frag_pack");
    }
    {
        TG g = tg_base(W + L.U, d.InPad, W + L.win_p, d.InPad, W + L.x0, d.D, M, d.D, d.InPad);
        g.bias = W + L.bin_p;
        TT(tgemm16_launch(g, s), "train_in_linear");
    }
    const float* x = W + L.x0;
    for (int l = 0; l < d.L; ++l) {
        const float* const* lp = params + P_LAYER0 + PL_COUNT * l;
        const TrainLayer& t = L.layers[l];
        {
            TG g = tg_base(x, d.D, lp[PL_QKV_W], d.D, W + t.qkv, 3 * d.D, M, 3 * d.D, d.D);
            g.bias = lp[PL_QKV_B];
            TT(lin_launch(g, t.qkv_f ? W + t.qkv_f : nullptr, s), "train_qkv");
        }
        {
            const Drop dr = make_drop(p_drop, seed, (unsigned)(l * 4 + 0));
            const float qs = 1.0f / sqrtf((float)d.dh);
            const float* qkv = W + t.qkv;
            float* att = W + t.att;
            float* ast = W + t.ast;
            const int H = d.H;
            TT(launch_mattn_fwd(qkv, att, ast, B, T, H, d.dh, qs, dr, s), "train_attention");
        }
        {
            TG g = tg_base(W + t.att, d.D, lp[PL_OUT_W], d.D, W + t.z1, d.D, M, d.D, d.D);
            g.bias = lp[PL_OUT_B];
            g.drop = make_drop(p_drop, seed, (unsigned)(l * 4 + 1));
            g.res = x; g.ldres = d.D;
            TT(lin_launch(g, t.wo_f ? W + t.wo_f : nullptr, s), "train_out_proj");
        }
        {
            const float* z = W + t.z1;
            float* xo = W + t.x1;
            float* st = W + t.st1;
            TT(ln_dispatch(d.D, [&](auto nv) {
                   hipLaunchKernelGGL((tln_fwd_kernel<decltype(nv)::value>), dim3((M + 3) / 4), dim3(256), 0, s, z, lp[PL_N1_W],
                                      lp[PL_N1_B], xo, st, M);
                   return hipGetLastError();
               }), "train_ln1");
        }
        {
            TG g = tg_base(W + t.x1, d.D, lp[PL_L1_W], d.D, W + t.hid, d.F, M, d.F, d.D);
            g.bias = lp[PL_L1_B];
            g.relu = 1;
            g.drop = make_drop(p_drop, seed, (unsigned)(l * 4 + 2));
            TT(lin_launch(g, t.w1_f ? W + t.w1_f : nullptr, s), "train_ffn1");
        }
        {
            TG g = tg_base(W + t.hid, d.F, lp[PL_L2_W], d.F, W + t.z2, d.D, M, d.D, d.F);
            g.bias = lp[PL_L2_B];
            g.drop = make_drop(p_drop, seed, (unsigned)(l * 4 + 3));
            g.res = W + t.x1; g.ldres = d.D;
            TT(lin_launch(g, t.w2_f ? W + t.w2_f : nullptr, s), "train_ffn2");
        }
        {
            const float* z = W + t.z2;
            float* xo = W + t.xo;
            float* st = W + t.st2;
            TT(ln_dispatch(d.D, [&](auto nv) {
                   hipLaunchKernelGGL((tln_fwd_kernel<decltype(nv)::value>), dim3((M + 3) / 4), dim3(256), 0, s, z, lp[PL_N2_W],
                                      lp[PL_N2_B], xo, st, M);
                   return hipGetLastError();
               }), "train_ln2");
        }
        x = W + t.xo;
    }
    if (d.with_rnn) {
        TG g = tg_base(x, d.D, rp[PR_WIH], d.D, W + L.ih, d.R, M, d.R, d.D);
        g.bias = W + L.bsum;
        TT(lin_launch(g, L.wih_f ? W + L.wih_f : nullptr, s), "train_rnn_ih");
    }
    }   // layer-by-layer path
    // what the output projection reads: the RNN states, or the encoder output when there is no RNN
    const float* head_in = d.with_rnn ? W + L.hall : (d.L ? W + L.layers[d.L - 1].xo : W + L.x0);
    if (d.with_rnn)
        TT(launch_rnn(d, W + L.ih, W + L.whh_f, W + L.hall, reinterpret_cast<unsigned*>(W + L.flags), B, T, auto_cluster(d, B, ecus, h->demoted != 0),
                      ecus, hall_armed, h->guard(), s), "train_rnn");
    {
        // fused path, windows of 40 frames: the register-resident projection of the inference path (tip_head.hip) on fragments
        // packed from the live W_out (27 -> 17 us at B = 256); otherwise the LDS-tiled GEMM
        static const bool win_k = !(tip_env("TIP_TRAIN_WIN_GEMM") && tip_env("TIP_TRAIN_WIN_GEMM")[0] == '0');
        hipError_t he = hipErrorInvalidValue;
        if (win_k && fused && fused_prep && L.wout_f && T % 40 == 0)
            he = launch_head_ksplit(W + L.hall, d.R, W + L.wout_f, lin_b, y, d.S, M, d.S, d.R, false, ecus, s);
        if (he == hipErrorInvalidValue) {
            TG g = tg_base(head_in, Kout, lin_w, Kout, y, d.S, M, d.S, Kout);
            g.bias = lin_b;
            he = tgemm16_launch(g, s);
        }
        TT(he, "train_head");
    }
    h->forward_count++;
    return TIP_OK;
}

int tip_train_backward(tip_handle* h, const float* const* params, int n_params, const float* dy, const void* saved,
                       size_t saved_bytes, void* scratch, size_t scratch_bytes, float* grads, size_t grads_floats, float p_drop,
                       unsigned long long seed, int B, int T, void* stream) {
    if (!h || !params || !dy || !saved || !scratch || !grads) return TIP_ERR_INVALID_ARG;
    if (tip_check(h, 0) != TIP_OK) return TIP_ERR_HANDOFF;
    if (n_params != (int)h->tensor_names.size()) return TIP_ERR_INVALID_ARG;
    const Dims& d = h->d;
    if (!train_supported(d, B, T)) return TIP_ERR_UNSUPPORTED_CONFIG;
    g_tgemm_cus = h->num_cus;
    const TrainSaved L = saved_layout(d, B, T);
    const TrainScratch S = scratch_layout(d, B, T);
    if (reinterpret_cast<uintptr_t>(saved) % 256 || saved_bytes < L.total * sizeof(float)) return TIP_ERR_WORKSPACE;
    if (reinterpret_cast<uintptr_t>(scratch) % 256 || scratch_bytes < S.total * sizeof(float)) return TIP_ERR_WORKSPACE;
    // gradient offsets in state-dict order
    std::vector<size_t> goff(n_params);
    size_t gtot = 0;
    for (int i = 0; i < n_params; ++i) {
        goff[i] = gtot;
        const auto& sh = h->tensor_shapes[i];
        gtot += (size_t)sh.first * (sh.second ? sh.second : 1);
    }
    if (grads_floats < gtot) return TIP_ERR_INVALID_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    CoopSerial serial(h->device, s);   // see tip_internal.h: forwards of different streams do not overlap on the device
    if (serial.status != hipSuccess) return train_fail(h, serial.status, "stream serialisation");
    g_tgemm_cus = h->num_cus;
    const float* W = static_cast<const float*>(saved);
    float* X = static_cast<float*>(scratch);
    const int M = B * T;
    const int Sp = round_up(d.S, 16);
    const int ncu = effective_cus(h->num_cus, s);   // the stream's CU mask counts (co-residency of the recurrence's clusters)
    const int rbase = P_LAYER0 + PL_COUNT * d.L;
    // (without an RNN the tail of the state dict is linear.weight [S, D], linear.bias: :43-46)
    const int g_lin_w = rbase + (d.with_rnn ? (int)PR_LIN_W : 0), g_lin_b = rbase + (d.with_rnn ? (int)PR_LIN_B : 1);
    const int Kout = d.with_rnn ? d.R : d.D;
    const float* head_in = d.with_rnn ? W + L.hall : (d.L ? W + L.layers[d.L - 1].xo : W + L.x0);
    float* part = X + S.part;
    float* colpart = X + S.colpart;

    // paper configuration: the feed-forward half of every layer's backward runs as one fused kernel per window
    static int use_fbwd = -1;   // TIP_TRAIN_FUSED_BWD=0: layer-by-layer (measurement)
    if (use_fbwd < 0) use_fbwd = (tip_env("TIP_TRAIN_FUSED_BWD") && tip_env("TIP_TRAIN_FUSED_BWD")[0] == '0') ? 0 : 1;
    const bool fbwd = use_fbwd && fused_supported(d, T) && fused_bwd_image_floats(d) > 0;
    if (fbwd) {
        std::vector<PackOp> ops;
        fused_bwd_pack_ops(d, params, 0, ops);
        TT(run_pack_ops(ops, X + S.bwimg, s), "bwd_pack");
    }
    // fused backward: the per-window LayerNorm / bias partials of every layer AND the three batch-wide bias partials (output
    // projection, recurrence, in_linear) stay in place and are reduced by ONE launch after the loop
    const bool col_multi = fbwd && 2 * d.L + 3 <= kColMulti;
    ColMulti cm;
    cm.Z = B;
    for (int i = 0; i < kColMulti; ++i) cm.src[i] = ColSrc{nullptr, nullptr, nullptr, nullptr, 0, 0, 0};
    // ---- output projection (:102): y = h W_out^T + b ---------------------------------------------------------------
    {
        ColSide pad;      // dy [M][S] -> dyp [M][S padded to 16] rides in the bias-gradient launch
        pad.kind = 1; pad.src = dy; pad.dst = X + S.dyp; pad.n = d.S; pad.npad = Sp; pad.total = (long long)M * Sp;
        TT(colsum(dy, d.S, M, d.S, col_multi ? X + S.colpart_out : colpart, grads + goff[g_lin_b], nullptr, s,
                  col_multi ? &cm.src[2 * d.L] : nullptr, pad), "bwd_db_out");
    }
    TT(grad_weight(X + S.dyp, Sp, Sp, d.S, head_in, Kout, Kout, M, part, S.part_floats, grads + goff[g_lin_w], ncu, s),
       "bwd_dW_out");
    static const bool win_k = !(tip_env("TIP_TRAIN_WIN_GEMM") && tip_env("TIP_TRAIN_WIN_GEMM")[0] == '0');   // TIP_TRAIN_WIN_GEMM=0: measurement
    const bool win_g = win_k && fbwd && train_fused_prep(d, T) && L.wout_tf && T <= 40 && Sp <= 160 && Sp % 4 == 0;
    float* gx = X + S.ga;     // gradient w.r.t. the current layer's output
    float* galt = X + S.gb;
    if (!d.with_rnn) {
        // no RNN: the projection's input gradient IS the gradient of the encoder output
        TG g = tg_base(X + S.dyp, Sp, W + L.wout_t, Sp, gx, d.D, M, d.D, Sp);
        TT(tgemm16_launch(g, s), "bwd_d_enc");
    } else {
    {
        hipError_t he = hipErrorInvalidValue;
        if (win_g) {
            WinGemmArgs wa{X + S.dyp, Sp, Sp, W + L.wout_tf, (int)((size_t)d.R * 160 * 4), X + S.dh, d.R};
            he = launch_win_gemm(d.R, 160, wa, B, T, ncu, s);
        }
        if (he == hipErrorInvalidValue) {
            TG g = tg_base(X + S.dyp, Sp, W + L.wout_t, Sp, X + S.dh, d.R, M, d.R, Sp);
            he = tgemm16_launch(g, s);
        }
        TT(he, "bwd_dH");
    }
    // ---- recurrence (:98-99), time reversed: delta_t = (dH_t + delta_{t+1} W_hh) * (1 - h_t^2) -------------------------
    TT(launch_rnn_bwd(d, X + S.dh, W + L.whh_b, W + L.hall, X + S.delta, reinterpret_cast<unsigned*>(const_cast<float*>(W + L.flags)),
                      B, T, auto_cluster(d, B, ncu, h->demoted != 0), ncu, h->guard(), s), "bwd_rnn");
    {
        ColSide sh;       // hprev[b,t] = h[b,t-1] (dW_hh's operand) rides in the recurrence-bias launch
        sh.kind = 2; sh.src = W + L.hall; sh.dst = X + S.hprev; sh.T = T; sh.R4 = d.R / 4; sh.total = (long long)M * d.R / 4;
        TT(colsum(X + S.delta, d.R, M, d.R, col_multi ? X + S.colpart_rnn : colpart, grads + goff[rbase + PR_BIH], grads + goff[rbase + PR_BHH],
                  s, col_multi ? &cm.src[2 * d.L + 1] : nullptr, sh), "bwd_db_rnn");
    }
    const float* enc = d.L ? W + L.layers[d.L - 1].xo : W + L.x0;
    {
        // dW_hh = delta^T h_prev and dW_ih = delta^T x share delta and the row range: one launch
        const DwProb pr[2] = {{X + S.delta, d.R, d.R, X + S.hprev, d.R, d.R, grads + goff[rbase + PR_WHH]},
                              {X + S.delta, d.R, d.R, enc, d.D, d.D, grads + goff[rbase + PR_WIH]}};
        hipError_t be = hipSuccess;
        // (the big-tile variant of the single launch serves the large configurations better than a two-problem batch)
        const bool big = (d.R / 128) * (d.D / 128) >= ncu;
        if (!big && grad_weight_batch(pr, 2, M, part, S.part_floats, ncu, s, &be)) {
            TT(be, "bwd_dW_rnn");
        } else {
            TT(grad_weight(pr[0].dY, d.R, d.R, d.R, pr[0].X, d.R, d.R, M, part, S.part_floats, pr[0].out, ncu, s), "bwd_dW_hh");
            TT(grad_weight(pr[1].dY, d.R, d.R, d.R, pr[1].X, d.D, d.D, M, part, S.part_floats, pr[1].out, ncu, s), "bwd_dW_ih");
        }
    }
    {
        hipError_t he = hipErrorInvalidValue;
        if (win_g) {
            WinGemmArgs wa{X + S.delta, d.R, d.R, W + L.wih_tf, (int)((size_t)d.D * d.R * 4), gx, d.D};
            he = launch_win_gemm(d.D, d.R, wa, B, T, ncu, s);
        }
        if (he == hipErrorInvalidValue) {
            TG g = tg_base(X + S.delta, d.R, W + L.wih_t, d.R, gx, d.D, M, d.D, d.R);
            he = lin_launch(g, L.wih_tf ? W + L.wih_tf : nullptr, s);
        }
        TT(he, "bwd_d_enc");
    }
    }   // with_rnn
    // ---- encoder layers, last to first -----------------------------------------------------------------------------------
    // fused backward: the per-window LayerNorm / bias partials of every layer stay in place and are reduced by ONE launch after
    // the loop (two small launches per layer otherwise)
    for (int l = d.L - 1; l >= 0; --l) {
        const int pb = P_LAYER0 + PL_COUNT * l;
        const float* const* lp = params + pb;
        const TrainLayer& t = L.layers[l];
        const float* x_in = l ? W + L.layers[l - 1].xo : W + L.x0;
        const int nln = (M + kLnRows - 1) / kLnRows;
        if (fbwd) {
            // LayerNorm2 backward -> dff2 (gc) -> dpre (gbig) -> dx1 (gx, in place), fused per window
            FfnBwdArgs fa;
            fa.wimg = X + S.bwimg; fa.wbytes = (int)(fused_bwd_image_floats(d) * 4); fa.layer = l;
            fa.dy = gx; fa.z2 = W + t.z2; fa.st2 = W + t.st2; fa.g2 = lp[PL_N2_W]; fa.hid = W + t.hid;
            fa.gbits = train_forward_is_fused(d, T) ? reinterpret_cast<const unsigned*>(W + t.hid + (size_t)M * d.F) : nullptr;
            fa.gate_scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.f;
            float* lnw_f = X + S.lnwin + (size_t)l * B * (9 * d.D + d.F);        // this layer's FFN-half partials
            float* lnw_a = lnw_f + (size_t)B * (3 * d.D + d.F);                   // ... and attention-half partials
            fa.dff2 = X + S.gc; fa.dpre = X + S.gbig; fa.dx1 = gx; fa.lnpart = lnw_f;
            const Drop dr = make_drop(p_drop, seed, (unsigned)(l * 4 + 3));
            fa.seed = dr.seed; fa.site = dr.site; fa.thresh = dr.thresh; fa.scale = dr.scale;
            TT(launch_ffn_bwd(d, fa, B, T, ncu, s), "bwd_ffn_fused");
            // per-window partials [dgamma2 | dbeta2 | d(linear2 bias) | d(linear1 bias)] -> the four gradient tensors
            if (!col_multi) {
                hipLaunchKernelGGL(colreduce_kernel, dim3((3 * d.D + d.F + 63) / 64), dim3(256), 0, s, lnw_f, B, 3 * d.D + d.F,
                                   grads + goff[pb + PL_N2_W], nullptr, 2 * d.D, grads + goff[pb + PL_L2_B], 3 * d.D,
                                   grads + goff[pb + PL_L1_B]);
                TT(hipGetLastError(), "bwd_ln2_params");
            } else {
                cm.src[2 * l] = ColSrc{lnw_f, grads + goff[pb + PL_N2_W], grads + goff[pb + PL_L2_B], grads + goff[pb + PL_L1_B],
                                       3 * d.D + d.F, 2 * d.D, 3 * d.D};
                cm.src[2 * l + 1] = ColSrc{lnw_a, grads + goff[pb + PL_N1_W], grads + goff[pb + PL_OUT_B], grads + goff[pb + PL_QKV_B],
                                           6 * d.D, 2 * d.D, 3 * d.D};
            }
            // dW2 / dW1 wait for the attention half: the layer's four weight gradients go out in one launch below
        } else {
        // LN2: gx -> dz2 (galt), dff2 = dz2 * keep3 (gc)
        {
            const Drop dr = make_drop(p_drop, seed, (unsigned)(l * 4 + 3));
            const float* z = W + t.z2;
            const float* st = W + t.st2;
            float* dzm = X + S.gc;
            TT(ln_dispatch(d.D, [&](auto nv) {
                   hipLaunchKernelGGL((tln_bwd_kernel<decltype(nv)::value>), dim3(nln), dim3(256), 0, s, gx, z, st, lp[PL_N2_W], galt,
                                      dzm, dr, colpart, M);
                   return hipGetLastError();
               }), "bwd_ln2");
            // norm weight and bias are consecutive tensors of D floats each: [dgamma | dbeta] land there in one go, the third
            // section (column sums of the masked gradient) is linear2's bias gradient
            hipLaunchKernelGGL(colreduce_kernel, dim3((3 * d.D + 63) / 64), dim3(256), 0, s, colpart, nln, 3 * d.D,
                               grads + goff[pb + PL_N2_W], nullptr, 2 * d.D, grads + goff[pb + PL_L2_B]);
            TT(hipGetLastError(), "bwd_ln2_params");
        }
        TT(grad_weight(X + S.gc, d.D, d.D, d.D, W + t.hid, d.F, d.F, M, part, S.part_floats, grads + goff[pb + PL_L2_W], ncu, s),
           "bwd_dW2");
        {
            // d(pre-ReLU) = (dff2 W2) * [hid > 0] / (1 - p)   (hid is saved AFTER ReLU and dropout)
            TG g = tg_base(X + S.gc, d.D, W + t.w2_t, d.D, X + S.gbig, d.F, M, d.F, d.D);
            g.gate = W + t.hid; g.ldgate = d.F;
            g.gate_scale = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.f;
            TT(lin_launch(g, t.w2_tf ? W + t.w2_tf : nullptr, s), "bwd_dhid");
        }
        TT(colsum(X + S.gbig, d.F, M, d.F, colpart, grads + goff[pb + PL_L1_B], nullptr, s), "bwd_db1");
        TT(grad_weight(X + S.gbig, d.F, d.F, d.F, W + t.x1, d.D, d.D, M, part, S.part_floats, grads + goff[pb + PL_L1_W], ncu, s),
           "bwd_dW1");
        {
            // dx1 = dz2 + dpre W1   -> gx
            TG g = tg_base(X + S.gbig, d.F, W + t.w1_t, d.F, gx, d.D, M, d.D, d.F);
            g.res = galt; g.ldres = d.D;
            TT(lin_launch(g, t.w1_tf ? W + t.w1_tf : nullptr, s), "bwd_dx1");
        }
        }
        if (fbwd) {
            // LayerNorm1 backward -> datt_o (gc) -> per-head dO -> attention backward -> dqkv (gbig) -> dx_in (gx, in place)
            AttnBwdArgs aa;
            aa.wimg = X + S.bwimg; aa.wbytes = (int)(fused_bwd_image_floats(d) * 4); aa.layer = l;
            aa.dx1 = gx; aa.z1 = W + t.z1; aa.st1 = W + t.st1; aa.g1 = lp[PL_N1_W];
            aa.qkv = W + t.qkv; aa.att = W + t.att; aa.ast = W + t.ast; aa.q_scale = 1.0f / sqrtf((float)d.dh);
            float* lnw_a = X + S.lnwin + (size_t)l * B * (9 * d.D + d.F) + (size_t)B * (3 * d.D + d.F);
            aa.dz1 = galt; aa.datt_o = X + S.datt; aa.dqkv = X + S.gbig2; aa.dx_in = gx; aa.lnpart = lnw_a;
            const Drop dr = make_drop(p_drop, seed, 0);
            aa.seed = dr.seed; aa.site0 = (unsigned)(l * 4 + 0); aa.site1 = (unsigned)(l * 4 + 1); aa.thresh = dr.thresh; aa.scale = dr.scale;
            TT(launch_attn_bwd(d, aa, B, T, ncu, s), "bwd_attn_fused");
            // per-window partials [dgamma1 | dbeta1 | d(out_proj bias) | d(in_proj bias)] -> the four gradient tensors
            if (!col_multi) {
                hipLaunchKernelGGL(colreduce_kernel, dim3((6 * d.D + 63) / 64), dim3(256), 0, s, lnw_a, B, 6 * d.D,
                                   grads + goff[pb + PL_N1_W], nullptr, 2 * d.D, grads + goff[pb + PL_OUT_B], 3 * d.D,
                                   grads + goff[pb + PL_QKV_B]);
                TT(hipGetLastError(), "bwd_ln1_params");
            }
            const DwProb pr[4] = {{X + S.gc, d.D, d.D, W + t.hid, d.F, d.F, grads + goff[pb + PL_L2_W]},
                                  {X + S.gbig, d.F, d.F, W + t.x1, d.D, d.D, grads + goff[pb + PL_L1_W]},
                                  {X + S.datt, d.D, d.D, W + t.att, d.D, d.D, grads + goff[pb + PL_OUT_W]},
                                  {X + S.gbig2, 3 * d.D, 3 * d.D, x_in, d.D, d.D, grads + goff[pb + PL_QKV_W]}};
            hipError_t be = hipSuccess;
            if (grad_weight_batch(pr, 4, M, part, S.part_floats, ncu, s, &be)) {
                TT(be, "bwd_dW_layer");
            } else {
                for (int i = 0; i < 4; ++i)
                    TT(grad_weight(pr[i].dY, pr[i].ldy, pr[i].n_out, pr[i].n_out, pr[i].X, pr[i].ldx, pr[i].K, M, part, S.part_floats,
                                   pr[i].out, ncu, s), "bwd_dW");
            }
        } else {
        // LN1: gx -> dz1 (galt), datt_o = dz1 * keep1 (gc)
        {
            const Drop dr = make_drop(p_drop, seed, (unsigned)(l * 4 + 1));
            const float* z = W + t.z1;
            const float* st = W + t.st1;
            float* dzm = X + S.gc;
            TT(ln_dispatch(d.D, [&](auto nv) {
                   hipLaunchKernelGGL((tln_bwd_kernel<decltype(nv)::value>), dim3(nln), dim3(256), 0, s, gx, z, st, lp[PL_N1_W], galt,
                                      dzm, dr, colpart, M);
                   return hipGetLastError();
               }), "bwd_ln1");
            hipLaunchKernelGGL(colreduce_kernel, dim3((3 * d.D + 63) / 64), dim3(256), 0, s, colpart, nln, 3 * d.D,
                               grads + goff[pb + PL_N1_W], nullptr, 2 * d.D, grads + goff[pb + PL_OUT_B]);
            TT(hipGetLastError(), "bwd_ln1_params");
        }
        TT(grad_weight(X + S.gc, d.D, d.D, d.D, W + t.att, d.D, d.D, M, part, S.part_floats, grads + goff[pb + PL_OUT_W], ncu, s),
           "bwd_dWo");
        {
            TG g = tg_base(X + S.gc, d.D, W + t.wo_t, d.D, X + S.datt, d.D, M, d.D, d.D);
            TT(lin_launch(g, t.wo_tf ? W + t.wo_tf : nullptr, s), "bwd_datt");
        }
        {
            const Drop dr = make_drop(p_drop, seed, (unsigned)(l * 4 + 0));
            const float qs = 1.0f / sqrtf((float)d.dh);
            const float* qkv = W + t.qkv;
            const float* att = W + t.att;
            const float* ast = W + t.ast;
            const float* datt = X + S.datt;
            float* dqkv = X + S.gbig;
            const int H = d.H;
            TT(launch_mattn_bwd(qkv, att, ast, datt, dqkv, B, T, H, d.dh, qs, dr, s), "bwd_attention");
        }
        TT(colsum(X + S.gbig, 3 * d.D, M, 3 * d.D, colpart, grads + goff[pb + PL_QKV_B], nullptr, s), "bwd_dbqkv");
        TT(grad_weight(X + S.gbig, 3 * d.D, 3 * d.D, 3 * d.D, x_in, d.D, d.D, M, part, S.part_floats, grads + goff[pb + PL_QKV_W], ncu, s),
           "bwd_dWqkv");
        {
            // dx_in = dz1 + dqkv W_qkv   -> gx
            TG g = tg_base(X + S.gbig, 3 * d.D, W + t.wqkv_t, 3 * d.D, gx, d.D, M, d.D, 3 * d.D);
            g.res = galt; g.ldres = d.D;
            TT(lin_launch(g, t.qkv_tf ? W + t.qkv_tf : nullptr, s), "bwd_dx_in");
        }
        }
    }
    // ---- in_linear (:79) ---------------------------------------------------------------------------------------------------
    TT(colsum(gx, d.D, M, d.D, col_multi ? X + S.colpart_in : colpart, X + S.dbin_p, nullptr, s, col_multi ? &cm.src[2 * d.L + 2] : nullptr),
       "bwd_db_in");
    if (col_multi) {
        int gxm = (3 * d.D + d.F + 63) / 64;
        for (int v : {(6 * d.D + 63) / 64, (d.R + 63) / 64, (d.S + 63) / 64}) gxm = v > gxm ? v : gxm;
        hipLaunchKernelGGL(colreduce_multi_kernel, dim3(gxm, 2 * d.L + 3), dim3(256), 0, s, cm);
        TT(hipGetLastError(), "bwd_ln_params");
    }
    TT(grad_weight(gx, d.D, d.D, d.D, W + L.U, d.InPad, d.InPad, M, part, S.part_floats, X + S.dwin_p, ncu, s), "bwd_dW_in");
    hipLaunchKernelGGL(finish_in_kernel, dim3(grid_for((long long)d.D * d.In)), dim3(256), 0, s, X + S.dwin_p, X + S.dbin_p,
                       grads + goff[P_IN_W], grads + goff[P_IN_B], d.D, d.H, d.In, d.InPad, d.n_imu_total + d.rootv0,
                       d.n_imu_total + d.rootv1);
    TT(hipGetLastError(), "bwd_finish_in");
    return TIP_OK;
}

// Gradients w.r.t. the INPUTS of the step (simple_transformer_with_state.py:63-79 backwards): dU = dx0 W_in' with dx0 = the gradient
// w.r.t. in_linear's output that tip_train_backward leaves in its scratch and W_in' = in_linear's weight with the channel shuffle
// folded into its rows and the root-velocity columns (:75) zeroed; d x_imu = dU[:, :n_imu], d x_s = dU[:, n_imu:] * keep_mask *
// keep_scale, zero where x_s was NaN (:65 overwrites those entries).
int tip_train_input_grads(tip_handle* h, const float* const* params, int n_params, const float* x_s, const float* keep_mask,
                          float keep_scale, void* scratch, size_t scratch_bytes, float* dx_imu, float* dx_s, int B, int T, void* stream) {
    if (!h || !params || !x_s || !scratch || (!dx_imu && !dx_s)) return TIP_ERR_INVALID_ARG;
    if (n_params != (int)h->tensor_names.size()) return TIP_ERR_INVALID_ARG;
    const Dims& d = h->d;
    if (!train_supported(d, B, T)) return TIP_ERR_UNSUPPORTED_CONFIG;
    g_tgemm_cus = h->num_cus;
    const TrainScratch S = scratch_layout(d, B, T);
    if (reinterpret_cast<uintptr_t>(scratch) % 256 || scratch_bytes < S.total * sizeof(float)) return TIP_ERR_WORKSPACE;
    hipStream_t s = static_cast<hipStream_t>(stream);
    float* X = static_cast<float*>(scratch);
    const int M = B * T;
    // W_in' [D][InPad] into the (consumed) partial-dW_in area, dU [M][InPad] into the (free) wide gradient buffer
    hipLaunchKernelGGL(prep_in_kernel, dim3(grid_for((long long)d.D * d.InPad)), dim3(256), 0, s, params[P_IN_W], params[P_IN_B],
                       X + S.dwin_p, X + S.dbin_p, d.D, d.H, d.In, d.InPad, d.n_imu_total + d.rootv0, d.n_imu_total + d.rootv1);
    TT(hipGetLastError(), "input_grads_prep");
    TG g = tg_base(X + S.ga, d.D, X + S.dwin_p, d.InPad, X + S.gbig, d.InPad, M, d.InPad, d.D);
    TT((tgemm_launch<0, 1>(g, 1, s)), "input_grads_gemm");
    hipLaunchKernelGGL(input_grads_kernel, dim3(grid_for((long long)M * d.In)), dim3(256), 0, s, X + S.gbig, x_s, keep_mask,
                       keep_mask ? keep_scale : 1.f, dx_imu, dx_s, (long long)M, d.n_imu_total, d.S, d.InPad);
    TT(hipGetLastError(), "input_grads_split");
    return TIP_OK;
}

}  // extern "C"

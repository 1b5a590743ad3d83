// tip_internal.h — shared between the C-ABI translation unit and the kernel translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/tip_hip.h"
#include "../../include/tip_hip_debug.h"

namespace tip {

// Measurement switches.  The launchers' A/B selections, in-kernel traces and ablations are driven by TIP_* environment variables —
// in a build with -DTIP_MEASURE only (`make measure` -> libtip_hip_measure.so, loaded by the tools under tools/ through
// TIP_LIB=measure).  In the default build tip_env() is a constant nullptr: every switch sits at its default and the kernels a handle
// launches depend on tip_set_option alone, never on the process environment.  The switches are listed in include/tip_hip_debug.h.
inline const char* tip_env(const char* name) {
#ifdef TIP_MEASURE
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

// tanh for the recurrence: sign(x) * (1 - e) / (1 + e), e = exp(-2|x|) on the hardware exp2 path.  No overflow, abs error
// ~2e-7 (the accurate libm tanhf costs ~1 us per step on the serial chain).  Every RNN kernel uses this one function,
// so the variants stay bit-identical to each other.
__device__ __forceinline__ float tip_tanh(float x) {
    const float e = __expf(-2.0f * fabsf(x));
    return copysignf(__fdividef(1.0f - e, 1.0f + e), x);
}

// one dropout site of the training step (tip_train.hip): keep(idx) = hash(seed, site, idx) >= thresh; thresh 0 = off
struct Drop {
    unsigned long long seed;
    unsigned site;
    unsigned thresh;
    float scale;   // 1 / (1 - p)
    unsigned key;  // tip_drop_key(seed, site), filled in by make_drop on the host
};
typedef Drop AttnDrop;

// keep(seed, site, idx): the counter-based dropout hash of the training step (include/tip_hip.h documents it).
// key = hi32(splitmix64-mix(seed + GOLDEN * (site + 1))) is wave-uniform (scalar unit, once per site); per element only
// 32-bit arithmetic remains: the 64-bit splitmix per ELEMENT of the first version cost a dozen quarter-rate multiplies.
__host__ __device__ __forceinline__ unsigned tip_drop_key(unsigned long long seed, unsigned site) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * ((unsigned long long)site + 1ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned)(z >> 32);
}
// kernels compute the key ONCE (wave-uniform, forced into an SGPR) and hash elements with tip_drop_hash_k
__device__ __forceinline__ unsigned tip_drop_key_s(unsigned long long seed, unsigned site) {
    return (unsigned)__builtin_amdgcn_readfirstlane((int)tip_drop_key(seed, site));
}
__device__ __forceinline__ unsigned tip_drop_hash_k(unsigned key, unsigned idx) {   // idx: element index mod 2^32
    unsigned z = idx * 0x9E3779B1u + key;
    z ^= z >> 16;
    z *= 0x7FEB352Du;
    z ^= z >> 15;
    z *= 0x846CA68Bu;
    z ^= z >> 16;
    return z;
}
__device__ __forceinline__ unsigned tip_drop_hash(unsigned long long seed, unsigned site, unsigned long long idx) {
    return tip_drop_hash_k(tip_drop_key(seed, site), (unsigned)idx);
}

// Where the fused encoder, run as the TRAINING forward, stashes its activations (float offsets into `sv`; the per-layer
// arrays of layer l start `l * layer_stride` further on) and the encoder dropout it applies (site = 4*layer + k).
struct FusedTrain {
    float* sv;
    // float offsets into `sv` in units of 64 floats (every stash array is 64-float aligned): 32-bit, because this struct
    // lives in SGPRs for the whole kernel and the training forward is short of them
    unsigned x0, qkv, ast, att, z1, st1, x1, hid, z2, st2, xo, layer_stride;
    unsigned u = 0;      // 0: U (the staged input rows [M][224], dW_in's operand) is written by a prologue launch; else its offset: the hybrid kernel stashes it
    unsigned long long seed;
    unsigned thresh;   // 0 = dropout off
    float scale;
};

constexpr int kGemmBM = 128;   // general GEMM block tile (rows)
constexpr int kGemmBN = 128;   // general GEMM block tile (cols); packed weights are padded to this
constexpr int kGemmBK = 16;    // K tile; packed K is padded to this
constexpr int kRnnTile = 16;   // windows per RNN tile (= MFMA 16x16x4 row count)
constexpr int kRnnRows4 = 0x44;  // TIP_OPT_RNN_CLUSTER value: 4-window tiles on 4-workgroup clusters, 4x4x1 MFMAs (rnn_rows4_kernel)

inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

struct Dims {
    int n_imu_total;  // input_size_imu (+18)
    int S;            // size_s
    int In;           // n_imu_total + S
    int InPad;        // In rounded to kGemmBK
    int D, H, dh, F, L, R;
    int with_rnn;
    int t_max;
    int rootv0, rootv1;  // x_s columns zeroed at simple_transformer_with_state.py:75
    float q_scale;       // applied inside the attention kernel (1 when folded into W_q)
    int fold_q_scale;
};

// One GEMM operand in the packed image: W [Npad][Kpad] row-major (zero padded) + bias [Npad].
struct PackedLinear {
    size_t w_off;   // float offset into the packed image
    size_t b_off;
    int N, K, Npad, Kpad;
    size_t f_off = 0;   // the same weight in 16x16x4 B-fragment order (+ tail padding) for launch_pgemm; 0 = not packed
};

struct PackedLayer {
    PackedLinear qkv, out, ff1, ff2;
    size_t g1_off, be1_off, g2_off, be2_off;  // [D] each
};

struct PackedLayout {
    PackedLinear in_lin;
    std::vector<PackedLayer> layers;
    PackedLinear rnn_ih;       // bias = b_ih + b_hh
    size_t whh_frag_off;       // W_hh in MFMA 16x16x4 B-fragment order: [R/16 nb][R/16 kb][64 lanes][4]
    PackedLinear out_lin;
    size_t out_frag_off;       // out-linear weight in B-fragment order [ceil(S/16)][K/16][64][4] (zero padded rows)
    // fused-plan section (paper configuration): weights in 16x16x4 B-fragment order, see tip_fused.hip
    size_t fused_off;
    size_t fused_floats;
    size_t total_floats;
};

// One destination array of the packed image, as a function of one (or the sum of two) source tensors: used by the on-device
// packer (tip_pack.hip).  dst element (n, k), n < N, k < K:  src[map(n)][k] (+ src2) for n < src_rows, k < src_cols and k
// outside [z0, z1), times `scale` for n < scale_rows, else 0; stored row-major or in 16x16x4 B-fragment order.
struct PackOp {
    const float* src;
    const float* src2;
    unsigned long long dst_off;   // float offset into the image
    int N, K;
    int src_rows, src_cols;
    int frag;                     // 0: [N][K] row-major, 1: [N/16][K/16][64 lanes][4]
    int shuffle_h, shuffle_dh;    // 0 = identity, else dst row a*H + b <- src row b*dh + a (channel shuffle :88-89)
    int z0, z1;
    int scale_rows;
    float scale;
    int transpose;                // 1: the source is stored [K][N] (src_rows = N, src_cols = K describe the LOGICAL matrix)
};
constexpr int kPackBatch = 32;
struct PackBatch {
    PackOp ops[kPackBatch];
};

// Workspace carve-up for the general plan (float offsets), M = B*T rows.
struct Workspace {
    size_t flow, xa, xb, big, att, hall, flags, lat, xchg;  // float offsets (flow: ALWAYS 0 — the one-launch few-stream form's flags and launch
                                                            // counters must not move when B or T change: tip_latency.hip)
    size_t total_bytes;
};

struct StageTimer {
    std::string name;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs;  // one (start, stop) per recorded launch
    size_t used = 0;                                        // pairs recorded since the last reset
};
constexpr size_t kMaxTimerPairs = 1 << 16;

// What every kernel with an inter-workgroup hand-off is given: where to report a wait that gave up, and (tests only) which
// cooperating workgroup to drop so that a wait MUST give up.
//   err   device-visible pointer to the handle's pinned host word (null = no reporting): a give-up stores 1 there (system
//         scope); tip_forward / tip_train_* / tip_check return TIP_ERR_HANDOFF while it is set.  The kernel that gave up
//         also POISONS what it produces from then on (canonical NaN), so the launch's outputs are NaN, never finite-but-wrong.
//   fault TIP_OPT_FAULT_INJECT bits: 1 = pair-split encoder, workgroup (pair 0, half 1) exits at entry; 2 = clustered RNN,
//         member 1 of group 0 exits at entry; 4 = GEMV RNN (latency plan), member 1 of stream 0 exits at entry.  With any bit
//         set the spin bounds shrink (2^14 polls) so the give-up takes milliseconds, not seconds.
struct Guard {
    unsigned* err = nullptr;
    int fault = 0;
};
constexpr unsigned kPoisonBits = 0x7FC00000u;   // canonical quiet NaN (NOT the all-ones RNN sentinel)
__device__ __forceinline__ unsigned guard_spin_limit(int fault, unsigned normal) { return (fault & 7) ? (1u << 14) : normal; }   // (bit 3 drops no member: full waits)
__device__ __forceinline__ void guard_report(unsigned* err) {
    if (err) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Launch-side caches are PER DEVICE, not per process: a process may hold one handle per GPU (include/tip_hip.h), and a
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) or an occupancy answer obtained on one device says nothing about the next.
constexpr int kMaxDevices = 64;
inline int tip_cur_device() {
    int dv = 0;
    if (hipGetDevice(&dv) != hipSuccess || dv < 0 || dv >= kMaxDevices) dv = 0;
    return dv;
}
struct PerDeviceFlag {
    bool v[kMaxDevices] = {};
    bool& cur() { return v[tip_cur_device()]; }
};
struct PerDeviceInt {
    int v[kMaxDevices];
    PerDeviceInt() { for (int& x : v) x = -1; }
    int& cur() { return v[tip_cur_device()]; }
};

// Forwards of different HIP streams are SERIALISED on the device.  The default plans launch cooperating kernels (RNN clusters,
// the latency plan's GEMV cluster, the opt-in pair-split encoder) whose members must all be resident at once; check_coresident
// checks one launch against an EMPTY GPU, so two forwards in flight on two streams could each hold half of the CUs and starve
// the other's partners (a ~1 s spin, NaN poison, TIP_ERR_HANDOFF).  Every entry point that launches such kernels brackets its
// launches with this guard: the first time a second stream shows up on a device an event is recorded behind the previous
// stream's work and the new stream waits on it; from then on every forward records the event and a stream switch waits on it
// (one hipEventRecord per forward, only in processes that really use several streams).  Streams are told apart by their handle
// value: a destroyed stream whose handle the runtime recycles counts as the same stream.  A forward fills the GPU by itself: nothing is lost by not overlapping two of them.
// A stream that is being CAPTURED into a HIP graph (hipStreamBeginCapture: the streaming engine's graph mode, or any caller's) is
// left alone: an event recorded inside a capture cannot order work outside it.  A captured forward is therefore not serialised
// against other streams by the library — replay the graph on the stream the other forwards use, or when none is in flight.
inline bool stream_is_capturing(hipStream_t s) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return cs != hipStreamCaptureStatusNone;
}

struct CoopSerial {
    int dev;
    hipStream_t stream;
    bool capturing;
    hipError_t status;      // hipSuccess, or what the wait / drain returned (the caller reports it)
    CoopSerial(int device, hipStream_t s);   // device = the handle's (tip_handle::device), not the thread's current one
    ~CoopSerial();
    CoopSerial(const CoopSerial&) = delete;
    CoopSerial& operator=(const CoopSerial&) = delete;
};

// CUs the launches of a stream can actually use: the device's count, or the population of the stream's CU mask
// (hipExtStreamCreateWithCUMask — the co-tenant / partitioned-GPU case).  Plan selection (rounds of #CUs windows), grid sizes and
// above all the co-residency bound of the cooperating kernels (clusters <= CUs / members) are computed from THIS number.
inline int effective_cus(int device_cus, hipStream_t s) {
    if (stream_is_capturing(s)) return device_cus;   // no runtime queries inside a capture; a graph is sized for the whole device
    uint32_t mask[32] = {0};
    if (hipExtStreamGetCUMask(s, 32, mask) != hipSuccess) {
        (void)hipGetLastError();
        return device_cus;
    }
    int c = 0;
    for (int i = 0; i < device_cus && i < 32 * 32; ++i) c += (mask[i >> 5] >> (i & 31)) & 1u;
    return (c > 0 && c < device_cus) ? c : device_cus;
}

// cooperating kernels need `grid` workgroups resident at once: checked against the runtime's own occupancy answer for THIS
// kernel / block size / dynamic LDS (what hipLaunchCooperativeKernel checks, without its 15-19 us per launch); cached.
template <typename K>
inline hipError_t check_coresident(K kernel, int threads, size_t smem, int grid, int num_cus, int* cache) {
    if (*cache < 0) {
        int nb = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, smem);
        if (e != hipSuccess) return e;
        *cache = nb;
    }
    return (long long)grid <= (long long)*cache * num_cus ? hipSuccess : hipErrorCooperativeLaunchTooLarge;
}

}  // namespace tip

struct tip_handle {
    tip_config cfg;
    tip::Dims d;
    tip::PackedLayout lay;
    std::vector<std::string> tensor_names;
    std::vector<std::pair<int, int>> tensor_shapes;  // rows, cols (cols = 0 for 1-D)
    const float* packed_dev = nullptr;
    int device = -1;
    int num_cus = 256;
    int plan = TIP_PLAN_AUTO;
    int profile = 0;
    int rnn_cluster = 0;
    uint64_t forward_count = 0;
    std::string last_hip_error;
    std::vector<tip::StageTimer> timers;
    int cur_timer = -1;
    unsigned* err_host = nullptr;   // 64-byte pinned, device-mapped host block: word 0 = "a hand-off wait gave up" (sticky); word 1 = "the
                                    // one-launch few-stream form found a producer on another XCD" (sticky; TIP_OPT_HANDOFF_KIND tells them apart)
    unsigned* err_dev = nullptr;    // the device's address of it
    int fault_inject = 0;           // TIP_OPT_FAULT_INJECT (tests)
    int fuse_head = 0;              // TIP_OPT_FUSE_HEAD
    int auto_demote = 1;            // TIP_OPT_AUTO_DEMOTE
    int f1s_parts = 0;              // TIP_OPT_F1S_PARTS: 0 = auto, 2, 4
    unsigned long long flow_epoch = 0;   // launches of lat_flow_kernel so far (+ a per-handle base): stamps the completion flags of a launch
    int demoted = 0;                // TIP_OPT_DEMOTED: set by tip_demote after a lost hand-off: AUTO then avoids every cooperating kernel
    int no_flow = 0;                // TIP_OPT_NO_FLOW: the few-stream plan never takes its one-launch form (launch chain for every B)
    tip::Guard guard() const { return tip::Guard{err_dev, fault_inject}; }
};

namespace tip {

// ---- launchers implemented in tip_general.hip (all asynchronous on `s`) ----
hipError_t launch_prologue(const Dims& d, const float* x_imu, const float* x_s, const float* keep_mask,
                           float keep_scale, float* U, int M, hipStream_t s);
// C[M,N] = epi(A[M,K(lda)] * W[Npad,Kpad]^T + bias (+ res)); flags: 1 = relu, 2 = residual
hipError_t launch_gemm(const float* A, int lda, const float* W, int Kpad, const float* bias, const float* res,
                       int ldres, float* C, int ldc, int M, int N, int Npad, int flags, hipStream_t s);
// same contract on the 16x16x4 / 64x64-tile kernel of tip_train.hip (K = reduction length, multiple of 4)
hipError_t launch_gemm16(const float* A, int lda, const float* W, int ldw, const float* bias, const float* res, int ldres,
                         float* C, int ldc, int M, int N, int K, int flags, hipStream_t s);
hipError_t launch_attention(const Dims& d, const float* qkv, float* out, int B, int T, hipStream_t s);
// matrix-core attention for any configuration (tip_attn.hip): out [M,D] = softmax(causal(q k^T * q_scale)) v per head;
// ast (nullable) receives (row max, 1 / row sum) per (window, head, query) for the training backward; drop = dropout on P
bool mattn_supported(int dh, int T);
hipError_t launch_mattn_fwd(const float* qkv, float* out, float* ast, int B, int T, int H, int dh, float q_scale, AttnDrop drop,
                            hipStream_t s);
hipError_t launch_mattn_bwd(const float* qkv, const float* o_saved, const float* ast, const float* d_o, float* dqkv, int B, int T,
                            int H, int dh, float q_scale, AttnDrop drop, hipStream_t s);
hipError_t launch_layernorm(float* x, const float* g, const float* b, int M, int D, hipStream_t s);
hipError_t launch_rnn(const Dims& d, const float* ih, const float* whh_frag, float* hall, unsigned* flags, int B,
                      int T, int cluster, int num_cus, bool hall_armed, const Guard& gd, hipStream_t s);
// one process-wide launch tag for every kernel family that tags the XCC-exchange words of a workspace (tip_general.hip)
unsigned next_rnn_launch_tag();
size_t rnn_flag_words(int B, int T);
// training step, backward recurrence: delta_t = (dH_t + delta_{t+1} W_hh) * (1 - h_t^2) (tip_train.hip)
hipError_t launch_rnn_bwd(const Dims& d, const float* dH, const float* whh_t_frag, const float* h_fwd, float* delta,
                          unsigned* flags, int B, int T, int cluster, int num_cus, const Guard& gd, hipStream_t s);

// ---- panel GEMM with fragment-ordered weights for big linears (tip_fused2.hip) ----
bool pgemm_shape_ok(int M, int N, int K);   // N % 512 == 0, K % 128 == 0, M >= 320
// wfrag: W [N][K] in 16x16x4 B-fragment order [N/16][K/16][64][4], followed by >= 2 KiB of readable padding
hipError_t launch_pgemm(const float* A, int lda, const float* wfrag, size_t wfrag_floats, const float* bias, const float* res, int ldres,
                        float* C, int ldc, int M, int N, int K, int flags, hipStream_t s);

// ---- on-device packing (tip_pack.hip) ----
hipError_t run_pack_ops(const std::vector<PackOp>& ops, float* img, hipStream_t s);

// ---- fused plan (tip_fused.hip) ----
// descriptors of the fused section for the on-device packer (t = 56 device pointers, base = float offset of the section)
void fused_pack_ops(const Dims& d, const float* const* t, size_t base, std::vector<PackOp>& ops);
bool fused_supported(const Dims& d, int T);
size_t fused_packed_floats(const Dims& d);
void fused_pack(const Dims& d, const float* const* tensors, float* dst);
bool fused_has_rnn_ih(const Dims& d);
// xout [B,T,D] and/or ih_out [B,T,R] (RNN input projection incl. b_ih+b_hh); either may be null
hipError_t launch_fused_encoder(const Dims& d, const float* fused_w, const float* x_imu, const float* x_s,
                                const float* keep_mask, float keep_scale, float* xout, float* ih_out, float* hall_sentinel,
                                int B, int T, int num_cus, hipStream_t s);
// hybrid row tiling (rows 0-31 on 16x16x4, rows 32-39 on 4x4x1 MFMAs: no padded rows outside the QKV projection); same contract
hipError_t launch_fused_encoder_h(const Dims& d, const float* fused_w, const float* x_imu, const float* x_s,
                                  const float* keep_mask, float keep_scale, float* xout, float* ih_out, float* hall_sentinel,
                                  int B, int T, int num_cus, hipStream_t s);
// fused backward of one encoder layer's feed-forward block for the training step (tip_fused.hip): LayerNorm2 backward ->
// d(hidden) -> d(LN1 output), one workgroup per window; weight fragments of W2^T / W1^T from the backward image
size_t fused_bwd_image_floats(const Dims& d);
void fused_bwd_pack_ops(const Dims& d, const float* const* t, size_t base, std::vector<PackOp>& ops);
struct FfnBwdArgs {
    const float* wimg;      // backward image
    int wbytes, layer;
    const float* dy;        // [M,D] gradient w.r.t. the layer output
    const float* z2;        // [M,D] pre-LayerNorm2 rows, st2 [M,2]
    const float* st2;
    const float* g2;        // LayerNorm2 weight
    const float* hid;       // [M,F] saved hidden (after ReLU and dropout)
    const unsigned* gbits;  // [M][32 dwords] the same rows' gates (hidden > 0) as bits, written by the fused training forward
                            // (tip_fused.hip: hidden_to_hbm), or null: the kernel then reads the gates from `hid`
    int hid_bytes;          // (set by the launcher)
    int trace;              // (set by the launcher) TIP_BWD_TRACE=1: phase stamps of workgroup 0 (measurement)
    float gate_scale;       // 1 / (1 - p)
    float* dff2;            // out [M,D]: gradient into linear2's output (dropout mask applied)
    float* dpre;            // out [M,F]: gradient into linear1's pre-activation
    float* dx1;             // out [M,D]: gradient w.r.t. the LayerNorm1 output
    float* lnpart;          // out [B][3*D + F]: per-window (dgamma2 | dbeta2 | d bias of linear2 | d bias of linear1) partial sums
    unsigned long long seed;
    unsigned site, thresh;
    float scale;
};
hipError_t launch_ffn_bwd(const Dims& d, const FfnBwdArgs& a, int B, int T, int num_cus, hipStream_t s);
// ... and of its attention block: LayerNorm1 backward -> d(attention output) -> attention backward -> d(layer input)
struct AttnBwdArgs {
    const float* wimg;
    int wbytes, layer;
    const float* dx1;       // [M,D] gradient w.r.t. the LayerNorm1 output
    const float* z1;        // [M,D] pre-LayerNorm1 rows, st1 [M,2]
    const float* st1;
    const float* g1;
    const float* qkv;       // [M,3D] saved in-projection output (raw q | k | v)
    const float* att;       // [M,D] saved attention output
    const float* ast;       // [B,H,T,2] softmax statistics
    float q_scale;
    float* dz1;             // out [M,D]: gradient w.r.t. the pre-LayerNorm1 sum (scratch; also the residual path)
    float* datt_o;          // out [M,D]: gradient into out_proj's output (dropout mask applied)
    float* dqkv;            // out [M,3D]
    int dqkv_bytes;         // (set by the launcher)
    int trace;              // (set by the launcher) TIP_BWD_TRACE=1: phase stamps of workgroup 0 (measurement)
    float* dx_in;           // out [M,D]: gradient w.r.t. the layer input (may alias dx1)
    float* lnpart;          // out [B][6*D]: per-window (dgamma1 | dbeta1 | d bias of out_proj | d bias of in_proj)
    unsigned long long seed;
    unsigned site0, site1, thresh;   // dropout sites: attention probabilities, after out_proj
    float scale;
};
hipError_t launch_attn_bwd(const Dims& d, const AttnBwdArgs& a, int B, int T, int num_cus, hipStream_t s);
// per-window dX products of the training backward outside the encoder layers (win_gemm_kernel, tip_fused.hip)
struct WinGemmArgs {
    const float* A;       // [B * T][lda]
    int lda, kvalid;      // row stride (floats, multiple of 4); valid columns (multiple of 4): the rest of K reads as zero
    const float* wfrag;   // W' [N][K] in 16x16x4 B-fragment order, zero padded
    int wbytes;
    float* out;           // [B * T][ldo]
    int ldo;
};
hipError_t launch_win_gemm(int N, int K, const WinGemmArgs& a, int B, int T, int num_cus, hipStream_t s);
// the same kernel as the training forward: activations stashed per `tr`, encoder dropout live (tip_train.hip)
hipError_t launch_fused_train(const Dims& d, const float* fused_w, const float* x_imu, const float* x_s, const float* keep_mask,
                              float keep_scale, float* ih_out, float* hall_sentinel, const FusedTrain& tr, int B, int T,
                              int num_cus, hipStream_t s);
// ... on the hybrid row tiling (rows 0-31 on 16x16x4, rows 32-39 on 4x4x1 MFMAs: no matrix-core work on the pad rows)
hipError_t launch_fused_train_h(const Dims& d, const float* fused_w, const float* x_imu, const float* x_s, const float* keep_mask,
                                float keep_scale, float* ih_out, float* hall_sentinel, const FusedTrain& tr, int B, int T,
                                int num_cus, hipStream_t s);
// true when launch_rnn(cluster) will run the sentinel-polling resident kernel (HALL must be pre-filled with all-ones;
// the fused encoder can do that for its own rows, otherwise launch_rnn memsets)
bool rnn_uses_sentinel(const Dims& d, int B, int T, int cluster);
// Y[M,N] = A[M,K(lda)] * Wfrag^T + bias with Wfrag in 16x16x4 B-fragment order [ceil(N/16)][K/16][64][4]
hipError_t launch_head_gemm(const float* A, long long lda, const float* wfrag, const float* bias, float* Y, int ldy,
                            int M, int N, int K, hipStream_t s);

// the same product with the weight resident in registers (tip_head.hip): K = 512, 128 < N <= 144, window length a multiple of 40
// (full output: M a multiple of 40; last_only: M = one row per window, bit-identical to row T-1 of the full output).
// hipErrorInvalidValue = shape not served (nothing launched).
hipError_t launch_head_ksplit(const float* A, long long lda, const float* wfrag, const float* bias, float* Y, int ldy, int M, int N,
                              int K, bool last_only, int num_cus, hipStream_t s);

// ---- latency plan (tip_latency.hip): one window spread over many CUs, for few concurrent streams ----
bool latency_supported(const Dims& d, int B, int T);
size_t latency_workspace_floats(int B, int T);
// td (nullable): the training step's encoder dropout (tip_forward_dropout) — seed / thresh / scale as make_drop (tip_train.hip) derives them
struct TrainDropout {
    unsigned long long seed;
    unsigned thresh;
    float scale;
    unsigned mkey = 0, mthresh = 0;   // past-state keep mask drawn in the kernel (mthresh 0: not): see tip_draw_keep_mask
};
TrainDropout make_train_dropout(float p, unsigned long long seed);
constexpr unsigned kStateMaskSite = 0xFFFFFFF0u;   // hash site of the past-state keep mask (encoder sites are 4 * layer + k)
// (key, thresh) of a keep mask with drop probability p under `seed`; false for p outside [0, 1)
bool state_mask_params(float p, unsigned long long seed, unsigned* key, unsigned* thresh);
hipError_t launch_keep_mask(float* mask, size_t n, unsigned key, unsigned thresh, hipStream_t s);
// What the ONE-launch form of the plan (lat_flow_kernel, few windows: stages, recurrence and output projection as roles of one
// launch) needs beyond the chain's arguments; null = launch chain.  *done says whether the projection ran inside (then the caller
// launches no launch_latency_head).
struct LatencyHead {
    const float* wfrag; const float* bias; float* y; int ldy; int N; bool last_only;
    unsigned long long nonce;   // the handle's stamp base (tip_handle::flow_epoch)
    bool* done;
    unsigned long long* flags;  // the workspace's flag / launch-counter area (Workspace::flow; latency_flow_flag_floats() floats)
};
size_t latency_flow_flag_floats();
hipError_t launch_latency_plan(const Dims& d, const float* fused_w, const float* whh_frag, const float* x_imu,
                               const float* x_s, const float* keep_mask, float keep_scale, float* ws, float* hall, int B,
                               int T, int num_cus, const Guard& gd, hipStream_t s, const TrainDropout* td = nullptr,
                               const LatencyHead* head = nullptr);

hipError_t launch_latency_head(const float* A, long long lda, const float* wfrag, const float* bias, float* Y, int ldy,
                               int M, int N, hipStream_t s);

hipError_t read_spin_timeouts_general(unsigned* out);
hipError_t read_spin_timeouts_latency(unsigned* out);
hipError_t read_spin_timeouts_fused2(unsigned* out);

// ---- two-window fused encoder (tip_fused2.hip): 80 rows = 5 MFMA row blocks, no padding; for >= 2 windows per CU ----
bool fused2_supported(const Dims& d, int T);
// partial-sum exchange images of the window-split forms below (sized for the larger of the two)
size_t fused2s_xchg_floats(int B);
// reuse_cache != null: the exact-streaming-reuse form (tip_forward_reuse) — layer 0 takes its in_linear rows and
// Q / K / V rows from the per-stream ring that launch_reuse_update maintains instead of computing them (x_imu / x_s are not read);
// the frame index comes from *frame_ctr (device) when given, else from frame_idx
hipError_t launch_fused_encoder2(const Dims& d, const float* fused_w, const float* x_imu, const float* x_s,
                                 const float* keep_mask, float keep_scale, float* ih_out, float* hall_sentinel, int B,
                                 int num_cus, hipStream_t s, const float* reuse_cache = nullptr, const int* frame_ctr = nullptr,
                                 int frame_idx = 0);
// ring of the reuse form: 64 floats of header (one frame tag per slot) + [B][40 slots][x 256 | q 256 | k 256 | v 256]
size_t reuse_cache_floats(int B);
// row T - 1 of every window -> slot (frame mod 40) of its stream's ring, tag <- frame
hipError_t launch_reuse_update(const Dims& d, const float* fused_w, const float* x_imu, const float* x_s, float* cache, const int* frame_ctr,
                               int frame_idx, int B, int T, hipStream_t s);
// window-split form: ONE window on two co-resident workgroups (columns split, 48 rows each), for 2 B <= #CUs, B <= 128
bool fused1s_fits(int B, int num_cus);
// ... or on FOUR (round 4: quads of heads, quarters of the hidden units and of the RNN input projection), for 4 B <= #CUs, B <= 64
bool fused1s_quad_fits(int B, int num_cus);
// parts: 2, 4, or 0 = four when they fit
hipError_t launch_fused_encoder1s(const Dims& d, const float* fused_w, const float* x_imu, const float* x_s,
                                  const float* keep_mask, float keep_scale, float* ih_out, float* hall_sentinel, float* xchg,
                                  int B, int num_cus, int parts, const Guard& gd, hipStream_t s);

}  // namespace tip

/* tip_hip.h — C-ABI of the MI355X-native TIP forward pass (libtip_hip.so).
 *
 * The reference has no FFI layer: its boundary for this path is the Python nn.Module
 * `TF_RNN_Past_State` (/root/reference/simple_transformer_with_state.py:8-102).  Each entry point below
 * names the piece of that interface it replaces; the Python host in
 * transformer-inertial-poser_amd/simple_transformer_with_state.py binds them with ctypes.
 *
 * Conventions: plain C, no torch types, no exceptions; every function returns 0 or a negative
 * tip_status; device pointers are caller-owned, contiguous, row-major fp32; tip_forward is asynchronous on
 * the caller's HIP stream; one handle per GPU, used from one thread at a time; the library allocates no
 * device memory (packed weights and workspace are caller-provided buffers; the one allocation is a 64-byte pinned HOST
 * block per handle, the hand-off error word of tip_check).  Arithmetic is fp32; the reference's `--double` switch
 * (train_model.py:84-85) is served by tip_forward_f64 (fp64 parameters and windows, every operation in IEEE double) for the
 * forward and by tip_train_forward_f64 / tip_train_backward_f64 for the training step.  Tensors are never converted silently.
 */
#ifndef TIP_HIP_H
#define TIP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TIP_ABI_VERSION 5 /* 2: packed image without the exploratory split-fp16 section unless asked for (TIP_CREATE_S16, tip_create_ex);
                             tip_max_batch; export list = this header (+ tip_hip_debug.h), everything else hidden
                             3: tip_forward_dropout, tip_draw_keep_mask, tip_train_input_grads; plan 9 (persistent latency kernel) and its 1 KiB of sync words in the packed image
                             removed; TIP_OPT_FUSE_HEAD reserved
                             4: plans 5 / 7 / 8 (pair-split, split-fp16) and TIP_OPT_PACK_SPLIT16 retired (and the split-fp16 trace hook of tip_hip_debug.h with them)
                             5: exact streaming reuse — tip_reuse_cache_bytes, tip_reuse_reset, tip_forward_reuse, tip_stream_frame_counter_offset,
                                tip_stream_ingest_newest */

/* The library is built with -fvisibility=hidden: the functions declared here (and the measurement hooks of
 * tip_hip_debug.h) are its whole dynamic symbol table (tests/test_host_cpu.py compares `nm -D` with the two headers). */
#define TIP_API __attribute__((visibility("default")))

typedef struct tip_handle tip_handle;
typedef void* tip_stream_t; /* hipStream_t */

/* Constructor arguments of TF_RNN_Past_State (simple_transformer_with_state.py:9-17). */
typedef struct tip_config {
    int32_t input_size_imu;  /* 72 */
    int32_t size_s;          /* 131 */
    int32_t rnn_hid_size;    /* 512 */
    int32_t tf_hid_size;     /* 1024 (dim_feedforward) */
    int32_t tf_in_dim;       /* 256  (d_model) */
    int32_t n_heads;         /* 16 */
    int32_t tf_layers;       /* 4 */
    int32_t with_rnn;        /* :15 */
    int32_t with_acc_sum;    /* :16  (+18 input columns, :20-22) */
    int32_t t_max;           /* typical longest window (40; 80 for the scaled config): a sizing hint only — tip_forward serves any
                                T >= 1 (the reference builds its causal mask for any length, :56-58,85); T > 128 takes the key-tiled
                                attention of the general plan */
} tip_config;

typedef enum tip_status {
    TIP_OK = 0,
    TIP_ERR_INVALID_ARG = -1,
    TIP_ERR_UNSUPPORTED_CONFIG = -2,
    TIP_ERR_NOT_READY = -3,      /* tip_forward before tip_attach_packed */
    TIP_ERR_WORKSPACE = -4,      /* workspace too small / misaligned */
    TIP_ERR_HIP = -5,            /* a HIP runtime call failed; see tip_last_hip_error */
    TIP_ERR_NO_DEVICE = -6,
    TIP_ERR_ALLOC = -7,
    TIP_ERR_HANDOFF = -8         /* an inter-workgroup hand-off wait of an EARLIER launch gave up (see tip_check) */
} tip_status;

/* tip_forward flags */
#define TIP_FWD_LAST_ROW_ONLY 0x1 /* y is [B, size_s] = row T-1 of every window: what RTRunnerMin.step consumes
                                     (real_time_runner_minimal.py:150) */
#define TIP_FWD_KEEP_MASK     0x2 /* keep_mask [B,T,size_s] supplied: x_s * mask * keep_scale, replaces the Bernoulli
                                     draw of nn.Dropout(past_state_dropout) (simple_transformer_with_state.py:77) */

/* execution plans (tip_set_option(TIP_OPT_PLAN, ...)) */
#define TIP_PLAN_AUTO    0
#define TIP_PLAN_GENERAL 1 /* layer-by-layer MFMA GEMM kernels, any configuration */
#define TIP_PLAN_FUSED   2 /* one workgroup = one window through all encoder layers (paper configuration) */
#define TIP_PLAN_FUSED2  4 /* two windows per workgroup (80 rows = 5 MFMA row blocks, no padding); AUTO picks it for
                              B >= 2 x CUs; bit-identical results to TIP_PLAN_FUSED */
/* plan value 5 is RETIRED (rounds 1-5: TIP_PLAN_FUSED2S, a window pair on two co-resident workgroups; superseded by TIP_PLAN_FUSEDH
   and TIP_PLAN_FUSED1S, removed in round 6): tip_set_option answers TIP_ERR_UNSUPPORTED_CONFIG */
#define TIP_PLAN_FUSEDH  6 /* TIP_PLAN_FUSED with a hybrid row tiling: rows 0-31 on 16x16x4 MFMAs, rows 32-39 on 4x4x1 MFMAs fed by the
                              same weight fragments — no matrix-core work on the pad rows 40-47 outside the QKV projection.  No
                              inter-workgroup hand-off in the encoder.  Rows 0-31 bit-identical to TIP_PLAN_FUSED. */
/* plan values 7 and 8 are RETIRED (rounds 3-5: exploratory split-fp16 emulation of the fp32 GEMM operands — narrower arithmetic than the
   reference's fp32, never a default; removed in round 6): tip_set_option answers TIP_ERR_UNSUPPORTED_CONFIG */
#define TIP_PLAN_LATENCY 3 /* one window spread over up to 64 CUs per stage + GEMV-cluster RNN (paper config, B <= 64);
                              AUTO picks it for B <= 32 (up to 48 where four CUs per window are not to be had, and up to 64 where
                              TIP_PLAN_FUSED1S does not apply: T < 40) */

#define TIP_PLAN_FUSED1S 10 /* window-split: ONE window on TWO co-resident workgroups of one XCD — a column split with two partial-sum
                               hand-offs at 48 rows (3/5 of the matrix work per workgroup) — for batches that leave at least half of the
                               CUs idle: needs 2 B <= #CUs and B <= 128.  Its own summation order (K-halves of out-proj / linear2 summed
                               across the partners).  While 4 B <= #CUs (and B <= 64) the window is carried by FOUR workgroups instead (quads
                               of heads, quarters of the hidden units and of the RNN input projection; partial sums added p0 + p1 + p2 + p3
                               in every partner; TIP_OPT_F1S_PARTS pins the form) — the two forms differ in summation order, each is
                               deterministic.  AUTO picks it for 32 < B <= #CUs / 2 at T = 40 and for remainders of 33-128 windows behind
                               whole rounds. */
/* plan value 9 is RESERVED (rounds 4: TIP_PLAN_LATENCY1, the latency chain as one persistent kernel — bit-identical, measured slower
   than the launch chain at every batch size, removed in round 5): tip_set_option rejects it */

#define TIP_OPT_PLAN        1
#define TIP_OPT_PROFILE     2 /* 0 off; 1: bracket every stage with a HIP event pair; 2: only the dominant stage; 3: the dominant
                                 stage of every fourth forward (a pair costs ~7 us of queue barriers per step at B = 256: this is
                                 the form a timed region carries).  Setting it resets the accumulated times. */
#define TIP_OPT_RNN_CLUSTER 3 /* workgroups cooperating on one RNN window-tile (1,2,4,8,16); 0 = auto */
#define TIP_RNN_CLUSTER_ROWS4 0x44 /* TIP_OPT_RNN_CLUSTER value: 4-window tiles on 4-workgroup clusters (4x4x1 MFMAs; rnn_hidden 512 only) */
#define TIP_OPT_FAULT_INJECT 4 /* TESTS ONLY.  Bit 0: the window-split encoder drops one workgroup of its first window; bit 1: the clustered
                                 RNN drops member 1 of its first cluster; bit 2: the latency plan's GEMV RNN drops member 1 of
                                 stream 0.  The partners' waits then MUST give up (after a shortened spin): the error path of
                                 tip_check is exercised deterministically.  Bit 3: nobody is dropped, but every cooperating kernel
                                 treats its partners as sitting on DIFFERENT XCDs (agent-scope stores, L1-bypassing loads, paced
                                 polls) wherever they really are: the path a placement across XCDs takes, bit-identical results.
                                 Bit 4: in the one-launch few-stream form one producer of window 0 stamps its completion flag as a
                                 workgroup on ANOTHER XCD would (what kernels of a foreign stream running beside it can cause): its
                                 consumers give up at once and report TIP_OPT_HANDOFF_KIND = 2.  0 = off (default). */

#define TIP_OPT_FUSE_HEAD 5 /* RESERVED: accepted (0 / 1) and ignored.  Rounds 3-4: the output projection as the epilogue of the recurrence
                               kernel (measured neutral, removed in round 5); round 5: the projection inside the recurrence's hop wait
                               with an L2-ring hand-off (correct, measured slower than the separate kernels: CHANGELOG.md). */

/* option 6 is RETIRED (rounds 3-5: TIP_OPT_PACK_SPLIT16, the split-fp16 sections of the packed image): TIP_ERR_INVALID_ARG */
#define TIP_OPT_AUTO_DEMOTE 7 /* 1 (default): hosts may answer the first TIP_ERR_HANDOFF of this handle by demoting it (TIP_OPT_DEMOTED)
                                 and re-issuing the call; 0: they report the error.  A flag for the host layer (the library itself never
                                 re-issues a call); the Python host honours it. */
#define TIP_OPT_DEMOTED     8 /* 1: TIP_PLAN_AUTO and the automatic TIP_OPT_RNN_CLUSTER choose only kernels WITHOUT inter-workgroup hand-offs
                                 (hybrid one-window / two-window encoder or the general plan, single-workgroup recurrence tiles): no
                                 co-residency needed, a co-tenant costs throughput instead of frames.  Explicit plans / cluster sizes are
                                 still honoured.  The training step's two recurrences likewise run on single-workgroup tiles.  Default 0. */
#define TIP_OPT_NO_FLOW    10 /* 1: the few-stream plan never takes its ONE-launch form (B <= 24: stages, recurrence and projection as roles of one
                                 launch, every workgroup of a window on the window's XCD) but the launch chain for every batch it serves —
                                 what a host answers a placement loss with (TIP_OPT_HANDOFF_KIND = 2: kernels of another stream were
                                 dispatched beside the launch and broke the id mod 8 = XCD rule).  The chain needs co-residency only.
                                 Default 0. */
#define TIP_OPT_HANDOFF_KIND 11 /* READ ONLY (tip_get_option): what the sticky hand-off word of tip_check says.  0: nothing pending; 1: a
                                 cooperating kernel's wait gave up (co-residency lost: the answer is TIP_OPT_DEMOTED); 2: only the
                                 one-launch few-stream form found a producer on another XCD (the answer is TIP_OPT_NO_FLOW). */
#define TIP_OPT_F1S_PARTS   9 /* workgroups that share ONE window under TIP_PLAN_FUSED1S: 2, 4, or 0 (default) = four while 4 B <= #CUs and
                                 B <= 64, two otherwise.  An explicit 4 outside that range is TIP_ERR_UNSUPPORTED_CONFIG at the forward. */

/* ---- lifetime: replaces TF_RNN_Past_State.__init__ (simple_transformer_with_state.py:9-54) ---------------- */
TIP_API int tip_abi_version(void);
TIP_API int tip_create(const tip_config* cfg, tip_handle** out);
TIP_API void tip_destroy(tip_handle* h);
TIP_API const char* tip_strerror(int status);
TIP_API const char* tip_last_hip_error(const tip_handle* h);
TIP_API int tip_set_option(tip_handle* h, int option, int value);
TIP_API int tip_get_option(const tip_handle* h, int option, int* value);

/* ---- parameters: replaces state_dict()/load_state_dict() (train_model.py:109-111,220-225;
 *      offline_testing_simple.py:96).  Tensor i is the i-th entry of the reference's state_dict(), same shape. */
TIP_API int tip_num_tensors(const tip_handle* h);
TIP_API int tip_tensor_info(const tip_handle* h, int i, const char** name, int* rows, int* cols /* 0 for 1-D */);
/* size of the packed weight image (padded / permuted / folded copy the kernels read) */
TIP_API int tip_packed_bytes(const tip_handle* h, size_t* bytes);
/* host: build the packed image from the n state-dict tensors (host pointers, fp32, reference layout).
 * Folds applied here: channel shuffle :88-89 into in_linear rows; root-velocity zeroing :75 into in_linear
 * columns; 1/sqrt(d_head) into W_q/b_q when exact; b_ih + b_hh; MFMA-fragment ordering of W_hh. */
TIP_API int tip_pack_weights(const tip_handle* h, const float* const* host_tensors, int n, void* packed_host_out, size_t bytes);
/* device: point the handle at a packed image resident in HBM (caller-owned; e.g. the buffer every rank
 * receives from the one-time RCCL broadcast).  Must stay valid until the next attach / destroy. */
/* tip_pack_weights on the GPU: `tensors` are DEVICE pointers (the live parameters), `packed_dev` a device buffer of
 * tip_packed_bytes(); asynchronous on `stream`.  Bit-identical image; microseconds instead of a host pack + 27 MB upload. */
TIP_API int tip_pack_weights_device(const tip_handle* h, const float* const* tensors, int n, void* packed_dev, size_t bytes,
                            tip_stream_t stream);
TIP_API int tip_attach_packed(tip_handle* h, const void* packed_device, size_t bytes);

/* ---- forward: replaces TF_RNN_Past_State.forward(x_imu, x_s) (simple_transformer_with_state.py:60-102) ---- */
/* Scratch of one forward, 256-byte aligned, caller-owned; its contents need not survive between calls (the caller may reuse the
 * memory for anything).  Two forwards in flight must not share one (the library serialises forwards of different streams on the
 * device, but see the module's per-stream buffers).  For the paper configuration the first 1.06 MiB hold the completion flags and
 * launch counters of the one-launch few-stream plan (B <= 24): a flag counts as set only if it equals a 56-bit stamp (per-handle
 * nonce + the counter kept in the same area), so stale or foreign contents are harmless; the area sits at offset 0 for every
 * (B, T), so that a workspace used with changing batch sizes / window lengths keeps its counters. */
TIP_API int tip_workspace_bytes(const tip_handle* h, int B, int T, size_t* bytes);
/* largest batch one tip_forward (fp64 = 0) / tip_forward_f64 (fp64 = 1) call serves at window length T (32-bit buffer offsets
 * and grid limits); beyond it the calls return TIP_ERR_UNSUPPORTED_CONFIG.  Windows are independent (:60-102 has no op across
 * batch elements): a host runs a larger batch as chunks of at most this size. */
TIP_API int tip_max_batch(const tip_handle* h, int T, int fp64, int* max_batch);
/* x_imu [B,T,input_size_imu(+18)], x_s [B,T,size_s] (NaNs allowed, :65), y [B,T,size_s] (or [B,size_s] with
 * TIP_FWD_LAST_ROW_ONLY).  Inputs are not modified (:63-64).  keep_mask may be NULL. */
TIP_API int tip_forward(tip_handle* h, const float* x_imu, const float* x_s, float* y, int B, int T, int flags,
                const float* keep_mask, float keep_scale, void* workspace, size_t workspace_bytes,
                tip_stream_t stream);

/* The forward WITH the training step's encoder dropout and WITHOUT its activation stash: what a `.train()`-mode module computes when
 * nobody differentiates it — the unedited reference runner (offline_testing_simple.py:98 never calls .eval();
 * real_time_runner_minimal.py:149).  Few streams only: the configurations TIP_PLAN_LATENCY serves (paper configuration, B <= 64,
 * T <= 40), on that plan's kernels with the four dropout sites of every encoder layer live; TIP_ERR_UNSUPPORTED_CONFIG otherwise
 * (and on a demoted handle): the caller then takes tip_train_forward.  The keep decisions are tip_train_forward's for the same
 * (p_drop, seed) — same hash, same element indices — so a later tip_train_forward(same arguments) + tip_train_backward differentiates
 * exactly the function evaluated here (the Python module does that when .backward() is called after all).  Uses the ATTACHED packed
 * image like tip_forward; flags / workspace as tip_forward.  Past-state dropout (:77): an explicit keep_mask (TIP_FWD_KEEP_MASK), or
 * p_state > 0 with keep_mask NULL — the mask is then drawn inside the first kernel from (p_state, state_seed), the same decisions
 * tip_draw_keep_mask(p_state, state_seed, ...) writes out as a [B,T,size_s] tensor of 0 / 1 (for the tip_train_forward that a later
 * backward needs); kept values are scaled by keep_scale either way. */
TIP_API int tip_forward_dropout(tip_handle* h, const float* x_imu, const float* x_s, float* y, int B, int T, int flags,
                        const float* keep_mask, float keep_scale, float p_state, unsigned long long state_seed, float p_drop,
                        unsigned long long seed, void* workspace, size_t workspace_bytes, tip_stream_t stream);
/* mask[i] = 1 (keep) or 0 for i < n: the counter-based hash of the training step at site 0xFFFFFFF0, drop probability p_state in [0, 1) */
TIP_API int tip_draw_keep_mask(float p_state, unsigned long long state_seed, float* mask, size_t n, tip_stream_t stream);

/* ---- forward in fp64: the module built under `--double` (train_model.py:62-63,84-85: torch.set_default_dtype(float64), fp64
 *      windows :161-164).  Same function as tip_forward (simple_transformer_with_state.py:60-102) with every operation in IEEE
 *      double on the fp64 matrix cores.  `params` = host array of n_params DEVICE pointers to the fp64 state-dict tensors in
 *      tip_tensor_info() order, RAW (nothing is packed, no tip_attach_packed needed); x_imu / x_s / keep_mask / y as in tip_forward
 *      but double; flags: TIP_FWD_LAST_ROW_ONLY; workspace of tip_forward_f64_bytes(), 256-byte aligned.  Any configuration and
 *      any T >= 1.  A debugging / verification path: layer-by-layer kernels, not tuned. */
TIP_API int tip_forward_f64_bytes(const tip_handle* h, int B, int T, size_t* bytes);
TIP_API int tip_forward_f64(tip_handle* h, const double* const* params, int n_params, const double* x_imu, const double* x_s, double* y,
                    int B, int T, int flags, const double* keep_mask, double keep_scale, void* workspace, size_t workspace_bytes,
                    tip_stream_t stream);

/* ---- measurement ------------------------------------------------------------------------------------------ */
/* number of tip_forward calls that launched HIP kernels since tip_create (lets tests prove the HIP path ran) */
TIP_API int tip_forward_count(const tip_handle* h, uint64_t* n);
/* after the stream is synchronised: per-stage totals accumulated since TIP_OPT_PROFILE was last set.
 * ms[i] = summed duration of stage names[i], launches[i] = event pairs summed.  Arrays of length `cap`;
 * returns the number of stages (<= cap) or a negative status. */
TIP_API int tip_profile_read(tip_handle* h, const char** names, float* ms, int* launches, int cap);
/* (synchronises the device) number of inter-workgroup hand-off waits that gave up since the library was loaded.
 * The cooperating kernels (RNN clusters) never spin forever; if a peer does not arrive within ~1 s they continue
 * and bump this counter: non-zero => the outputs of that launch are invalid.  Must be 0 in a healthy process. */
TIP_API int tip_spin_timeouts(unsigned* count);
/* Hand-off failures are ERRORS, not numbers.  The cooperating plans (pair-split encoder, clustered / GEMV RNN) rely on
 * all their workgroups being resident at once: the launchers refuse a grid the runtime's occupancy query says cannot be
 * (hipErrorCooperativeLaunchTooLarge -> TIP_ERR_HIP), but another process or stream holding CUs can still starve a
 * partner.  A wait that gives up (bounded spin, ~1 s) (a) makes that workgroup produce NaN from there on, which the rest of
 * the forward propagates into the affected output rows — never finite-but-wrong values; (b) sets a sticky word in the
 * handle's pinned host block.  While it is set, tip_forward / tip_train_forward / tip_train_backward return
 * TIP_ERR_HANDOFF at entry, i.e. the call AFTER the failed launch completed reports it.  tip_check(h, 0) returns
 * TIP_ERR_HANDOFF or TIP_OK without synchronising (synchronise the stream first for a definitive answer about launches in
 * flight); tip_check(h, 1) also clears the word.  The GPU must be exclusively this process's for the cooperating plans to
 * run at speed; TIP_PLAN_FUSED / TIP_PLAN_GENERAL with TIP_OPT_RNN_CLUSTER = 1 need no co-residency at all.
 * Inside one process the library helps: tip_forward / tip_forward_f64 / tip_train_forward / tip_train_backward calls issued on
 * DIFFERENT streams of the handle's device (any handle) are serialised on the device (one event wait per stream switch; streams
 * are told apart by handle value, so a destroyed stream whose handle the runtime recycles counts as the same stream; calls on a
 * CAPTURING stream are exempt — see the HIP-graph note below), and a stream created with a CU mask
 * (hipExtStreamCreateWithCUMask) gets plan, grid and cluster sizes for the CUs its mask leaves.
 * After a lost hand-off a host can keep going without co-residency: tip_check(h, 1), tip_set_option(h, TIP_OPT_DEMOTED, 1). */
TIP_API int tip_check(tip_handle* h, int clear);

/* ---- streaming front/back-end: the model-facing half of RTRunnerMin.step
 *      (real_time_runner_minimal.py:59-85 record_raw_imu / record_state_aa_and_c, :87-112 smooth_and_split_s_c,
 *      :131-167 window build + pose assembly) for n_streams lock-stepped streams, entirely on the device.
 *      `state` is a caller-owned device buffer of tip_stream_state_bytes(); frame / call counters are the caller's.
 *      Per frame f = 0,1,2,...:  T = tip_stream_window_len(f);
 *          tip_stream_ingest(state, raw_imu[n,72], n, f, x_imu[n,T,90], x_s[n,T,131], stream);
 *          if (T > 0) { tip_forward(..., TIP_FWD_LAST_ROW_ONLY) -> y_last[n,131];
 *                       tip_stream_consume(state, y_last, n, f - 5, s_rest[n,111] (= s_t[3:114]), c_t[n,20], stream); }
 *      PyBullet FK and the SBP root-translation correction (:169-194) stay with the host.
 *      HIP graphs: kernel arguments are frozen at capture, so once the window is full (frame_idx >= 43 has been ingested, T = 40
 *      from then on) both calls accept TIP_STREAM_FRAME_AUTO for frame_idx / call_idx — "the frame after the last one ingested",
 *      read from a counter the ingest kernel keeps in `state`.  The triple ingest(AUTO) -> tip_forward -> consume(AUTO) can then be
 *      captured once (hipStreamBeginCapture) and replayed per frame: one graph launch instead of ~23 kernel launches
 *      (tip_amd.StreamingEngine(use_graph=True)).  Entry points called on a capturing stream skip the cross-stream serialisation
 *      and the CU-mask query (see tip_check above): replay such a graph when no other forward is in flight on the device.
 *      (tip_forward itself may be captured the same way for any batch; the clustered recurrence's per-launch XCC-exchange words are
 *      cleared by the kernels at their end for T >= 2, so a replay never reads a previous replay's words; capture T = 1 launches of
 *      more than 64 windows only if they are not replayed.) */
#define TIP_STREAM_FRAME_AUTO (-1)
TIP_API int tip_stream_state_bytes(int n_streams, size_t* bytes);
TIP_API int tip_stream_reset(void* state, const float* s_init /* [n,114] device */, int n_streams, tip_stream_t stream);
TIP_API int tip_stream_window_len(int frame_idx); /* 0 while the 11-tap smoother primes (frames 0..4), then 1..40 */
TIP_API int tip_stream_ingest(void* state, const float* raw_imu, int n_streams, int frame_idx, float* x_imu, float* x_s,
                      tip_stream_t stream);
TIP_API int tip_stream_consume(void* state, const float* y_last, int n_streams, int call_idx, float* s_rest, float* c_t,
                       tip_stream_t stream);
/* tip_stream_ingest for a frame whose forward is tip_forward_reuse on FULL windows: the state is advanced exactly as by
 * tip_stream_ingest, but only row T-1 of x_imu / x_s [n,T,*] is written (rows 0 .. T-2 keep whatever they held) — the reuse forward
 * takes every older row from its ring, and at >= 1024 streams gathering the 35-KB windows is most of the ingest's time. */
TIP_API int tip_stream_ingest_newest(void* state, const float* raw_imu, int n_streams, int frame_idx, float* x_imu, float* x_s,
                             tip_stream_t stream);

/* ---- exact streaming reuse: tip_forward for lock-stepped streams whose windows slide by one frame per call.
 *      In the runner a frame's model inputs never change once recorded (real_time_runner_minimal.py:74,85,137: raw_imu_buffer,
 *      s_and_c_in_buffer and imu_acc_sum_buffer are append-only), so with the stochastic parts off — past_state_dropout = 0, in_dropout = 0,
 *      module in .eval() — the frame's in_linear row (:79) and its layer-0 Q / K / V rows are the same numbers in each of the 40 windows
 *      it appears in (6.8 % of a window's FLOPs).  tip_forward_reuse keeps them per stream in `cache`, a caller-owned device ring of
 *      tip_reuse_cache_bytes() (256-byte aligned; 40 slots x 4 KiB per stream), and computes per call only the NEWEST row's:
 *        call k = 0, 1, 2, ... with frame_idx = f0 + k (consecutive; any f0 >= 0) and windows x_imu / x_s [B,T,*] whose rows 0 .. T-2
 *        are rows 1 .. T-1 of the previous call's windows (T growing 1 .. 40 while the history fills, then T = 40 and sliding);
 *        T < 40: the ring is written, the forward is tip_forward's; T = 40: layer 0 of the two-window encoder (TIP_PLAN_FUSED2, for any
 *        B) reads the ring instead of running prologue, in_linear and the QKV projection.  The results are BIT-IDENTICAL to
 *        tip_forward under TIP_PLAN_FUSED2 (AUTO's own choice for whole multiples of 2 x #CUs windows, e.g. 1024 streams) on the same
 *        windows: the ring's rows are produced by the same MFMA instruction over the same packed fragments in the same k order.
 *      frame_ctr (nullable): a DEVICE int holding the frame index, read by the kernels instead of frame_idx — what a captured HIP graph
 *        needs (arguments are frozen at capture); with the streaming front end: the counter tip_stream_ingest keeps at byte offset
 *        tip_stream_frame_counter_offset() of `state`.
 *      Every slot is tagged with the frame it was written for: a window whose 40 slots are not frames c-39 .. c (a skipped or repeated
 *      call, a ring that was reset or never primed) yields NaN rows, never another frame's numbers.  tip_reuse_reset clears the tags
 *      (required once before the first call, whenever the streams restart, and after tip_attach_packed of a new weight image: the
 *      ring's rows are functions of the weights — the Python host ties a ring to the image it was filled under and refuses a stale
 *      one).  flags: TIP_FWD_LAST_ROW_ONLY; a keep mask is refused
 *      (TIP_ERR_INVALID_ARG), other configurations than the paper's answer TIP_ERR_UNSUPPORTED_CONFIG.  Reference for what is reused:
 *      simple_transformer_with_state.py:63-79 (row-wise prologue + in_linear), torch functional.py:5785 (_in_projection_packed). */
TIP_API int tip_reuse_cache_bytes(const tip_handle* h, int n_streams, size_t* bytes);
TIP_API int tip_reuse_reset(void* cache, size_t cache_bytes, tip_stream_t stream);
TIP_API int tip_forward_reuse(tip_handle* h, const float* x_imu, const float* x_s, float* y, int B, int T, int flags, void* cache,
                      size_t cache_bytes, int frame_idx, const int* frame_ctr, void* workspace, size_t workspace_bytes,
                      tip_stream_t stream);
TIP_API int tip_stream_frame_counter_offset(size_t* bytes);

/* ---- training step: the model call of train_model.py:171-196 ---------------------
 * Replaces `y_pred = model(x_imu, x_s + noise)` in train mode (train_model.py:175) and the model part of
 * `loss.backward()` (train_model.py:192).  Loss, gradient clipping and the optimiser stay in PyTorch.
 *   params    host array of n_params device pointers: the state-dict tensors in tip_tensor_info() order, RAW (as the
 *             optimiser updates them; nothing is packed by the caller)
 *   keep_mask / keep_scale  as in tip_forward (past-state dropout :77); NULL = keep everything.  Input dropout :73 is
 *             applied by the caller to x_imu (no gradient flows to the inputs)
 *   p_drop, seed  dropout of the four nn.TransformerEncoderLayer sites (attention probabilities, after out_proj, after
 *             ReLU, after linear2; torch default p = 0.1, simple_transformer_with_state.py:26-29).  A mask is never
 *             stored: element idx of site s is kept iff  lowbias32((idx mod 2^32) * 0x9E3779B1 + key(seed, s)) >=
 *             floor(p * 2^32), where key = hi32(splitmix64_mix(seed + 0x9E3779B97F4A7C15 * (s + 1))) (once per site) and
 *             lowbias32(z): z ^= z>>16; z *= 0x7FEB352D; z ^= z>>15; z *= 0x846CA68B; z ^= z>>16 (all mod 2^32);
 *             site s = 4*layer + {0 attention P [B,H,T,T], 1 out_proj [M,D], 2 ffn hidden [M,F], 3 linear2 [M,D]},
 *             idx = row-major element index.
 *             Kept values are scaled by 1/(1-p).  p_drop = 0 switches it off.
 *   saved     activation stash written by the forward and read by the backward (tip_train_bytes: saved_bytes)
 *   scratch   backward workspace (scratch_bytes); grads = one flat buffer, tensors in tip_tensor_info() order.
 * Supported: with or without the RNN (:43-46), rnn_hid_size any multiple of 64 up to 512 (512: register-resident cluster kernels;
 * other widths: the streaming kernel in both directions), tf_in_dim 256/512/1024, head width 16/32/64, T <= 128 — else
 * TIP_ERR_UNSUPPORTED_CONFIG (the Python module then differentiates its torch-op composite instead).  A handle with
 * TIP_OPT_DEMOTED set runs both recurrences on single-workgroup tiles (no inter-workgroup hand-off). */
TIP_API int tip_train_bytes(const tip_handle* h, int B, int T, size_t* saved_bytes, size_t* scratch_bytes);
/* where one stashed activation of encoder layer `layer` lives inside `saved` (float offset, float count): */
#define TIP_SAVED_QKV  0 /* [M,3D] in-projection output (q | k | v)                                   */
#define TIP_SAVED_ATT  1 /* [M,D]  attention output before out_proj                                    */
#define TIP_SAVED_X1   2 /* [M,D]  LayerNorm1 output                                                    */
#define TIP_SAVED_HID  3 /* [M,F]  linear1 output after ReLU and dropout (> 0 exactly where the unit's gate is open) */
#define TIP_SAVED_XOUT 4 /* [M,D]  layer output (LayerNorm2)                                            */
#define TIP_SAVED_HALL 5 /* [M,R]  RNN states h_t (layer argument ignored but must be valid)           */
TIP_API int tip_train_saved_view(const tip_handle* h, int B, int T, int what, int layer, size_t* float_offset, size_t* floats);
TIP_API int tip_train_forward(tip_handle* h, const float* const* params, int n_params, const float* x_imu, const float* x_s,
                      const float* keep_mask, float keep_scale, float p_drop, unsigned long long seed, float* y, void* saved,
                      size_t saved_bytes, int B, int T, tip_stream_t stream);
TIP_API int tip_train_backward(tip_handle* h, const float* const* params, int n_params, const float* dy, const void* saved,
                       size_t saved_bytes, void* scratch, size_t scratch_bytes, float* grads, size_t grads_floats, float p_drop,
                       unsigned long long seed, int B, int T, tip_stream_t stream);

/* Gradients w.r.t. the step's INPUTS, to be called right after tip_train_backward on the same scratch (which still holds the gradient
 * w.r.t. in_linear's output): dx_imu [B,T,input_size_imu(+18)] and / or dx_s [B,T,size_s] (either may be NULL).  x_s, keep_mask and
 * keep_scale as given to tip_train_forward: d x_s carries the keep mask and is zero where x_s was NaN (:65) and in the root-velocity
 * columns (:75). */
TIP_API int tip_train_input_grads(tip_handle* h, const float* const* params, int n_params, const float* x_s, const float* keep_mask,
                          float keep_scale, void* scratch, size_t scratch_bytes, float* dx_imu, float* dx_s, int B, int T,
                          tip_stream_t stream);

/* ---- the same step for a module built under `--double` (train_model.py:84-85): fp64 parameters (raw, state-dict order), windows,
 *      keep mask, cotangent and gradients; same dropout decisions as the fp32 step for the same seed (kept values scaled by the fp32
 *      value of 1 / (1 - p), widened).  Any configuration tip_forward_f64 serves (with or without the RNN, any widths) with T <= 128 and
 *      (4 T d_head + 2 T^2) doubles of LDS <= 160 KB (attention backward: T <= 90 at head width 16) — else TIP_ERR_UNSUPPORTED_CONFIG.
 *      A debugging / verification path like tip_forward_f64: layer by layer, deterministic, not tuned.  saved / scratch: 256-byte
 *      aligned, tip_train_bytes_f64(); grads: one flat fp64 buffer, tensors in tip_tensor_info() order. */
TIP_API int tip_train_bytes_f64(const tip_handle* h, int B, int T, size_t* saved_bytes, size_t* scratch_bytes);
TIP_API int tip_train_forward_f64(tip_handle* h, const double* const* params, int n_params, const double* x_imu, const double* x_s,
                          const double* keep_mask, double keep_scale, float p_drop, unsigned long long seed, double* y, void* saved,
                          size_t saved_bytes, int B, int T, tip_stream_t stream);
TIP_API int tip_train_backward_f64(tip_handle* h, const double* const* params, int n_params, const double* dy, const void* saved,
                           size_t saved_bytes, void* scratch, size_t scratch_bytes, double* grads, size_t grads_doubles, float p_drop,
                           unsigned long long seed, int B, int T, tip_stream_t stream);

/* ---- train-set combiner and window gather -------------------------------------------------
 * tip_combine_sequence replaces the per-file body of store_imu_s_info (preprocess_and_combine_syn_amass.py:73-101):
 *   imu [L_imu,72], s = nimble_qdq [L_s,114], c = constrs [>=min(L),20]: fp64 DEVICE arrays as unpickled; bias [18] = the
 *   constant accelerometer bias the caller drew (:85); nan_root_vel = 1 for augmented-DIP files (:61-62).
 *   Writes float32 rows imu_out [n,72], sum_out [n,18], s_out [n,131] and returns n = min(L_imu, L_s) - 8, or 0 when the
 *   file is too short (min(L) <= 40, :68-70: nothing written), or a negative tip_status.
 * tip_gather_windows replaces TrainSubDataset's window slicing (training_data_loader.py:53-58,72-86): for each sampled
 *   end frame t_idx[i] (device int64, seq_length <= t < n_frames): x_imu[i] = [IMU[t-T:t] | SUM[t-T:t]] ([T,90]; [T,72]
 *   when sum_c is NULL), x_s[i] = S[t-T:t], y[i] = S[t-T+1:t+1]. */
TIP_API int tip_combine_frames(int L_imu, int L_s);
TIP_API int tip_combine_scratch_bytes(int L_imu, int L_s, size_t* bytes);
TIP_API int tip_combine_sequence(const double* imu, const double* s, const double* c, int L_imu, int L_s, const double* bias,
                         int nan_root_vel, float* imu_out, float* sum_out, float* s_out, void* scratch, size_t scratch_bytes,
                         tip_stream_t stream);
TIP_API int tip_gather_windows(const float* imu_c, const float* sum_c, const float* s_c, long long n_frames, const long long* t_idx,
                       int n, int T, float* x_imu, float* x_s, float* y, tip_stream_t stream);

/* ---- training losses and their gradient ---------------------------------------------------
 * Replaces learning_utils.py:13-78 as train_model.py:177-189 combines them: loss = loss_constr_multi + loss_q_only_2axis +
 * loss_jerk.  pred / gt: DEVICE fp32 rows [B*T, W], W = n_pose + n_vel + 4*n_sbp (reference: 108 + 3 + 20), row strides ld_*
 * in floats (so the reference's column slices can be passed as they are); n_vel is 3 or 0.  `terms` selects which of the
 * three functions are evaluated; a column group a term does not use may have width 0 (loss_jerk alone: n_vel = n_sbp = 0,
 * gt may be NULL).  Masking as the reference: rows whose GT root x/y are NaN leave the root-velocity terms, rows with any
 * NaN GT constraint leave the constraint terms; a mean over no rows is NaN.
 *   tip_loss_forward   stats[0..3] = total, loss_q, loss_c, loss_j (0 for a term not selected); the rest of stats
 *                      (TIP_LOSS_STATS floats, device) carries the normalisers the backward needs.
 *   tip_loss_backward  dpred[B*T, W] = gout * d total / d pred (gout: DEVICE scalar, NULL = 1); every column is written.
 * Deterministic: fixed-order fp64 reduction of per-workgroup partial sums.  W <= 256 (else TIP_ERR_UNSUPPORTED_CONFIG). */
#define TIP_LOSS_Q 1 /* loss_q_only_2axis (learning_utils.py:50-78) */
#define TIP_LOSS_C 2 /* loss_constr_multi (learning_utils.py:13-35) */
#define TIP_LOSS_J 4 /* loss_jerk         (learning_utils.py:38-47) */
#define TIP_LOSS_STATS 16
TIP_API int tip_loss_ws_bytes(int B, int T, size_t* bytes);
TIP_API int tip_loss_forward(const float* pred, long long ld_pred, const float* gt, long long ld_gt, int B, int T, int n_pose, int n_vel,
                     int n_sbp, int terms, float* stats, void* ws, size_t ws_bytes, tip_stream_t stream);
TIP_API int tip_loss_backward(const float* pred, long long ld_pred, const float* gt, long long ld_gt, int B, int T, int n_pose, int n_vel,
                      int n_sbp, int terms, const float* stats, const float* gout, float* dpred, long long ld_dpred,
                      tip_stream_t stream);

/* the same two passes for fp64 rows (train_model.py --double): pred / gt / stats / gout / dpred in double, everything else as above */
TIP_API int tip_loss_forward_f64(const double* pred, long long ld_pred, const double* gt, long long ld_gt, int B, int T, int n_pose, int n_vel,
                         int n_sbp, int terms, double* stats, void* ws, size_t ws_bytes, tip_stream_t stream);
TIP_API int tip_loss_backward_f64(const double* pred, long long ld_pred, const double* gt, long long ld_gt, int B, int T, int n_pose, int n_vel,
                          int n_sbp, int terms, const double* stats, const double* gout, double* dpred, long long ld_dpred,
                          tip_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* TIP_HIP_H */

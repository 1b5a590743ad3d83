/* tip_hip_debug.h — measurement hooks of libtip_hip.so.  NOT part of the drop-in boundary (include/tip_hip.h is): these read
 * in-kernel time stamps and launch counters that the profiling scripts under tools/ and two launch-path assertions in tests/
 * use.  They are exported so that those scripts can reach them through ctypes; nothing in the product path calls them.
 * Every reader copies `n` 64-bit words from a __device__ trace array (filled only when the matching TIP_*_TRACE environment
 * switch was set at launch) and returns 0, -1 for a bad `n`, -5 for a HIP error.
 *
 * Environment switches of the MEASUREMENT BUILD (csrc: `make measure` -> libtip_hip_measure.so, -DTIP_MEASURE; the default library
 * reads none of them — its kernel selection depends on tip_set_option only):
 *   traces (fill the arrays the readers below copy): TIP_FUSEDH_TRACE, TIP_FUSED2_TRACE, TIP_BWD_TRACE, TIP_RNN_TRACE, TIP_HEAD_TRACE,
 *     TIP_S16_TRACE
 *   ablations (wrong results, timing only): TIP_FUSED_ABLATE, TIP_RNN_ABLATE (bits: 1 polls never wait, 2 no MFMAs, 4 no row touches,
 *     16 A fragments of the first two batches only — a quarter of the LDS reads, 32 clamp instead of tanh; 128+ trace stamp selection)
 *   A/B selections: TIP_AUTO_SPLIT=0 (no rounds + remainder split), TIP_RNN_ROWS4=0 / TIP_RNN_W4=0|1 / TIP_RNN_C16=4 / TIP_RNN_HANDOFF=0 /
 *     TIP_RNN_PREPOLL=0 / TIP_RNN_ROTATE=0 (recurrence variants), TIP_HEAD=old (streaming projection kernel), TIP_GENERAL_PGEMM=0 /
 *     TIP_GENERAL_GEMM=32 / TIP_GENERAL_ATTN=v (general plan), TIP_TRAIN_FUSED=0 / TIP_TRAIN_FUSED_BWD=0 / TIP_TRAIN_FWD_PADDED /
 *     TIP_TRAIN_WIN_GEMM=0 / TIP_TRAIN_PGEMM=0 / TIP_TGEMM_TILE / TIP_TGEMM16_TILE / TIP_DW_KERNEL / TIP_DW_SPLITS / TIP_DW_SPLITDIV /
 *     TIP_DW_GROUP / TIP_DW_NB (training step)
 */
#ifndef TIP_HIP_DEBUG_H
#define TIP_HIP_DEBUG_H

#include "tip_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

TIP_API int tip_debug_read_fh_wg(unsigned long long* out, int n);      /* hybrid encoder: per-workgroup start / end stamps (tools/fh_trace.py) */
TIP_API int tip_debug_read_fh_trace(unsigned long long* out, int n);   /* hybrid encoder: phase stamps of workgroup 0 */
TIP_API int tip_debug_read_bwd_trace(unsigned long long* out, int n);  /* fused backward kernels: phase stamps (tools/bwd_trace.py) */
TIP_API int tip_debug_pgemm_launches(unsigned long long* out);         /* panel-GEMM launches since load (tests: the scaled widths take it) */
TIP_API int tip_debug_read_f2s_cross_xcd(unsigned* out);               /* pair-split plan: pairs whose halves sat on different XCDs */
TIP_API int tip_debug_read_f2_trace(unsigned long long* out, int n);   /* two-window encoder phase stamps (tools/f2_trace.py) */
TIP_API int tip_debug_read_flow_trace(unsigned long long* out, int n);  /* few-stream dataflow kernel: (entry, inputs ready, stored, published) stamps per stage (tools/flow_trace.py) */
TIP_API int tip_debug_read_f2s_trace(unsigned long long* out, int n);  /* window-split hand-off stamps */
TIP_API int tip_debug_read_rnn_trace(unsigned long long* out, int n);  /* clustered recurrence hand-off stamps (tools/rnn_trace.py) */
TIP_API int tip_debug_clock_probe(unsigned long long* dev_out, void* stream); /* s_memtime / s_memrealtime pair (bench.py: clock under load) */
TIP_API int tip_debug_read_head_wg(unsigned long long* out, int n);    /* output projection: per-workgroup lifetimes (tools/head_trace.py) */
TIP_API int tip_debug_read_head_trace(unsigned long long* out, int n); /* output projection: tile stamps */

#ifdef __cplusplus
}
#endif
#endif /* TIP_HIP_DEBUG_H */

/* tip_hip_debug.h — measurement hooks of libtip_hip.so.  NOT part of the drop-in boundary (include/tip_hip.h is): these read
 * in-kernel time stamps and launch counters that the profiling scripts under tools/ and two launch-path assertions in tests/
 * use.  They are exported so that those scripts can reach them through ctypes; nothing in the product path calls them.
 * Every reader copies `n` 64-bit words from a __device__ trace array (filled only when the matching TIP_*_TRACE environment
 * switch was set at launch) and returns 0, -1 for a bad `n`, -5 for a HIP error. */
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
TIP_API int tip_debug_read_f2s_trace(unsigned long long* out, int n);  /* pair-split hand-off stamps (tools/f2s_trace.py) */
TIP_API int tip_debug_read_rnn_trace(unsigned long long* out, int n);  /* clustered recurrence hand-off stamps (tools/rnn_trace.py) */
TIP_API int tip_debug_clock_probe(unsigned long long* dev_out, void* stream); /* s_memtime / s_memrealtime pair (bench.py: clock under load) */
TIP_API int tip_debug_read_head_wg(unsigned long long* out, int n);    /* output projection: per-workgroup lifetimes (tools/head_trace.py) */
TIP_API int tip_debug_read_head_trace(unsigned long long* out, int n); /* output projection: tile stamps */
TIP_API int tip_debug_read_s16_trace(unsigned long long* out, int n);  /* split-fp16 encoder phase stamps (tools/s16_trace.py) */

#ifdef __cplusplus
}
#endif
#endif /* TIP_HIP_DEBUG_H */

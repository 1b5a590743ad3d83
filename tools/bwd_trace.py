#!/usr/bin/env python3
"""Measurement only: phase timeline of the two fused backward kernels (workgroup 0, thread 0) from s_memtime stamps
(100 MHz ticks -> 10 ns each).  usage: TIP_BWD_TRACE=1 python tools/bwd_trace.py"""
import contextlib, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ.setdefault("TIP_LIB", "measure")   # the launchers' TIP_* switches exist in the measurement build only (csrc: make measure)
import tip_amd
from tip_amd import synth, lib as tlib

assert os.environ.get("TIP_BWD_TRACE"), "set TIP_BWD_TRACE=1"
cfg = synth.PAPER
with contextlib.redirect_stdout(sys.stderr):
    m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                                  dropout=0.0, in_dropout=0.0, past_state_dropout=0.0, with_acc_sum=True)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
m = m.cuda().train()
x_imu, x_s = synth.make_inputs(cfg, 64, 40)
xi = torch.tensor(np.tile(x_imu, (4, 1, 1))).cuda()
xs = torch.tensor(np.nan_to_num(np.tile(x_s, (4, 1, 1)))).cuda()
tgt = torch.randn(256, 40, 131, device="cuda")
for _ in range(4):
    for p in m.parameters():
        p.grad = None
    m(xi, xs).backward(tgt)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
assert tlib.load().tip_debug_read_bwd_trace(buf, 16) == 0
t = np.array(buf[:16], dtype=np.float64) * 0.01   # us
f, a = t[:8] - t[0], t[8:15] - t[8]
names_f = ["start", "LN2 bwd + partials", "chunk0: d(hidden) product", "chunk0: barrier", "chunk0: rows/bias out", "chunk0: dx1 product",
           "all 4 chunks", "dx1 rows out"]
names_a = ["start", "LN1 bwd + partials", "head 0 start", "head 1 start", "heads done", "barrier", "dx_in product + store"]
print("ffn_bwd (us since start, last layer processed):")
for n, v in zip(names_f, f):
    print(f"  {v:8.2f}  {n}")
print("attn_bwd:")
for n, v in zip(names_a, a):
    print(f"  {v:8.2f}  {n}")

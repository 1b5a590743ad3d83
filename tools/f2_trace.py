#!/usr/bin/env python3
"""Measurement only: phase timeline of the two-window encoder (workgroup 0, first pair, layer 1) from s_memtime stamps.
usage: TIP_FUSED2_TRACE=1 python tools/f2_trace.py"""
import contextlib, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ.setdefault("TIP_LIB", "measure")   # the launchers' TIP_* switches exist in the measurement build only (csrc: make measure)
import tip_amd
from tip_amd import synth, lib as tlib
cfg = synth.PAPER
with contextlib.redirect_stdout(sys.stderr):
    m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                                  dropout=0.0, in_dropout=0.0, past_state_dropout=0.0, with_acc_sum=True)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
m = m.cuda().eval()
m.set_plan("fused2")
x_imu, x_s = synth.make_inputs(cfg, 256, 40)
xi, xs = torch.tensor(np.tile(x_imu, (4, 1, 1))).cuda(), torch.tensor(np.tile(x_s, (4, 1, 1))).cuda()
with torch.no_grad():
    for _ in range(5):
        m(xi, xs)
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
assert tlib.load().tip_debug_read_f2_trace(buf, 64) == 0
t = np.array(buf[:], dtype=np.float64)
def row(name, a, b, mf=0):
    c = t[b] - t[a]
    print(f"  {name:46s} {c:8.0f} cyc = {c / 2400:6.2f} us" + (f"   MFMA issue (2 waves/SIMD) {64 * mf:6.0f} cyc = {6400 * mf / c:5.1f} %" if mf else ""))
for q in range(4):
    s0 = 0 if q == 0 else 5 * q
    row(f"quad {q}: Q|K + V projection + epilogues", s0, 1 + 5 * q, mf=5 * 16 * 4 + 2.5 * 16 * 4)
    row(f"quad {q}: barrier", 1 + 5 * q, 2 + 5 * q)
    row(f"quad {q}: attention (LDS planes)", 2 + 5 * q, 3 + 5 * q, mf=48)
    row(f"quad {q}: barrier", 3 + 5 * q, 4 + 5 * q)
    row(f"quad {q}: out-projection partial (K = 64) + barrier", 4 + 5 * q, 5 + 5 * q, mf=5 * 2 * 4 * 4)
row("residual epilogue + barrier", 20, 21)
row("LayerNorm1 + barrier", 21, 22)
for f in range(8):
    row(f"FFN chunk {f}: linear1 + ReLU epilogue + barrier", 22 if f == 0 else 24 + 2 * (f - 1), 23 + 2 * f, mf=5 * 16 * 4)
    row(f"FFN chunk {f}: linear2 partial + barrier", 23 + 2 * f, 24 + 2 * f, mf=5 * 2 * 8 * 4)
row("residual epilogue + barrier", 38, 40)
row("LayerNorm2 + barrier", 40, 41)
row("whole layer 1 (two windows)", 0, 41)

#!/usr/bin/env python3
"""Measurement only: phase timeline of the exploratory split-fp16 encoder (workgroup 0, layer 1) from s_memtime stamps.
usage: TIP_S16_TRACE=1 python tools/s16_trace.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
os.environ.setdefault("TIP_LIB", "measure")   # the launchers' TIP_* switches exist in the measurement build only (csrc: make measure)
from tip_amd import synth, lib as tlib
from sweep import model_for
m = model_for(synth.PAPER)
m.set_plan("fused16")
x_imu, x_s = synth.make_inputs(synth.PAPER, 64, 40)
xi = torch.tensor(np.tile(x_imu, (4, 1, 1))).cuda(); xs = torch.tensor(np.tile(x_s, (4, 1, 1))).cuda()
with torch.no_grad():
    for _ in range(30):
        m(xi, xs)
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
assert tlib.load().tip_debug_read_s16_trace(buf, 64) == 0
t = np.array(buf[:], dtype=np.float64)
def row(name, a, b, mf16=0, mf32=0):
    cyc = t[b] - t[a]
    ideal = 2 * (mf16 * 17 + mf32 * 32)     # two waves per SIMD
    print(f"  {name:46s} {cyc:8.0f} cyc" + (f"   MFMA issue {ideal:6.0f} cyc = {100 * ideal / cyc:5.1f} %" if ideal else ""))
row("prologue (input staging as split planes)", 0, 1)
row("in_linear + epilogue + split", 1, 2, mf16=7 * 18)
print("layer 1:")
row("head 0: Q|K|V projection (K = 256)", 8, 9, mf16=8 * 27)
row("head 0: attention (registers, fp32 MFMA)", 9, 10, mf32=48)
row("head 1: Q|K|V projection", 10, 11, mf16=8 * 27)
row("head 1: attention", 11, 12, mf32=48)
row("barrier", 12, 13)
row("out-projection + residual epilogue + barrier", 13, 14, mf16=8 * 18)
row("LayerNorm1 (+ split) + barrier", 14, 16)
for f in range(4):
    prev = 16 if f == 0 else 19 + 3 * (f - 1)
    row(f"FFN chunk {f}: linear1 + ReLU/split epilogue", prev, 17 + 3 * f, mf16=8 * 18)
    row(f"FFN chunk {f}: barrier", 17 + 3 * f, 18 + 3 * f)
    row(f"FFN chunk {f}: linear2 partial + barrier", 18 + 3 * f, 19 + 3 * f, mf16=8 * 18)
row("residual epilogue + barrier", 28, 30)
row("LayerNorm2 (+ split) + barrier", 30, 31)
row("whole layer 1", 8, 31)
row("RNN input projection + stores + sentinel", 40, 41, mf16=2 * 8 * 18)
row("whole window", 0, 41)

#!/usr/bin/env python3
"""Measurement only: tile timeline of the register-resident output projection (tip_head.hip) from s_memtime stamps.
usage: TIP_HEAD_TRACE=1 python tools/head_trace.py [B]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
os.environ.setdefault("TIP_LIB", "measure")   # the launchers' TIP_* switches exist in the measurement build only (csrc: make measure)
from tip_amd import synth, lib as tlib
from sweep import model_for
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
m = model_for(synth.PAPER)
x_imu, x_s = synth.make_inputs(synth.PAPER, min(B, 64), 40)
reps = (B + 63) // 64
xi = torch.tensor(np.tile(x_imu, (reps, 1, 1))[:B]).cuda(); xs = torch.tensor(np.tile(x_s, (reps, 1, 1))[:B]).cuda()
with torch.no_grad():
    for _ in range(5):
        m(xi, xs)
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 32)()
assert tlib.load().tip_debug_read_head_trace(buf, 32) == 0
t = np.array(buf[:], dtype=np.float64)
for wg, o in ((0, 0), (100, 16)):
    # s_memtime ticks at 100 MHz on gfx950 (REFCLK) — report both raw ticks and us
    tt = t[o:o + 16]
    print(f"workgroup {wg}: raw", [int(v - tt[0]) for v in tt])

wg = (ctypes.c_ulonglong * 512)()
assert tlib.load().tip_debug_read_head_wg(wg, 512) == 0
w = np.array(wg[:], dtype=np.float64).reshape(256, 2) / 100.0       # us
n = min(256, (B * 40 + 39) // 40)
w = w[:n]
t0 = w[:, 0].min()
dur = w[:, 1] - w[:, 0]
print(f"all {n} workgroups (s_memrealtime): entries spread {w[:, 0].max() - t0:.2f} us, first entry -> last exit {w[:, 1].max() - t0:.2f} us")
print(f"  per-workgroup lifetime: min {dur.min():.2f}  median {np.median(dur):.2f}  p90 {np.percentile(dur, 90):.2f}  max {dur.max():.2f} us")
order = np.argsort(-dur)[:8]
print("  slowest workgroups (id, xcd = id % 8, lifetime):", [(int(i), int(i) % 8, round(float(dur[i]), 2)) for i in order])
byx = [float(np.median(dur[np.arange(n) % 8 == x])) for x in range(8)]
print("  median lifetime per XCD:", [round(v, 2) for v in byx])

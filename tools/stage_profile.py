#!/usr/bin/env python3
"""Per-stage time of one forward from the library's own HIP-event stage timers (TIP_OPT_PROFILE=1).
usage: python tools/stage_profile.py [paper|scaled] [B] [T] [plan]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import sweep  # noqa: E402
from tip_amd import synth  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "scaled"
cfg = synth.SCALED if name == "scaled" else synth.PAPER
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
T = int(sys.argv[3]) if len(sys.argv) > 3 else 80
plan = sys.argv[4] if len(sys.argv) > 4 else "auto"
m = sweep.model_for(cfg)
r = sweep.measure(m, cfg, B, T, False, plan, 3)
print(f"{name} B={B} T={T} plan={plan}: {r['ms_per_forward']:.3f} ms, {r['tflops']:.1f} TFLOP/s ({100*r['frac_fp32_mfma_peak']:.1f} % of peak)")
m.set_plan(plan, profile=1)
x_imu, x_s = synth.make_inputs(cfg, 8, T, seed=5)
import numpy as np  # noqa: E402
xi = torch.tensor(np.tile(x_imu, ((B + 7) // 8, 1, 1))[:B]).cuda()
xs = torch.tensor(np.tile(x_s, ((B + 7) // 8, 1, 1))[:B]).cuda()
with torch.no_grad():
    for _ in range(3):
        m(xi, xs)
    torch.cuda.synchronize()
    m.profile_read()
    for _ in range(3):
        m(xi, xs)
    torch.cuda.synchronize()
tot = 0.0
for nm, ms, n in m.profile_read():
    print(f"  {nm:18s} {ms / 3:9.3f} ms per forward  ({n // 3} launches)")
    tot += ms / 3
print(f"  sum {tot:.3f} ms")

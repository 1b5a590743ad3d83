#!/usr/bin/env python3
"""Measurement only: hand-off timeline of the pair-split fused encoder (workgroup 0) from s_memtime stamps
(shader-clock ticks, ~2.2 GHz under load), and how many pairs ended up straddling XCDs.
usage: python tools/f2s_trace.py [--batch 256]"""
import argparse, contextlib, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ.setdefault("TIP_LIB", "measure")   # the launchers' TIP_* switches exist in the measurement build only (csrc: make measure)
import tip_amd
from tip_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
a = ap.parse_args()
cfg = synth.PAPER
with contextlib.redirect_stdout(sys.stderr):
    m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                                  dropout=0.0, in_dropout=0.0, past_state_dropout=0.0, with_acc_sum=True)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
m = m.cuda().eval()
m.set_plan("fused2s")
x_imu, x_s = synth.make_inputs(cfg, a.batch, 40)
xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
with torch.no_grad():
    for _ in range(5):
        m(xi, xs)
torch.cuda.synchronize()
lib = tip_amd.lib.load()
c = ctypes.c_uint()
assert lib.tip_debug_read_f2s_cross_xcd(ctypes.byref(c)) == 0
buf = (ctypes.c_ulonglong * 32)()
assert lib.tip_debug_read_f2s_trace(buf, 32) == 0
t = np.array(buf[:32], dtype=np.float64).reshape(8, 4)
print("workgroups whose partner sat on another XCD (all launches so far):", c.value)
print("hand-off  store+ack  counter+poll  load+add   compute until next   [ticks]")
for k in range(8):
    nxt = t[k + 1, 0] - t[k, 3] if k < 7 else 0.0
    print(f"   {k}     {t[k,1]-t[k,0]:8.0f}  {t[k,2]-t[k,1]:10.0f}  {t[k,3]-t[k,2]:9.0f}   {nxt:12.0f}")
print("first hand-off start -> last hand-off end:", t[7, 3] - t[0, 0], "ticks; spin time-outs:", tip_amd.lib.spin_timeouts())

#!/usr/bin/env python3
"""Measurement only: per-step timeline of the clustered RNN kernel (workgroup 0) from s_memtime stamps.
usage: TIP_RNN_TRACE=1 python tools/rnn_trace.py [--cluster C]"""
import contextlib, ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import tip_amd
from tip_amd import synth, lib as tlib
cfg = synth.PAPER
with contextlib.redirect_stdout(sys.stderr):
    m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                                  dropout=0.0, in_dropout=0.0, past_state_dropout=0.0, with_acc_sum=True)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
m = m.cuda().eval()
cl = int(sys.argv[sys.argv.index("--cluster") + 1]) if "--cluster" in sys.argv else 0
m.set_plan("fused", rnn_cluster=cl)
x_imu, x_s = synth.make_inputs(cfg, 256, 40)
xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
with torch.no_grad():
    for _ in range(5):
        m(xi, xs)
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 256)()
lib = tlib.load()
assert lib.tip_debug_read_rnn_trace(buf, 256) == 0
t = np.array(buf[:160], dtype=np.float64).reshape(40, 4)
ghz = 0.1  # s_memtime ticks at 100 MHz on this part if constant-rate; printed raw and as deltas
pull, mma, done = t[1:, 0], t[1:, 1], t[1:, 2]
print("ticks per step (median):", np.median(np.diff(t[1:, 0])), " by the stores-out stamps:", np.median(np.diff(t[1:, 2])))
print("pull-done -> mfma-done  :", np.median(mma - pull))
print("mfma-done -> stores-out :", np.median(done - mma))
print("stores-out -> next pull :", np.median(pull[1:] - done[:-1]))
probe = t[1:, 3]
if probe.max() > 0:
    print("stores-out -> arrival probe passes :", np.median(probe[1:] - done[:-1]))
    print("probe passes -> tile pulled + in LDS:", np.median(pull - probe))
x = np.array(buf[160:252], dtype=np.float64).reshape(23, 4)     # steps 1..23: first round done, loop exit, LDS written, rounds
if t[0, 3] > 0 and t[1, 3] > t[0, 3]:   # rows4 kernel: slots [0][3] = kernel entry, [2][3] = XCC exchange done, [1][3] = weights in registers
    print("entry -> XCC exchange done:", t[2, 3] - t[0, 3], " -> weights loaded:", t[1, 3] - t[2, 3], " -> step 0 stores out:", t[0, 2] - t[1, 3],
          " | entry -> last step done:", t[39, 2] - t[0, 3])
    probe = probe * 0
if x[:, 0].max() > 0 and probe.max() == 0:   # rows4 kernel: no arrival probe; times relative to the previous step's stores
    d = done[:23]
    print("stores-out -> first pull round checked   :", np.median(x[1:, 0] - d[:-1]))
    print("first round -> pull loop exit            :", np.median(x[:, 1] - x[:, 0]), " rounds (thread 0):", np.median(x[:, 3]), x[:, 3].max())
    print("loop exit -> own LDS writes done         :", np.median(x[:, 2] - x[:, 1]))
    print("LDS written -> barrier passed            :", np.median(t[1:24, 0] - x[:, 2]))
elif x[:, 0].max() > 0:
    pr = t[1:24, 3]
    print("probe passes -> first pull round checked:", np.median(x[:, 0] - pr))
    print("first round -> pull loop exit            :", np.median(x[:, 1] - x[:, 0]), " rounds (thread 0):", np.median(x[:, 3]), x[:, 3].max())
    print("loop exit -> own LDS writes done         :", np.median(x[:, 2] - x[:, 1]))
    print("LDS written -> barrier passed            :", np.median(t[1:24, 0] - x[:, 2]))
print("first rows:", t[:4])

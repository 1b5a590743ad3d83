#!/usr/bin/env python3
"""Measurement only: per-step timeline of the clustered RNN kernel (workgroup 0) from s_memtime stamps.
usage: TIP_RNN_TRACE=1 python tools/rnn_trace.py [--cluster C]     (C = 0 / 0x44: four-window tiles; 16: the 16-window kernel)
TIP_RNN_ABLATE=128+256*k with k = 1 (barrier passed) / 2 (MFMAs done) / 3 (poll loop left): two stamps per step only — the
stores-out stamp and point k — for phase lengths with little instrumentation in the way (TWO-STAMP line)."""
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
cl = int(sys.argv[sys.argv.index("--cluster") + 1], 0) if "--cluster" in sys.argv else 0
m.set_plan("fused", rnn_cluster=cl)
NB = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 256
x_imu, x_s = synth.make_inputs(cfg, NB, 40)
xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
with torch.no_grad():
    for _ in range(5):
        m(xi, xs)
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 256)()
lib = tlib.load()
assert lib.tip_debug_read_rnn_trace(buf, 256) == 0
t = np.array(buf[:160], dtype=np.float64).reshape(40, 4)
x = np.array(buf[160:252], dtype=np.float64).reshape(23, 4)     # steps 1..23: first round done, loop exit, LDS written, rounds
pull, mma, done = t[1:, 0], t[1:, 1], t[1:, 2]
rows4 = cl in (0, 0x44) and t[0, 3] > 0 and t[1, 3] > t[0, 3]   # rnn_rows4_kernel: slots [0][3] = kernel entry, [1][3] = weights in registers
print("kernel:", "rnn_rows4_kernel (4-window tiles, 4-workgroup clusters)" if rows4 else "rnn_resident_kernel (16-window tiles)")
print("ticks per step (median):", np.median(np.diff(done)))
print("TWO-STAMP: previous stores-out -> selected point:", float(np.median(mma[1:] - done[:-1])), "  selected point -> stores-out:", float(np.median(done - mma)))
print("pull-done (barrier passed) -> mfma-done  :", np.median(mma - pull))
print("mfma-done -> stores-out                  :", np.median(done - mma))
print("stores-out -> next pull-done             :", np.median(pull[1:] - done[:-1]))
if rows4:
    print("entry -> exchange done, weights loaded:", t[1, 3] - t[0, 3], " -> step 0 stores out:", t[0, 2] - t[1, 3],
          " | entry -> last step done:", t[39, 2] - t[0, 3])
    if x[:, 0].max() > 0:
        d = done[:23]
        print("stores-out -> first pull round checked   :", np.median(x[1:, 0] - d[:-1]))
        print("first round -> pull loop exit            :", np.median(x[:, 1] - x[:, 0]), " rounds (thread 0):", np.median(x[:, 3]), x[:, 3].max())
        print("loop exit -> own LDS writes done         :", np.median(x[:, 2] - x[:, 1]))
        print("LDS written -> barrier passed            :", np.median(t[1:24, 0] - x[:, 2]))
    print("(each stamp is an s_memtime + a global store by thread 0: the stamped workgroup runs ~8 % slower than the others;")
    print(" TIP_RNN_ABLATE=128 keeps only the stores-out stamp)")
else:
    probe = t[1:, 3]
    if probe.max() > 0:
        print("stores-out -> arrival probe passes :", np.median(probe[1:] - done[:-1]))
        print("probe passes -> tile pulled + in LDS:", np.median(pull - probe))
    if x[:, 0].max() > 0:
        pr = t[1:24, 3]
        print("probe passes -> first pull round checked:", np.median(x[:, 0] - pr))
        print("first round -> pull loop exit            :", np.median(x[:, 1] - x[:, 0]), " rounds (thread 0):", np.median(x[:, 3]), x[:, 3].max())
        print("loop exit -> own LDS writes done         :", np.median(x[:, 2] - x[:, 1]))
        print("LDS written -> barrier passed            :", np.median(t[1:24, 0] - x[:, 2]))

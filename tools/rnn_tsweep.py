#!/usr/bin/env python3
"""Measurement only: RNN stage time against the window length (start-up cost vs per-step cost of the clustered recurrence).
usage: python tools/rnn_tsweep.py [B] [--cluster C]"""
import contextlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ.setdefault("TIP_LIB", "measure")   # the launchers' TIP_* switches exist in the measurement build only (csrc: make measure)
import tip_amd
from tip_amd import synth
cfg = synth.PAPER
with contextlib.redirect_stdout(sys.stderr):
    m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                                  dropout=0.0, in_dropout=0.0, past_state_dropout=0.0, with_acc_sum=True)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
m = m.cuda().eval()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(args[0]) if args else 256
cl = int(sys.argv[sys.argv.index("--cluster") + 1], 0) if "--cluster" in sys.argv else 0
res = []
for T in (40, 16, 24, 32, 40):   # (the first pass warms up and is dropped)
    x_imu, x_s = synth.make_inputs(cfg, min(B, 256), T)
    xi = torch.tensor(np.tile(x_imu, ((B + 255) // 256, 1, 1))[:B]).cuda()
    xs = torch.tensor(np.tile(x_s, ((B + 255) // 256, 1, 1))[:B]).cuda()
    m.set_plan("fusedh", profile=1, rnn_cluster=cl)
    with torch.no_grad():
        for _ in range(40):
            m(xi, xs)
    torch.cuda.synchronize()
    st = {n: ms / k for n, ms, k in m.profile_read()}
    res.append((T, st["rnn_recurrence"] * 1e3))
    print(f"B={B} T={T:2d}: rnn {res[-1][1]:6.1f} us   " + " ".join(f"{k} {v*1e3:.1f}" for k, v in st.items()), flush=True)
ts, us = np.array([r[0] for r in res[1:]], float), np.array([r[1] for r in res[1:]])
p, s0 = np.polyfit(ts, us, 1)
print(f"fit: start-up {s0:.1f} us + {p:.3f} us per step")

#!/usr/bin/env python3
"""Race screen for the cooperating RNN kernels: many forwards per batch size, every result compared bit for bit with the first,
no hand-off time-out allowed.  usage: python tools/rnn_soak.py [iterations per batch size]"""
import contextlib, os, sys
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
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t0 = tlib.spin_timeouts()
for B, T in ((1, 40), (3, 40), (37, 40), (200, 40), (256, 40), (300, 40), (640, 33), (1024, 40), (1500, 40)):
    x_imu, x_s = synth.make_inputs(cfg, min(B, 256), T, seed=B)
    reps = (B + x_imu.shape[0] - 1) // x_imu.shape[0]
    xi = torch.tensor(np.tile(x_imu, (reps, 1, 1))[:B]).cuda()
    xs = torch.tensor(np.tile(x_s, (reps, 1, 1))[:B]).cuda()
    m.set_plan("fusedh" if B > 64 else "fused")
    with torch.no_grad():
        ref = m(xi, xs).clone()
        bad = 0
        for i in range(iters):
            y = m(xi, xs)
            if not torch.equal(y, ref):
                bad += 1
        torch.cuda.synchronize()
    m.check_handoffs()
    print(f"B={B:5d} T={T}: {iters} forwards, {bad} differing, finite {bool(torch.isfinite(ref).all())}", flush=True)
print("spin time-outs:", tlib.spin_timeouts() - t0)

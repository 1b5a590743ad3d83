#!/usr/bin/env python3
"""Endurance run of the headline step: `seconds` of back-to-back forwards (B = 256, T = 40, AUTO plan), the output of every 2000th
step compared bit for bit with the first one's, hand-off time-outs counted, per-chunk ms/step (drift).
usage: python tools/endurance.py [seconds = 600] [batch = 256]"""
import contextlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import tip_amd
from tip_amd import synth, lib as tlib
cfg = synth.PAPER
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 600.0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
with contextlib.redirect_stdout(sys.stderr):
    m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                                  dropout=0.0, in_dropout=0.0, past_state_dropout=0.0, with_acc_sum=True)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
m = m.cuda().eval()
x_imu, x_s = synth.make_inputs(cfg, B, 40, seed=3)
xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
t0s = tlib.spin_timeouts()
with torch.no_grad():
    ref = m(xi, xs).clone()
    torch.cuda.synchronize()
    chunk, steps, bad, ms = 2000, 0, 0, []
    t_end = time.time() + seconds
    while time.time() < t_end:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(chunk):
            y = m(xi, xs)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1) / chunk)
        steps += chunk
        bad += (not torch.equal(y, ref))
m.check_handoffs()
ms = np.array(ms)
print(f"B={B}: {steps} steps in {seconds:.0f} s, {bad} of {len(ms)} sampled outputs differing from the first; ms/step per 2000-step chunk: "
      f"min {ms.min():.4f} median {np.median(ms):.4f} max {ms.max():.4f} (first {ms[0]:.4f}, last {ms[-1]:.4f}); spin time-outs {tlib.spin_timeouts() - t0s}")

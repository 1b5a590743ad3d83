#!/usr/bin/env python3
"""Race screen for the latency plan (AUTO for <= 64 streams: GEMV recurrence on 4-workgroup clusters with granule hand-off): many
forwards per batch size, every result compared bit for bit with the first.  usage: python tools/latency_soak.py [iterations = 20000]"""
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
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
t0 = tlib.spin_timeouts()
m.set_plan("latency")
for B, last in ((64, True), (64, False), (33, True), (8, True), (1, True)):
    x_imu, x_s = synth.make_inputs(cfg, B, 40, seed=B)
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    f = m.forward_last if last else m
    with torch.no_grad():
        ref = f(xi, xs).clone()
        bad, first = 0, -1
        for i in range(iters):
            y = f(xi, xs)
            if not torch.equal(y, ref):
                bad += 1
                if first < 0:
                    first = i
                    d = (y - ref).abs()
                    rows = torch.nonzero(d.reshape(B, -1).amax(dim=1) > 0).flatten().tolist()
                    print(f"   first difference at forward {i}: streams {rows[:8]} max |diff| {d.max().item():.3e}", flush=True)
        torch.cuda.synchronize()
    m.check_handoffs()
    print(f"latency plan B={B:3d} last={last}: {iters} forwards, {bad} differing, finite {bool(torch.isfinite(ref).all())}", flush=True)
print("spin time-outs:", tlib.spin_timeouts() - t0)

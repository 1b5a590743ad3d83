"""Measurement only: recurrence stage time against the row stride of its state buffer (round 3: does the HALL layout alias in L2?)."""
import contextlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tip_amd
from tip_amd import synth
cfg = synth.PAPER
with contextlib.redirect_stdout(sys.stderr):
    m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                                  dropout=0.0, in_dropout=0.0, past_state_dropout=0.0, with_acc_sum=True)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
m = m.cuda().eval()
for T in (40, 39, 38, 37, 36, 33, 32):
    x_imu, x_s = synth.make_inputs(cfg, 256, T)
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    m.set_plan("auto", profile=1)
    with torch.no_grad():
        for _ in range(10): m(xi, xs)
        torch.cuda.synchronize(); m.profile_read(); m.set_plan("auto", profile=1)
        for _ in range(30): m(xi, xs)
        torch.cuda.synchronize()
    st = {n: ms / k for n, ms, k in m.profile_read()}
    print(f"T={T}: rnn {st['rnn_recurrence']*1e3:.1f} us = {st['rnn_recurrence']*1e3/T:.3f} us/step  (row stride {T*2048} B = {T*2048/4096:.1f} x 4 KiB)  head {st['out_linear']*1e3:.1f}", flush=True)

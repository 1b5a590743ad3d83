#!/usr/bin/env python3
"""Measurement only: RNN stage time of the 16-row clustered kernels (auto cluster) against the four-row-tile kernel, per batch.
usage: python tools/rnn_variants2.py [B ...]"""
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
for B in [int(a) for a in sys.argv[1:]] or [16, 64, 128, 200, 256, 300, 512, 768, 1024, 2048]:
    x_imu, x_s = synth.make_inputs(cfg, min(B, 256), 40)
    xi = torch.tensor(np.tile(x_imu, ((B + 255) // 256, 1, 1))[:B]).cuda()
    xs = torch.tensor(np.tile(x_s, ((B + 255) // 256, 1, 1))[:B]).cuda()
    row = []
    for cl in (4, 8, 16, 0x44):
        m.set_plan("fusedh", profile=1, rnn_cluster=cl)
        with torch.no_grad():
            for _ in range(30):
                m(xi, xs)
        torch.cuda.synchronize()
        st = {n: ms / k for n, ms, k in m.profile_read()}
        row.append(f"{'rows4' if cl == 0x44 else 'c' + str(cl)} {st['rnn_recurrence'] * 1e3:6.1f}")
    print(f"B={B:5d}: " + "  ".join(row), flush=True)

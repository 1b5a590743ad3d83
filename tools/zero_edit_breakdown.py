#!/usr/bin/env python3
"""Host-side breakdown of the unedited runner's model call (tools/zero_edit_loop.py): H2D of the two inputs, the module call (returns
with the kernels queued), the blocking .cpu().  usage: python tools/zero_edit_breakdown.py [frames = 300]"""
import contextlib, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import tip_amd
from tip_amd import synth
cfg = synth.PAPER
with contextlib.redirect_stdout(sys.stderr):
    m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4, dropout=0.0,
                                  in_dropout=0.0, past_state_dropout=0.8, with_acc_sum=True)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
m = m.cuda()
x_imu, x_s = synth.make_inputs(cfg, 1, 40, seed=1234)
h_i, h_s = torch.tensor(x_imu), torch.nan_to_num(torch.tensor(x_s))
warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rows = []
from tip_amd import lib as tlib
_orig = tlib.Handle.forward_dropout
_stamp = [0.0, 0.0]
def _fd(self, *a, **k):
    _stamp[0] = time.perf_counter()
    r = _orig(self, *a, **k)
    _stamp[1] = time.perf_counter()
    return r
tlib.Handle.forward_dropout = _fd
for i in range(n + 50):
    t0 = time.perf_counter()
    a = h_i.cuda(); b = h_s.cuda()
    t1 = time.perf_counter()
    y = m(a, b)
    t2 = time.perf_counter()
    y = y.cpu()
    t3 = time.perf_counter()
    row = y.squeeze(0)[-1, :].detach().numpy()
    t4 = time.perf_counter()
    if i >= 50:
        rows.append([(t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t4 - t0) * 1e3, (_stamp[0] - t1) * 1e3, (_stamp[1] - _stamp[0]) * 1e3, (t2 - _stamp[1]) * 1e3])
r = np.median(np.array(rows), axis=0)
print(f"p50 ms: two .cuda() {r[0]:.4f} | module call (async) {r[1]:.4f} | .cpu() {r[2]:.4f} | row to numpy {r[3]:.4f} | total {r[4]:.4f}")
print(f"  module call = {r[5]:.4f} before the library call + {r[6]:.4f} inside tip_forward_dropout (20 launches) + {r[7]:.4f} after")
if "--profile" in sys.argv:
    import cProfile, pstats
    a, b = h_i.cuda(), h_s.cuda()
    pr = cProfile.Profile(); pr.enable()
    for _ in range(300):
        y = m(a, b)
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(40)

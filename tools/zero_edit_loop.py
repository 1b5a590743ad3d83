#!/usr/bin/env python3
"""The unedited reference runner's model call in a loop (.train() mode, past_state_dropout 0.8, autograd recording, host tensors in,
row T-1 out): for kernel timelines under rocprofv3 (`rocprofv3 --kernel-trace ... -- python tools/zero_edit_loop.py`) and for p50s.
usage: python tools/zero_edit_loop.py [frames = 300]"""
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
m = m.cuda()                      # no .eval(): offline_testing_simple.py:98
x_imu, x_s = synth.make_inputs(cfg, 1, 40, seed=1234)
h_i, h_s = torch.tensor(x_imu), torch.nan_to_num(torch.tensor(x_s))
warnings.simplefilter("ignore")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
def pct(v):
    v = np.asarray(v)
    return "p50 %.4f  p90 %.4f  p95 %.4f  p99 %.4f  max %.4f ms" % (np.median(v), np.percentile(v, 90), np.percentile(v, 95), np.percentile(v, 99), v.max())
ts = []
for i in range(n + 50):
    t0 = time.perf_counter()
    y = m(h_i.cuda(), h_s.cuda()).cpu()          # real_time_runner_minimal.py:149
    row = y.squeeze(0)[-1, :].detach().numpy()   # :150
    if i >= 50:
        ts.append((time.perf_counter() - t0) * 1e3)
print(f"zero-edit runner call: p50 {np.median(ts):.4f} ms  p95 {np.percentile(ts, 95):.4f} ms  ({n} frames)")
print("  host call T=40    :", pct(ts))
# the first 40 frames of a run: T grows 1 -> 40 (3 passes; the first pass meets every window length for the first time)
grow = [[], [], []]
for rep in range(3):
    for t in range(1, 41):
        a, b = h_i[:, :t], h_s[:, :t]
        t0 = time.perf_counter()
        y = m(a.cuda(), b.cuda()).cpu()
        row = y.squeeze(0)[-1, :].detach().numpy()
        grow[rep].append((time.perf_counter() - t0) * 1e3)
for rep in range(3):
    print(f"  growing T pass {rep}  :", pct(grow[rep]), " argmax T =", int(np.argmax(grow[rep])) + 1)
# device time of the same call (events; inputs resident): what the kernels cost without the host protocol
d_i, d_s = h_i.cuda(), h_s.cuda()
dts = []
for i in range(n):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); y = m(d_i, d_s); e1.record(); e1.synchronize()
    dts.append(e0.elapsed_time(e1))
print("  device only T=40  :", pct(dts))
# the pieces of the host protocol on their own: the two .cuda() of pageable host tensors, the .cpu() of a resident output
c = []
for i in range(n):
    t0 = time.perf_counter(); a = h_i.cuda(); b = h_s.cuda(); c.append((time.perf_counter() - t0) * 1e3)
print("  two .cuda() alone :", pct(c))
yd = torch.zeros(1, 40, 131, device="cuda"); torch.cuda.synchronize(); c = []
for i in range(n):
    t0 = time.perf_counter(); yh = yd.cpu(); c.append((time.perf_counter() - t0) * 1e3)
print("  .cpu() alone      :", pct(c))

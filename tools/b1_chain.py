#!/usr/bin/env python3
"""B = 1 (few-stream) forward: p50 of the .eval() forward (full / last row), of the .train()-mode call with dropout live (what the unedited
runner pays on the device) and the back-to-back period of 300 queued forwards.  Environment switches of the measurement build apply
(TIP_LIB=measure is set here).  usage: [TIP_LAT_XCD=1 ...] python tools/b1_chain.py [B = 1] [T = 40]"""
import contextlib, os, sys, warnings
os.environ.setdefault("TIP_LIB", "measure")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import tip_amd
from tip_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = synth.PAPER
warnings.simplefilter("ignore")


def model(p_state):
    with contextlib.redirect_stdout(sys.stderr):
        m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4, dropout=0.0,
                                      in_dropout=0.0, past_state_dropout=p_state, with_acc_sum=True)
    m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
    return m.cuda()


def p50(fn, n=300):
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return np.median(ts), np.percentile(ts, 95)


x_imu, x_s = synth.make_inputs(cfg, B, T, seed=1234)
xi, xs = torch.tensor(x_imu).cuda(), torch.nan_to_num(torch.tensor(x_s)).cuda()
me = model(0.0).eval()
mt = model(0.8)
with torch.no_grad():
    for _ in range(50):
        me(xi, xs); me.forward_last(xi, xs)
    a = p50(lambda: me(xi, xs)); b = p50(lambda: me.forward_last(xi, xs))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        me.forward_last(xi, xs)
    e1.record(); e1.synchronize()
    period = e0.elapsed_time(e1) / 300
for _ in range(50):
    mt(xi, xs)
c = p50(lambda: mt(xi, xs))
me.check_handoffs(); mt.check_handoffs()
tags = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("TIP_") and k != "TIP_LIB")
print(f"B={B} T={T} [{tags}] eval full p50 {a[0]*1e3:.1f} p95 {a[1]*1e3:.1f} us | last row p50 {b[0]*1e3:.1f} p95 {b[1]*1e3:.1f} | back-to-back {period*1e3:.1f} | "
      f".train() dropout p50 {c[0]*1e3:.1f} p95 {c[1]*1e3:.1f}")

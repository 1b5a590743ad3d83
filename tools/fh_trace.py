#!/usr/bin/env python3
"""Measurement only: phase timeline of the hybrid one-window encoder (workgroup 0, layer 1) from s_memtime stamps.
usage: TIP_FUSEDH_TRACE=1 python tools/fh_trace.py"""
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
TRAIN = "--train" in sys.argv     # the training forward (stash + the four dropout sites, p = 0.1) instead of the inference kernel
m = m.cuda().train() if TRAIN else m.cuda().eval()
if TRAIN:
    m.ENCODER_DROPOUT = float(os.environ.get("FH_P", "0.1"))
else:
    m.set_plan("fusedh")
x_imu, x_s = synth.make_inputs(cfg, 256, 40)
xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(np.nan_to_num(x_s)).cuda()
if TRAIN:
    for _ in range(60):
        y = m(xi, xs)
        del y
    torch.cuda.synchronize()
else:
    with torch.no_grad():
        for _ in range(300):        # long enough for the clocks to settle
            m(xi, xs)
        torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
assert tlib.load().tip_debug_read_fh_trace(buf, 64) == 0
t = np.array(buf[:], dtype=np.float64)
GHZ = 2.4
def d(a, b): return (t[b] - t[a])
def row(name, cyc, mfma16=0, mfma4=0):
    ideal = mfma16 * 32 + mfma4 * 8          # issue cycles of this wave's MFMAs x 2 waves per SIMD
    print(f"  {name:44s} {cyc:8.0f} cyc = {cyc / GHZ / 1e3:6.2f} us" + (f"   MFMA issue (2 waves/SIMD) {2 * ideal:6.0f} cyc = {200 * ideal / cyc:5.1f} %" if ideal else ""))
print("window: prologue (input staging)", d(0, 1)); print("        in_linear + epilogue + barrier", d(1, 2))
print("layer 1:")
for c in range(2):
    row(f"head {c}: QKV projection (2 blocks + tail)", d(8 if c == 0 else 10, 9 + 4 * c), mfma16=2 * 3 * 16 * 4, mfma4=2 * 3 * 16 * 4)
    row(f"head {c}: attention (registers)", d(9 + 4 * c, 10 + 4 * c), mfma16=48)
row("barrier", d(14, 15))
row("out-projection (K = 256) + barrier", d(15, 16), mfma16=2 * 2 * 16 * 4, mfma4=2 * 2 * 16 * 4)
row("residual epilogue + barrier", d(16, 17))
row("LayerNorm1 + barrier", d(17, 18))
for f in range(4):
    prev = 18 if f == 0 else 21 + 3 * (f - 1)
    row(f"FFN chunk {f}: linear1 + ReLU epilogue", d(prev, 19 + 3 * f), mfma16=2 * 2 * 16 * 4, mfma4=2 * 2 * 16 * 4)
    row(f"FFN chunk {f}: barrier", d(19 + 3 * f, 20 + 3 * f))
    row(f"FFN chunk {f}: linear2 partial + barrier", d(20 + 3 * f, 21 + 3 * f), mfma16=2 * 2 * 16 * 4, mfma4=2 * 2 * 16 * 4)
row("residual epilogue + barrier", d(30, 31))
row("LayerNorm2 + barrier", d(31, 32))
row("whole layer 1", d(8, 32))
print("tail: RNN input projection + stores + sentinel", d(40, 41), "cyc =", d(40, 41) / GHZ / 1e3, "us")
print("whole window (prologue .. end):", d(0, 41), "cyc =", d(0, 41) / GHZ / 1e3, "us")

wg = (ctypes.c_ulonglong * 512)()
if tlib.load().tip_debug_read_fh_wg(wg, 512) == 0:
    w = np.array(wg[:], dtype=np.float64).reshape(256, 2) / 100.0       # us (s_memrealtime: 100 MHz, one counter for the device)
    t0 = w[:, 0].min()
    dur = w[:, 1] - w[:, 0]
    print(f"all 256 workgroups: entries spread {w[:, 0].max() - t0:.2f} us; first entry -> last exit {w[:, 1].max() - t0:.2f} us")
    print(f"  lifetime per workgroup: min {dur.min():.2f}  median {np.median(dur):.2f}  p90 {np.percentile(dur, 90):.2f}  max {dur.max():.2f} us")
    print("  median lifetime per XCD (workgroup id % 8):", [round(float(np.median(dur[np.arange(256) % 8 == x])), 2) for x in range(8)])
    order = np.argsort(-dur)[:6]
    print("  slowest:", [(int(i), round(float(dur[i]), 2)) for i in order])

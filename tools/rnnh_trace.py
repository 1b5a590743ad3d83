#!/usr/bin/env python3
"""Measurement only: per-step timeline of rnn_head_kernel (workgroup 0, thread 0) from s_memtime stamps.
usage: TIP_RNNH_TRACE=1 python tools/rnnh_trace.py [--batch B]"""
import contextlib, ctypes, os, sys
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
m.set_plan("fusedh")
NB = int(sys.argv[sys.argv.index("--batch") + 1]) if "--batch" in sys.argv else 256
x_imu, x_s = synth.make_inputs(cfg, min(NB, 256), 40)
xi = torch.tensor(np.tile(x_imu, ((NB + 255) // 256, 1, 1))[:NB]).cuda()
xs = torch.tensor(np.tile(x_s, ((NB + 255) // 256, 1, 1))[:NB]).cuda()
with torch.no_grad():
    for _ in range(5):
        m(xi, xs)
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (12 * 64))()
assert tlib.load().tip_debug_read_rnnh_trace(buf, 12 * 64) == 0
c = [int(buf[12 * 63 + i]) for i in range(3)]
print(f"all workgroups: {c[0]} wave-launches, in-stream poll missed in {c[1]} wave-steps ({c[1] / max(c[0], 1) / 39:.3f} of them), {c[2]} extra poll rounds "
      f"({c[2] / max(c[1], 1):.2f} per miss)")
if not (int(os.environ.get("TIP_RNNH_KNOB", "0")) & 256):
    sys.exit(0)
t = np.array(buf[:12 * 40], dtype=np.float64).reshape(40, 12)
if int(os.environ.get("TIP_RNNH_KNOB", "0")) & 512:
    st, nx = t[2:38], t[3:39]
    print("LIGHT: store -> store", np.median(nx[:, 2] - st[:, 2]), " barrier passed -> h stored", np.median(st[:, 2] - st[:, 0]),
          " h stored -> pull complete", np.median(st[:, 7] - st[:, 2]), " pull complete -> next barrier passed", np.median(nx[:, 0] - st[:, 7]))
    sys.exit(0)
S = lambda a: f"{np.median(a):7.0f}"
st = t[2:39]           # steady-state steps
prev = t[1:38]
print("ticks per step (store -> store)        :", S(st[:, 2] - prev[:, 2]))
print("barrier passed -> recurrence MFMAs done:", S(st[:, 1] - st[:, 0]))
print("MFMAs done -> h stored (tanh)          :", S(st[:, 2] - st[:, 1]))
print("h stored -> projection MFMAs done      :", S(st[:, 3] - st[:, 2]))
print("projection MFMAs done -> phase B end   :", S(st[:, 4] - st[:, 3]))
print("phase B end -> poll 1 checked          :", S(st[:, 5] - st[:, 4]), " pending:", np.mean(st[:, 9]))
p2 = st[:, 9] > 0
if p2.any():
    print("poll 1 -> poll 2 checked               :", S(st[p2, 6] - st[p2, 5]), " pending:", np.mean(st[p2, 10]))
print("h stored -> pull complete              :", S(st[:, 7] - st[:, 2]))
print("pull complete -> LDS written           :", S(st[:, 8] - st[:, 7]))
nx = t[3:40]
print("LDS written -> barrier passed          :", S(nx[:, 0] - st[:, 8]))

#!/usr/bin/env python3
"""Stage stamps of the few-stream dataflow kernel (lat_flow_kernel), window 0 / workgroup 0 of every stage: entry, inputs ready (wait
over), results stored, flag published — in shader cycles relative to stage 0's entry (all on XCD 0).  usage: python tools/flow_trace.py [B=1]"""
import contextlib, ctypes, os, sys, warnings
os.environ["TIP_LIB"] = "measure"; os.environ["TIP_FLOW_TRACE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import tip_amd
from tip_amd import synth, lib as tlib
TRAIN = "--train" in sys.argv      # the .train()-mode call (tip_forward_dropout: four dropout sites + the keep mask drawn in the first role)
argv = [a for a in sys.argv[1:] if not a.startswith("--")]
B = int(argv[0]) if argv else 1
cfg = synth.PAPER
warnings.simplefilter("ignore")
with contextlib.redirect_stdout(sys.stderr):
    m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4, dropout=0.0,
                                  in_dropout=0.0, past_state_dropout=0.8 if TRAIN else 0.0, with_acc_sum=True)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
m = m.cuda().train() if TRAIN else m.cuda().eval()
x_imu, x_s = synth.make_inputs(cfg, B, 40, seed=1234)
xi, xs = torch.tensor(x_imu).cuda(), torch.nan_to_num(torch.tensor(x_s)).cuda()
names = ["prologue", "in"] + [f"L{l}.{r}" for l in range(4) for r in ("qkv+attn", "out-proj", "ffn1", "ffn2")] + ["rnn-ih", "rnn", "head"]
rows = []
step = []
inst = []
with torch.no_grad():
    for it in range(30):
        (m(xi, xs) if TRAIN else m.forward_last(xi, xs)); torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * (len(names) * 4 + 12))()
        assert tlib.load().tip_debug_read_flow_trace(buf, len(names) * 4 + 12) == 0
        step.append(np.array(buf[len(names) * 4: len(names) * 4 + 6], dtype=np.float64))
        inst.append(np.array([buf[5]] + list(buf[len(names) * 4 + 8: len(names) * 4 + 11]) + [buf[6]], dtype=np.float64))
        a = np.array(buf[:len(names) * 4], dtype=np.float64).reshape(-1, 4)
        rows.append(a - a[0, 0])                      # relative to this forward's first stamp (every workgroup of window 0 sits on XCD 0)
r = np.median(np.stack(rows[10:]), axis=0)
t0 = r[0, 0]
print(f"B={B}: shader cycles (2.4 GHz: 2400 = 1 us), relative to stage 0's entry; 'hop' = inputs ready - producer's flag published")
prev_pub = None
for n, (e, w, st, pub) in zip(names, r):
    hop = (w - prev_pub) if prev_pub is not None else float('nan')
    print(f"{n:12s} entry {e-t0:9.0f}  ready {w-t0:9.0f}  stored {st-t0:9.0f}  published {pub-t0:9.0f} | body {st-w:7.0f}  publish {pub-st:6.0f}  hop {hop:7.0f}")
    prev_pub = pub
print(f"total (head stored) {r[-1,2]-t0:.0f} cycles = {(r[-1,2]-t0)/2400:.1f} us")
st = np.median(np.stack(step[10:]) - np.stack(step[10:])[:, :1], axis=0)
print("recurrence, member 0, step 20: poll starts 0 | own granules seen %.0f | all threads through the barrier %.0f | dot products + reduction %.0f | "
      "tanh + granule stored %.0f | step 21 stored %.0f (= one step)" % tuple(st[1:]))
ii = np.median(np.stack(inst[10:]) - np.stack(inst[10:])[:, :1], axis=0)
print("in_linear role, workgroup 0: body starts 0 | window staged in LDS %.0f | MFMAs %.0f | partials reduced %.0f | stored %.0f" % tuple(ii[1:]))

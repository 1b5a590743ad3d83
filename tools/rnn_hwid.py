#!/usr/bin/env python3
"""Measurement only: which workgroups of the four-window recurrence share a CU (HW_ID of every workgroup's first wave).
usage: TIP_RNN_TRACE=1 [TIP_RNN_W4=1] python tools/rnn_hwid.py [B]"""
import contextlib, ctypes, os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ.setdefault("TIP_LIB", "measure")   # the launchers' TIP_* switches exist in the measurement build only (csrc: make measure)
import tip_amd
from tip_amd import synth, lib as tlib
cfg = synth.PAPER
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
with contextlib.redirect_stdout(sys.stderr):
    m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                                  dropout=0.0, in_dropout=0.0, past_state_dropout=0.0, with_acc_sum=True)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
m = m.cuda().eval()
x_imu, x_s = synth.make_inputs(cfg, B, 40)
xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
with torch.no_grad():
    for _ in range(3):
        m(xi, xs)
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 2048)()
assert tlib.load().tip_debug_read_rnn_trace(buf, 2048) == 0
ids = np.array(buf[1024:1536], dtype=np.uint64)
n = int((ids != 0).sum())
print("workgroups stamped:", n)
where = collections.defaultdict(list)
for wg in range(512):
    v = int(ids[wg])
    if v == 0:
        continue
    hw, xcc = v & 0xffffffff, v >> 32
    cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
    where[(xcc, se, sh, cu)].append(wg)
print("distinct (xcc, se, sh, cu):", len(where))
for k in sorted(where)[:40]:
    print(k, where[k], " local index j = id >> 3:", [w >> 3 for w in where[k]])

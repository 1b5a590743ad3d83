#!/usr/bin/env python3
"""Training forward / forward+backward times of the library TIP_LIB selects (A/B runs of two builds in one gpurun call).
usage: TIP_LIB=[measure] python tools/train_ab.py [p_drop = 0.1]"""
import contextlib, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import tip_amd
from tip_amd import synth
warnings.simplefilter("ignore")
cfg = synth.PAPER
p = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
with contextlib.redirect_stdout(sys.stderr):
    m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4, dropout=0.0,
                                  in_dropout=0.0, past_state_dropout=0.8, with_acc_sum=True)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
m = m.cuda().train()
m.ENCODER_DROPOUT = p
x_imu, x_s = synth.make_inputs(cfg, 64, 40, seed=5)
xi = torch.tensor(np.tile(x_imu, (4, 1, 1))).cuda()
xs = torch.tensor(np.nan_to_num(np.tile(x_s, (4, 1, 1)))).cuda()
tgt = torch.randn(256, 40, 131, device="cuda")
def fwd():
    return m(xi, xs)
def fb():
    for q in m.parameters():
        q.grad = None
    fwd().backward(tgt)
def timed(fn, n=30):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
for _ in range(10):
    fb()
torch.cuda.synchronize()
f = min(timed(fwd) for _ in range(4))
b = min(timed(fb) for _ in range(4))
print(f"lib={os.environ.get('TIP_LIB') or 'default'} p={p}: forward {f:.4f} ms   forward+backward {b:.4f} ms")

#!/usr/bin/env python3
"""Training-step soak: two models stepped in lockstep on the same batches with the same dropout seeds must stay BIT-identical (the
kernels are deterministic — fixed summation orders, split-K partials summed in order — so a difference is a race), finite, with no
hand-off time-out.  usage: python tools/train_soak.py [steps = 200] [batch = 256]"""
import contextlib, copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import tip_amd
from tip_amd import synth, lib as tlib
cfg = synth.PAPER
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
T = 40
with contextlib.redirect_stdout(sys.stderr):
    ma = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                                   dropout=0.0, in_dropout=0.0, past_state_dropout=0.8, with_acc_sum=True)
ma.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
ma = ma.cuda().train()
ma.ENCODER_DROPOUT = 0.1
mb = copy.deepcopy(ma)
opts = [torch.optim.AdamW(m.parameters(), lr=1e-4) for m in (ma, mb)]
pool = []
for i in range(8):
    x_imu, x_s = synth.make_inputs(cfg, B, T, seed=300 + i)
    tgt = synth.normal(400 + i, "tgt", B * T * cfg["size_s"]).reshape(B, T, -1).astype(np.float32) * 0.3
    pool.append((torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda(), torch.tensor(tgt).cuda()))
t0 = tlib.spin_timeouts()
bad = nonfinite = 0
losses = []
for it in range(steps):
    xi, xs, tgt = pool[it % 8]
    ls = []
    for m, opt in zip((ma, mb), opts):
        torch.manual_seed(1000 + it)            # same past-state mask and the same encoder-dropout seed for both models
        opt.zero_grad()
        y = m(xi, xs)
        loss = ((y - tgt) ** 2).mean()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
        opt.step()
        ls.append(loss.detach())
    if it % 10 == 0 or it == steps - 1:
        same = all(torch.equal(pa, pb) for pa, pb in zip(ma.parameters(), mb.parameters())) and torch.equal(ls[0], ls[1])
        fin = all(bool(torch.isfinite(p).all()) for p in ma.parameters())
        bad += (not same); nonfinite += (not fin)
        losses.append(float(ls[0]))
torch.cuda.synchronize()
print(f"B={B} T={T}, {steps} training steps (dropout 0.1 / 0.8, AdamW, clip) on two models in lockstep: {bad} checks differing, {nonfinite} non-finite; "
      f"loss {losses[0]:.4f} -> {losses[-1]:.4f}; HIP training forwards {ma.hip_forward_count()}; spin time-outs {tlib.spin_timeouts() - t0}")

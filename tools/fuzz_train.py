#!/usr/bin/env python3
"""Randomised training-step fuzz: random (B, T <= 40), inputs and cotangents; the HIP training step (dropout off) against
  * B * T <= 800: the fp64 training oracle differentiating the SAME linear piece (ReLU gates taken from the run under test, as in
    tests/test_train_gpu.py): every parameter gradient to 1e-5 relative L2, outputs to 5e-5;
  * larger batches: the module's torch-op composite on the same GPU: outputs to 5e-5, gradients to 1e-2 relative L2 — loose on
    purpose: the composite's own fp32 forward puts a few ReLU gates on the other side of zero, and a flipped gate is a discrete change
    of the gradient (seen: 1.7e-3 on one FFN weight at B = 9, where the oracle with the run's own gates agrees to 1e-6).
usage: python tools/fuzz_train.py [seconds = 300] [seed = 0]"""
import contextlib, copy, os, sys, time, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import tip_amd
from tip_amd import synth, lib as tlib
from oracle import train_oracle
cfg = synth.PAPER
seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
with contextlib.redirect_stdout(sys.stderr):
    ma = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                                   dropout=0.0, in_dropout=0.0, past_state_dropout=0.0, with_acc_sum=True)
w = synth.make_weights(cfg, seed=2)
ma.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
ma = ma.cuda().train()
ma.ENCODER_DROPOUT = 0.0
ma.keep_train_stash = True
mb = copy.deepcopy(ma)
mb.use_hip_training = False
cases = 0
worst = (0.0, None)
worst2 = (0.0, None)
nsmall = 0
t_end = time.time() + seconds
t0 = tlib.spin_timeouts()
while time.time() < t_end:
    B = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 15, 16, 17, 31, 33, 63, 64, 65, 100, 128, 255, 256, 257, 300]))
    T = int(rng.choice([1, 2, 3, 4, 5, 8, 15, 16, 17, 23, 31, 32, 33, 35, 36, 37, 39, 40, 40, 40]))
    seed = int(rng.randint(1 << 30))
    x_imu, x_s = synth.make_inputs(cfg, B, T, seed=seed, nan_frac=float(rng.choice([0.0, 0.02])))
    cot = torch.tensor(synth.normal(seed & 0xffff, "cot", B * T * cfg["size_s"]).reshape(B, T, -1).astype(np.float32)).cuda()
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    small = B * T <= 800
    ma.zero_grad(set_to_none=True)
    n0 = ma.hip_forward_count()
    y = ma(xi, xs)
    assert ma.hip_forward_count() == n0 + 1, (B, T)
    (y * cot).sum().backward()
    torch.cuda.synchronize()
    assert torch.isfinite(y).all(), (B, T, seed)
    if small:
        gates = [(ma.train_activation(tlib.TIP_SAVED_HID, l) > 0).cpu().numpy().reshape(B, T, cfg["tf_hid_size"]) for l in range(cfg["tf_layers"])]
        yo, go = train_oracle.step(cfg, w, x_imu, x_s, cot.cpu().numpy(), relu_gates=gates)
        ey = float(np.abs(y.detach().cpu().numpy() - yo).max())
        assert ey < 5e-5, (B, T, seed, ey)
        for n, pa in ma.named_parameters():
            ref = go[n]
            err = float(np.linalg.norm(pa.grad.cpu().numpy().astype(np.float64) - ref) / (np.linalg.norm(ref) + 1e-30))
            assert np.isfinite(err) and err < 1e-5, (B, T, seed, n, err, "vs oracle")
            if err > worst[0]: worst = (err, (B, T, n, "oracle"))
    else:
        mb.zero_grad(set_to_none=True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            yb = mb(xi, xs)
        (yb * cot).sum().backward()
        ey = float((y.detach() - yb.detach()).abs().max())
        assert ey < 5e-5, (B, T, seed, ey)
        for (n, pa), (_, pb) in zip(ma.named_parameters(), mb.named_parameters()):
            ref = pb.grad.double()
            err = float((pa.grad.double() - ref).norm() / (ref.norm() + 1e-30))
            assert np.isfinite(err) and err < 1e-2, (B, T, seed, n, err, "vs composite")
            if err > worst2[0]: worst2 = (err, (B, T, n, "composite"))
    nsmall += small
    cases += 1
torch.cuda.synchronize()
print(f"{cases} random training steps in {seconds:.0f} s ({nsmall} against the fp64 oracle with the run's ReLU gates, gradients within 1e-5: worst {worst[0]:.2e} at {worst[1]}; "
      f"{cases - nsmall} against the torch-op composite, within 1e-2: worst {worst2[0]:.2e} at {worst2[1]}); outputs within 5e-5; spin time-outs {tlib.spin_timeouts() - t0}")

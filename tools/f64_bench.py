#!/usr/bin/env python3
"""--double (train_model.py:84-85) on the GPU: forward (eval) and forward + backward (train, dropout 0.1) of the paper model in fp64,
HIP path (tip_forward_f64 / tip_train_*_f64) against the torch-op composite (what the reference's module costs on stock PyTorch-ROCm:
rocBLAS dgemm, ATen).  One JSON line.  usage: python tools/f64_bench.py [--batch 256]"""
import argparse, contextlib, json, os, sys, warnings
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import tip_amd
from tip_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
cfg = synth.PAPER
torch.set_default_dtype(torch.float64)
with contextlib.redirect_stdout(sys.stderr):
    m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                                  dropout=0.1, in_dropout=0.0, past_state_dropout=0.8, with_acc_sum=True)
m.load_state_dict({k: torch.tensor(v, dtype=torch.float64) for k, v in synth.make_weights(cfg, seed=0).items()})
m = m.cuda()
x_imu, x_s = synth.make_inputs(cfg, a.batch, 40)
xi, xs = torch.tensor(x_imu, dtype=torch.float64).cuda(), torch.tensor(np.nan_to_num(x_s), dtype=torch.float64).cuda()
tgt = torch.randn(a.batch, 40, 131, device="cuda")

def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n

def fwd_eval():
    with torch.no_grad():
        return m(xi, xs)

def fwd_bwd():
    for p in m.parameters():
        p.grad = None
    m(xi, xs).backward(tgt)

res = {"config": f"paper config, fp64, B={a.batch} T=40"}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for name, hip in (("hip", True), ("torch_ops", False)):
        m.use_hip_training = hip
        m.eval()
        if hip:
            n0 = m.hip_forward_count()
            res["hip_ms_forward_eval"] = timed(fwd_eval, a.steps)
            assert m.hip_forward_count() > n0
        else:
            res["torch_ops_ms_forward_eval"] = timed(lambda: m._forward_torch_ops(xi, xs, keep_mask=None, apply_in_dropout=False), a.steps)
        m.train()
        res[f"{name}_ms_fwd_bwd_train"] = timed(fwd_bwd, a.steps)
flops = synth.flops_per_window(cfg, 40) * a.batch
res["hip_fwd_tflops"] = flops / res["hip_ms_forward_eval"] * 1e-9
res["hip_fwd_bwd_tflops"] = 3 * flops / res["hip_ms_fwd_bwd_train"] * 1e-9
res["torch_ops_fwd_bwd_tflops"] = 3 * flops / res["torch_ops_ms_fwd_bwd_train"] * 1e-9
print(json.dumps(res))

#!/usr/bin/env python3
"""Training-step measurement: the model call + backward of train_model.py:171-196 at the
reference's batch size (256 windows x 40 frames), HIP path vs the torch-op composite on the same GPU (what the
reference's own nn.Module costs on PyTorch-ROCm: rocBLAS / MIOpen / ATen kernels).  One JSON line.
usage: python tools/train_bench.py [--batch 256] [--steps 20] [--p-drop 0.1]"""
import argparse
import contextlib
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import tip_amd  # noqa: E402
from tip_amd import synth  # noqa: E402

PEAK = 157.3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--p-drop", type=float, default=0.1)
    ap.add_argument("--no-composite", action="store_true")
    ap.add_argument("--config", default="paper", choices=["paper", "scaled"], help="scaled = BASELINE.json configs[4] (T=80)")
    a = ap.parse_args()
    cfg = synth.PAPER if a.config == "paper" else synth.SCALED
    B, T = a.batch, (40 if a.config == "paper" else 80)
    with contextlib.redirect_stdout(sys.stderr):
        m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=cfg["rnn_hid_size"], tf_hid_size=cfg["tf_hid_size"],
                                      tf_in_dim=cfg["tf_in_dim"], n_heads=cfg["n_heads"], tf_layers=cfg["tf_layers"],
                                      dropout=0.0, in_dropout=0.0, past_state_dropout=0.8, with_acc_sum=True)
    w = synth.make_weights(cfg, seed=0)
    m.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    m = m.cuda().train()
    m.ENCODER_DROPOUT = a.p_drop
    opt = torch.optim.AdamW(m.parameters(), lr=1e-4)
    x_imu, x_s = synth.make_inputs(cfg, min(B, 64), T, seed=5)
    reps = (B + x_imu.shape[0] - 1) // x_imu.shape[0]
    xi = torch.tensor(np.tile(x_imu, (reps, 1, 1))[:B]).cuda()
    xs = torch.tensor(np.nan_to_num(np.tile(x_s, (reps, 1, 1))[:B])).cuda()
    tgt = torch.randn(B, T, cfg["size_s"], device="cuda")

    def run(hip, what):
        m.use_hip_training = hip

        def fwd():
            return m(xi, xs)

        def step():
            opt.zero_grad(set_to_none=True)
            y = fwd()
            loss = ((y - tgt) ** 2).mean()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
            opt.step()

        def fb():
            for p in m.parameters():
                p.grad = None
            y = fwd()
            y.backward(tgt)

        fn = {"forward": fwd, "fwd_bwd": fb, "step": step}[what]
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _ in range(a.warmup):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.steps):
                fn()
            e1.record()
            e1.synchronize()
        return e0.elapsed_time(e1) / a.steps

    fl = synth.flops_per_window(cfg, T)
    out = {"config": f"{a.config} config, train mode, B={B} T={T}, encoder dropout p={a.p_drop}, past-state dropout 0.8, AdamW",
           "flops_forward": B * fl, "flops_fwd_bwd": 3 * B * fl}
    for what in ("forward", "fwd_bwd", "step"):
        out["hip_ms_" + what] = run(True, what)
    out["hip_fwd_bwd_tflops"] = 3 * B * fl / out["hip_ms_fwd_bwd"] / 1e9
    out["hip_fwd_bwd_frac_fp32_mfma_peak"] = out["hip_fwd_bwd_tflops"] / PEAK
    out["hip_windows_per_s_step"] = B / out["hip_ms_step"] * 1e3
    if not a.no_composite:
        for what in ("forward", "fwd_bwd", "step"):
            out["torch_ops_ms_" + what] = run(False, what)
        out["speedup_step_vs_torch_ops"] = out["torch_ops_ms_step"] / out["hip_ms_step"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()

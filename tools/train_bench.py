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

os.environ.setdefault("TIP_LIB", "measure")   # the launchers' TIP_* switches exist in the measurement build only (csrc: make measure)
import tip_amd  # noqa: E402
from tip_amd import synth  # noqa: E402

PEAK = 157.3
N_SBPS = 5


def torch_losses(y_pred, y):
    """Baseline only: the loss part of the reference loop (train_model.py:177-189) as the torch ops it issues — third
    difference for the jerk term, boolean-mask row selection for the root-velocity and constraint terms, one BCE and one
    squared error per constraint (learning_utils.py:13-78)."""
    F = torch.nn.functional
    W, nc = y_pred.shape[-1], 4 * N_SBPS
    pose = y_pred[:, :, :W - nc - 3]
    jit = pose[:, 3:] - 3 * pose[:, 2:-1] + 3 * pose[:, 1:-2] - pose[:, :-3]
    l_j = (jit ** 2).mean() * 100.0
    p2, g2 = y_pred.reshape(-1, W), y.reshape(-1, W)
    pq, gq = p2[:, :W - nc].clone(), g2[:, :W - nc].clone()
    l_q = ((pq[:, :-3] - gq[:, :-3]) ** 2).mean() * 100.0
    keep = ~torch.any(gq[:, -3:-1].isnan(), dim=1)
    l_q = l_q + ((gq[:, -3:-1][keep] - pq[:, -3:-1][keep]) ** 2).mean() * 6.0 + ((gq[:, -1:][keep] - pq[:, -1:][keep]) ** 2).mean() * 12.0
    pc, gc = p2[:, W - nc:], g2[:, W - nc:]
    keep = ~torch.any(gc.isnan(), dim=1)
    pc, gc = pc[keep].clone(), gc[keep].clone()
    l_c = 0.0
    for i in range(N_SBPS):
        l_c = l_c + F.binary_cross_entropy(torch.sigmoid(pc[:, 4 * i:4 * i + 1]), gc[:, 4 * i:4 * i + 1]) \
            + ((pc[:, 4 * i + 1:4 * i + 4] - gc[:, 4 * i + 1:4 * i + 4] * 5.0) ** 2).mean() * 4.0
    return l_c / N_SBPS * 2.5 + l_q + l_j


def make_targets(B, T, W, seed=9):
    """GT rows with the training set's NaN pattern: root velocity NaN on ~30 % of the rows, constraints on ~20 %."""
    rng = np.random.RandomState(seed)
    gt = (rng.standard_normal((B, T, W)) * 0.5).astype(np.float32)
    c = gt[:, :, W - 4 * N_SBPS:].reshape(B, T, N_SBPS, 4)
    c[..., 0] = rng.rand(B, T, N_SBPS) < 0.4
    c[..., 1:] = rng.uniform(-0.25, 0.25, (B, T, N_SBPS, 3))
    gt[rng.rand(B, T) < 0.3, W - 4 * N_SBPS - 3:W - 4 * N_SBPS] = np.nan
    gt[rng.rand(B, T) < 0.2, W - 4 * N_SBPS:] = np.nan
    return gt


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
    gt = torch.tensor(make_targets(B, T, cfg["size_s"])).cuda()

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

        def ref_step():
            # the body of the reference's batch loop (train_model.py:171-198): noise, forward, three losses, backward, clip, AdamW
            noise = (torch.rand(xs.size(), device="cuda") - 0.5) * 0.06
            y_pred = m(xi, xs + noise)
            loss = tip_amd.learning_utils.train_loss(y_pred, gt, N_SBPS) if hip else torch_losses(y_pred, gt)
            opt.zero_grad()
            loss.backward()
            torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0)
            opt.step()

        def loss_only():
            yp = tgt.detach().requires_grad_(True)
            loss = tip_amd.learning_utils.train_loss(yp, gt, N_SBPS) if hip else torch_losses(yp, gt)
            loss.backward()

        def fb():
            for p in m.parameters():
                p.grad = None
            y = fwd()
            y.backward(tgt)

        fn = {"forward": fwd, "fwd_bwd": fb, "step": step, "ref_step": ref_step, "loss_fwd_bwd": loss_only}[what]
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
    whats = ("forward", "fwd_bwd", "step", "loss_fwd_bwd", "ref_step")
    for what in whats:
        out["hip_ms_" + what] = run(True, what)
    out["hip_fwd_bwd_tflops"] = 3 * B * fl / out["hip_ms_fwd_bwd"] / 1e9
    out["hip_fwd_bwd_frac_fp32_mfma_peak"] = out["hip_fwd_bwd_tflops"] / PEAK
    out["hip_windows_per_s_step"] = B / out["hip_ms_step"] * 1e3
    if not a.no_composite:
        for what in whats:
            out["torch_ops_ms_" + what] = run(False, what)
        out["speedup_step_vs_torch_ops"] = out["torch_ops_ms_step"] / out["hip_ms_step"]
        out["speedup_ref_step_vs_torch_ops"] = out["torch_ops_ms_ref_step"] / out["hip_ms_ref_step"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Measure every BASELINE.json configuration that fits one MI355X (absolute numbers + fraction of fp32-MFMA peak).
Writes one JSON object per line.  usage: python tools/sweep.py > profiles/rNN/sweep_n1.jsonl"""
import contextlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import tip_amd  # noqa: E402
from tip_amd import synth  # noqa: E402

PEAK = 157.3


def model_for(cfg):
    with contextlib.redirect_stdout(sys.stderr):
        m = tip_amd.TF_RNN_Past_State(
            cfg["input_size_imu"], cfg["size_s"], rnn_hid_size=cfg["rnn_hid_size"], tf_hid_size=cfg["tf_hid_size"],
            tf_in_dim=cfg["tf_in_dim"], n_heads=cfg["n_heads"], tf_layers=cfg["tf_layers"], dropout=0.0, in_dropout=0.0,
            past_state_dropout=0.0, with_rnn=True, with_acc_sum=True)
    w = synth.make_weights(cfg, seed=0)
    m.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    m = m.cuda().eval()
    m.refresh_packed()
    m.freeze_packed(True)
    return m


def measure(m, cfg, B, T, last, plan, iters):
    x_imu, x_s = synth.make_inputs(cfg, min(B, 64), T, seed=5)
    reps = (B + x_imu.shape[0] - 1) // x_imu.shape[0]
    xi = torch.tensor(np.tile(x_imu, (reps, 1, 1))[:B]).cuda()
    xs = torch.tensor(np.tile(x_s, (reps, 1, 1))[:B]).cuda()
    m.set_plan(plan)
    fn = m.forward_last if last else m
    with torch.no_grad():
        for _ in range(5):
            fn(xi, xs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn(xi, xs)
        e1.record()
        e1.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = synth.flops_per_window(cfg, T)
    return {"B": B, "T": T, "last_row_only": last, "plan": plan, "ms_per_forward": ms, "frames_per_s": B / ms * 1e3,
            "tflops": B * fl / ms / 1e9, "frac_fp32_mfma_peak": B * fl / ms / 1e9 / PEAK}


def main():
    out = []
    m = model_for(synth.PAPER)
    for (B, last, plan, it, name) in [
            (1, False, "auto", 300, "batch=1 single stream (60x real-time needs <= 0.278 ms)"),
            (1, True, "auto", 300, "batch=1 single stream, last row only (what RTRunnerMin consumes)"),
            (256, False, "auto", 50, "configs[1]: batch=256 seq_len=40 paper config"),
            (1024, True, "auto", 20, "configs[2]: 1024 streams, last row only (needs >= 60 forwards/s)"),
            (1024, False, "auto", 20, "configs[3] per-GPU share: 8192 streams / 8 GPUs, full output"),
            (8192, False, "auto", 5, "batch=8192 on ONE GPU"),
            (256, False, "general", 30, "batch=256, general (layer-by-layer) plan")]:
        r = measure(m, synth.PAPER, B, 40, last, plan, it)
        r["config"] = name
        out.append(r)
        print(json.dumps(r), flush=True)
    del m
    torch.cuda.empty_cache()
    ms = model_for(synth.SCALED)
    r = measure(ms, synth.SCALED, 512, 80, False, "auto", 3)
    r["config"] = "configs[4] per-GPU share: scaled model (12 layers, d=1024, ffn=4096, T=80), 4096/8 = 512 windows"
    print(json.dumps(r), flush=True)


if __name__ == "__main__":
    main()

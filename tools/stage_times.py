#!/usr/bin/env python3
"""Per-stage GPU time of the forward (in-library HIP events, TIP_OPT_PROFILE) over a batch sweep.
usage: python tools/stage_times.py [--B 256,1024] [--T 40] [--last] [--plan auto] [--iters 30]
Prints one line per batch: total ms and the average microseconds of every stage (fused_encoder, rnn_recurrence, out_linear ...)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tip_amd import synth  # noqa: E402
from sweep import model_for  # noqa: E402
import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", default="256")
    ap.add_argument("--T", type=int, default=40)
    ap.add_argument("--last", action="store_true")
    ap.add_argument("--plan", default="auto")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--config", default="paper", choices=["paper", "scaled"])
    a = ap.parse_args()
    cfg = synth.PAPER if a.config == "paper" else synth.SCALED
    m = model_for(cfg)
    for B in [int(b) for b in a.B.split(",")]:
        x_imu, x_s = synth.make_inputs(cfg, min(B, 64), a.T, seed=5)
        reps = (B + x_imu.shape[0] - 1) // x_imu.shape[0]
        xi = torch.tensor(np.tile(x_imu, (reps, 1, 1))[:B]).cuda()
        xs = torch.tensor(np.tile(x_s, (reps, 1, 1))[:B]).cuda()
        fn = m.forward_last if a.last else m
        with torch.no_grad():
            m.set_plan(a.plan)
            for _ in range(10):
                fn(xi, xs)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                fn(xi, xs)
            e1.record()
            e1.synchronize()
            total = e0.elapsed_time(e1) / a.iters
            m.set_plan(a.plan, profile=1)
            for _ in range(a.iters):
                fn(xi, xs)
            torch.cuda.synchronize()
            st = {k: round(ms_ / max(n_, 1) * 1e3, 2) for (k, ms_, n_) in m.profile_read()}
            m.set_plan(a.plan)
        print(json.dumps({"B": B, "T": a.T, "last": a.last, "plan": a.plan, "ms": round(total, 4), "stage_us": st}), flush=True)


if __name__ == "__main__":
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    main()

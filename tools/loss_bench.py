#!/usr/bin/env python3
"""GPU time of the fused training losses (tip_loss_forward / tip_loss_backward, csrc/tip_loss.hip) through the C-ABI, no
torch ops in the timed region.  One JSON line.  usage: python tools/loss_bench.py [--batch 256] [--iters 200]"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import tip_amd  # noqa: E402
from train_bench import make_targets, N_SBPS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--terms", type=int, default=7, help="bit mask: 1 loss_q_only_2axis, 2 loss_constr_multi, 4 loss_jerk")
    a = ap.parse_args()
    B, T, W = a.batch, 40, 131
    lib = tip_amd.lib.load()
    pred = torch.randn(B, T, W, device="cuda")
    gt = torch.tensor(make_targets(B, T, W)).cuda()
    dy = torch.empty_like(pred)
    stats = torch.zeros(16, device="cuda")
    nb = ctypes.c_size_t()
    lib.tip_loss_ws_bytes(B, T, ctypes.byref(nb))
    ws = torch.zeros(nb.value // 8, dtype=torch.float64, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def fwd():
        assert lib.tip_loss_forward(pred.data_ptr(), W, gt.data_ptr(), W, B, T, 108, 3, N_SBPS, a.terms, stats.data_ptr(), ws.data_ptr(),
                                    nb.value, st) == 0

    def bwd():
        assert lib.tip_loss_backward(pred.data_ptr(), W, gt.data_ptr(), W, B, T, 108, 3, N_SBPS, a.terms, stats.data_ptr(), None,
                                     dy.data_ptr(), W, st) == 0

    out = {"workload": f"B={B} T={T} W={W}, three losses fused", "bytes_rows": B * T * W * 4}
    for name, fn, passes in (("forward", fwd, 2), ("backward", bwd, 3)):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        out[name + "_us"] = us
        out[name + "_GBps"] = passes * B * T * W * 4 / us / 1e3      # forward: read pred + gt; backward: + write dpred
    out["loss"] = stats[:4].tolist()
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Row f-3 measurement: on-device train-set combiner and window gather vs the HBM roofline (8 TB/s), with the numpy
restatement of the reference's CPU code timed beside it on a bounded sample.  One JSON line.
usage: python tools/data_bench.py [--frames 2000000]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import tip_amd  # noqa: E402
from tip_amd import lib as tlib  # noqa: E402

HBM_PEAK = 8000.0   # GB/s, MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=2_000_000)
    ap.add_argument("--batch", type=int, default=256)
    a = ap.parse_args()
    L = a.frames
    dev = torch.device("cuda:0")
    lib = tlib.load()
    g = torch.Generator(device=dev).manual_seed(0)
    # random rotations by Rodrigues' formula (well-conditioned inputs for the root-frame inverse)
    v = torch.randn(L * 6, 3, device=dev, dtype=torch.float64, generator=g)
    th = v.norm(dim=1, keepdim=True)
    k = v / th
    K = torch.zeros(L * 6, 3, 3, device=dev, dtype=torch.float64)
    K[:, 0, 1], K[:, 0, 2], K[:, 1, 0], K[:, 1, 2], K[:, 2, 0], K[:, 2, 1] = -k[:, 2], k[:, 1], k[:, 2], -k[:, 0], -k[:, 1], k[:, 0]
    q = torch.eye(3, device=dev, dtype=torch.float64) + torch.sin(th)[:, :, None] * K + (1 - torch.cos(th))[:, :, None] * (K @ K)
    del K, v
    imu = torch.cat([q.reshape(L, 54), torch.randn(L, 18, device=dev, dtype=torch.float64, generator=g)], dim=1).contiguous()
    s = (torch.randn(L, 114, device=dev, dtype=torch.float64, generator=g) * 0.5).contiguous()
    c = torch.rand(L, 20, device=dev, dtype=torch.float64, generator=g).contiguous()
    bias = torch.zeros(18, device=dev, dtype=torch.float64)
    n = lib.tip_combine_frames(L, L)
    IMU = torch.empty(n, 72, device=dev)
    SUM = torch.empty(n, 18, device=dev)
    S = torch.empty(n, 131, device=dev)
    scratch = torch.empty(n * 18, dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def combine():
        rc = lib.tip_combine_sequence(imu.data_ptr(), s.data_ptr(), c.data_ptr(), L, L, bias.data_ptr(), 0, IMU.data_ptr(),
                                      SUM.data_ptr(), S.data_ptr(), scratch.data_ptr(), scratch.numel() * 8, st)
        assert rc == n, rc

    def timed(fn, iters):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / iters

    ms_c = timed(combine, 10)
    # algorithmic bytes per kept frame: fp64 inputs (72 + 114 + 20) read once, float32 outputs (72 + 18 + 131) written once
    bytes_c = n * ((72 + 114 + 20) * 8 + (72 + 18 + 131) * 4)
    T, B = 40, a.batch
    ends = torch.randint(T, n - 1, (B,), device=dev, dtype=torch.int64)
    xi = torch.empty(B, T, 90, device=dev)
    xs = torch.empty(B, T, 131, device=dev)
    y = torch.empty(B, T, 131, device=dev)

    def gather():
        rc = lib.tip_gather_windows(IMU.data_ptr(), SUM.data_ptr(), S.data_ptr(), n, ends.data_ptr(), B, T, xi.data_ptr(),
                                    xs.data_ptr(), y.data_ptr(), st)
        assert rc == 0

    ms_g = timed(gather, 200)
    bytes_g = B * ((T * 90 + (T + 1) * 131) * 4 + T * (90 + 131 + 131) * 4)   # rows read once + three outputs written
    # CPU: numpy restatement of the reference code on a bounded sample
    from oracle import data_oracle
    Ls = 50_000
    hi, hs, hc = imu[:Ls].cpu().numpy(), s[:Ls].cpu().numpy(), c[:Ls].cpu().numpy()
    t0 = time.time()
    a_, b_, c_ = data_oracle.combine_sequence(hi, hs, hc, np.zeros(18))
    cpu_c = time.time() - t0
    t0 = time.time()
    reps = 20
    for _ in range(reps):
        ws = [data_oracle.window(a_, b_, c_, int(t), T) for t in np.random.randint(T, len(a_) - 1, B)]
        _ = (np.array([w[0] for w in ws]), np.array([w[1] for w in ws]), np.array([w[2] for w in ws]))
    cpu_g = (time.time() - t0) / reps
    out = {"frames": n,
           "combine_ms": ms_c, "combine_frames_per_s": n / ms_c * 1e3, "combine_GBps": bytes_c / ms_c / 1e6,
           "combine_frac_hbm_peak": bytes_c / ms_c / 1e6 / HBM_PEAK,
           "gather_batch": B, "gather_us": ms_g * 1e3, "gather_GBps": bytes_g / ms_g / 1e6,
           "gather_frac_hbm_peak": bytes_g / ms_g / 1e6 / HBM_PEAK,
           "cpu_port": {"combine_frames_per_s": (Ls - 8) / cpu_c, "gather_batch_ms": cpu_g * 1e3, "cores": 1,
                        "sample": f"numpy restatement (oracle/data_oracle.py): {Ls} frames combined, {reps} batches of {B} windows gathered"}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

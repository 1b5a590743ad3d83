#!/usr/bin/env python3
"""Single-stream closed loop (the configuration a live demo runs): per-frame latency of StreamingEngine.step at n streams once
the 40-frame window is full — p50 / p95 over >= 300 frames, device time (events around the step), host time (wall time of the
call without waiting for the GPU) and synchronous wall time.  Two lines: launch by launch, then
StreamingEngine(use_graph=True) (one HIP-graph launch per frame).  usage: python tools/stream_latency.py [n=1] [frames=400]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import tip_amd  # noqa: E402
from tip_amd import synth  # noqa: E402
from sweep import model_for  # noqa: E402


def measure(m, n, frames=400, use_graph=False):
    from scipy.spatial.transform import Rotation
    rng = np.random.RandomState(n)
    base = Rotation.random(n * 6, random_state=n).as_matrix().reshape(n, 54).astype(np.float32)
    s_init = (rng.randn(n, 114) * 0.2).astype(np.float32)
    eng = tip_amd.streaming.StreamingEngine(m, s_init, use_graph=use_graph)
    dev_frames = [torch.tensor(np.concatenate([base, rng.randn(n, 18).astype(np.float32)], axis=1)).cuda() for _ in range(8)]
    for f in range(60 + 200):                    # prime the smoother, fill the window, and let the clocks settle
        eng.step(dev_frames[f % 8])
    torch.cuda.synchronize()
    wall, dev, host = [], [], []
    for f in range(frames):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        eng.step(dev_frames[f % 8])
        e1.record()
        t1 = time.perf_counter()
        e1.synchronize()
        t2 = time.perf_counter()
        wall.append((t2 - t0) * 1e3)
        host.append((t1 - t0) * 1e3)
        dev.append(e0.elapsed_time(e1))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for f in range(frames):                      # back to back, no per-frame synchronisation: throughput
        eng.step(dev_frames[f % 8])
    torch.cuda.synchronize()
    thr = (time.perf_counter() - t0) / frames * 1e3
    pct = lambda v, q: float(np.percentile(v, q))
    return {"streams": n, "frames": frames, "hip_graph": bool(use_graph),
            "sync_wall_ms_p50": pct(wall, 50), "sync_wall_ms_p95": pct(wall, 95),
            "device_ms_p50": pct(dev, 50), "device_ms_p95": pct(dev, 95),
            "host_call_ms_p50": pct(host, 50), "host_call_ms_p95": pct(host, 95),
            "back_to_back_ms_per_frame": thr, "realtime_factor_60fps_p50": (1e3 / 60.0) / pct(wall, 50)}


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    m = model_for(synth.PAPER)
    print(json.dumps(measure(m, n, frames)), flush=True)
    print(json.dumps(measure(m, n, frames, use_graph=True)), flush=True)

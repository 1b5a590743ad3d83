"""Exact streaming reuse (tip_forward_reuse) against the engine that recomputes every window: closed-loop
ms per frame for n lock-stepped streams, both engines in ONE run on one box (fresh boxes differ by up to 10 %).

    gpurun -- 'python tools/reuse_bench.py [n ...]'      (default: 512 1024 2048 8192)
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tip_amd  # noqa: E402
from tip_amd import synth  # noqa: E402


def timed(eng, frames, reps=3, n=40):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(n):
            eng.step(frames[f % 8])
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / n * 1e3)
    return best


def main():
    from scipy.spatial.transform import Rotation
    frames_only = "--frames-only" in sys.argv      # (tools/stream_kernels_by_n.sh: a short closed loop under rocprofv3, no timing)
    ns = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [512, 1024, 2048, 8192]
    cfg = synth.PAPER
    m = tip_amd.TF_RNN_Past_State(cfg["input_size_imu"], cfg["size_s"], rnn_hid_size=cfg["rnn_hid_size"], tf_hid_size=cfg["tf_hid_size"],
                                  tf_in_dim=cfg["tf_in_dim"], n_heads=cfg["n_heads"], tf_layers=cfg["tf_layers"], dropout=0.0,
                                  in_dropout=0.0, past_state_dropout=0.0, with_rnn=True, with_acc_sum=True)
    m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
    m = m.cuda().eval()
    m.freeze_packed(True)
    if frames_only:
        for n in ns:
            rng = np.random.RandomState(n)
            base = Rotation.random(n * 6, random_state=n).as_matrix().reshape(n, 54).astype(np.float32)
            fr = torch.tensor(np.concatenate([base, rng.randn(n, 18).astype(np.float32)], axis=1)).cuda()
            eng = tip_amd.streaming.StreamingEngine(m, (rng.randn(n, 114) * 0.2).astype(np.float32), reuse=n >= 512)
            for f in range(104):
                eng.step(fr)
            torch.cuda.synchronize()
        return
    for n in ns:
        rng = np.random.RandomState(n)
        base = Rotation.random(n * 6, random_state=n).as_matrix().reshape(n, 54).astype(np.float32)
        s_init = (rng.randn(n, 114) * 0.2).astype(np.float32)
        frames = [torch.tensor(np.concatenate([base, rng.randn(n, 18).astype(np.float32)], axis=1)).cuda() for _ in range(8)]
        res = {"streams": n}
        for tag, plan, kw in (("recompute_auto", "auto", {}), ("recompute_fused2", "fused2", {}), ("reuse", "auto", {"reuse": True})):
            eng = tip_amd.streaming.StreamingEngine(m, s_init, **kw)
            m.set_plan("auto")
            for f in range(44):                 # windows still growing: the two-window encoder serves full windows only
                eng.step(frames[f % 8])
            m.set_plan(plan)
            for f in range(44, 64):
                eng.step(frames[f % 8])
            res[f"{tag}_ms_per_frame"] = timed(eng, frames)
            del eng
        m.set_plan("auto")
        res["reuse_vs_auto"] = res["reuse_ms_per_frame"] / res["recompute_auto_ms_per_frame"]
        res["reuse_vs_fused2"] = res["reuse_ms_per_frame"] / res["recompute_fused2_ms_per_frame"]
        print(json.dumps(res), flush=True)
    # stage timers at 1024 streams: what the ring's writer and the ring-reading encoder cost
    n = 1024
    rng = np.random.RandomState(1)
    base = Rotation.random(n * 6, random_state=1).as_matrix().reshape(n, 54).astype(np.float32)
    s_init = (rng.randn(n, 114) * 0.2).astype(np.float32)
    fr = torch.tensor(np.concatenate([base, rng.randn(n, 18).astype(np.float32)], axis=1)).cuda()
    for kw in ({}, {"reuse": True}):
        eng = tip_amd.streaming.StreamingEngine(m, s_init, **kw)
        for f in range(50):
            eng.step(fr)
        m.set_plan("auto", profile=1)
        for f in range(20):
            eng.step(fr)
        torch.cuda.synchronize()
        print(json.dumps({"engine": "reuse" if kw else "recompute",
                          "stages_us_per_frame": {k: round(ms / max(1, c) * 1e3, 1) for k, ms, c in m.profile_read()}}), flush=True)
        m.set_plan("auto", profile=0)
        del eng


if __name__ == "__main__":
    main()

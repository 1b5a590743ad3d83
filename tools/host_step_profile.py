#!/usr/bin/env python3
"""Decision evidence for SURVEY.md section 8 row f-4: how long is the HOST half of the reference's per-frame step?

Runs only in the build container (imports /root/reference as it is).  RTRunnerMin.step and RTRunner.step are driven for a
few hundred frames with the model call replaced by a constant-time stand-in (returns a fixed output row), so what is timed
is everything AROUND `self.model(...)`: numpy window assembly, imu_rotate_to_local, the 6-tap filter, 6D -> axis-angle, FK
and the SBP / terrain root correction (real_time_runner.py:140-262, 264-277, 334-382, 451-496).

*** STUBBED FK ***  fairmotion -> scipy Rotation and pybullet / SimAgent -> a kinematic stand-in that returns fixed link
transforms (the stubs of tests/golden/make_runner_golden.py).  The PyBullet FK call itself therefore costs ~nothing here:
the numbers are a LOWER bound on the real host time per frame (real FK adds the C++ articulated-body update, typically
0.1-0.3 ms for a 19-link character).  Everything numpy/Python in the step is the reference's own code, unmodified.

usage: python tools/host_step_profile.py [--frames 400] > profiles/r02/host_step_profile.json
"""
import argparse
import cProfile
import io
import json
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
REF = "/root/reference"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=400)
    args = ap.parse_args()
    from make_runner_golden import FakeChar, install_stubs, smooth_imu_sequence
    install_stubs()
    sys.path.insert(0, REF)
    import amass_char_info
    from real_time_runner_minimal import RTRunnerMin
    from real_time_runner import RTRunner

    torch.set_num_threads(1)                       # the reference's evaluation setting (offline_testing_simple.py:34)
    torch.Tensor.cuda = lambda self, *a, **k: self  # harness only: no GPU here
    rng = np.random.RandomState(0)
    y_row = np.zeros(131, dtype=np.float32)
    y_row[:108] = np.tile(np.array([1, 0, 0, 1, 0, 0], dtype=np.float32), 18) + 0.01 * rng.randn(108).astype(np.float32)
    y_row[111::4] = rng.randn(5)                   # SBP logits of both signs: contacts switch on and off

    class ConstModel(torch.nn.Module):
        """constant-time stand-in for the model call: the forward is what the GPU kernels replace, not what is timed"""

        def __init__(self):
            super().__init__()
            self.t_in_model = 0.0
            self.calls = 0

        def forward(self, x_imu, x_s):
            t0 = time.perf_counter()
            B, T = x_imu.shape[0], x_imu.shape[1]
            y = torch.from_numpy(np.broadcast_to(y_row * (1.0 + 0.02 * np.sin(0.3 * self.calls)), (B, T, 131)).copy())
            self.calls += 1
            self.t_in_model += time.perf_counter() - t0
            return y

    s_init = np.zeros(114)
    s_init[3:57] = np.random.RandomState(5).randn(54) * 0.3
    s_init[2] = 0.95
    raw = smooth_imu_sequence(args.frames, 3)

    def fk_capable_char():
        ch = FakeChar(amass_char_info)
        return ch

    out = {"note": "STUBBED FK: fairmotion -> scipy, pybullet/SimAgent -> fixed-transform kinematic stand-in "
                   "(tests/golden/make_runner_golden.py); model call replaced by a constant-time stand-in; one CPU thread. "
                   "Host-side time per frame is a LOWER bound on the real runner's.",
           "frames": args.frames, "cpu": open("/proc/cpuinfo").read().split("model name")[1].split("\n")[0].strip(": \t")}

    def drive(name, make_runner, step):
        model = ConstModel()
        runner = make_runner(model)
        root = np.array([0.0, 0.0, 0.95])
        ts = []
        prof = cProfile.Profile()
        for t in range(args.frames):
            m0 = model.t_in_model
            t0 = time.perf_counter()
            if t >= args.frames // 2:
                prof.enable()
            res = step(runner, raw[t], root, t)
            prof.disable()
            dt = time.perf_counter() - t0 - (model.t_in_model - m0)
            root = res["qdq"][:3]
            ts.append(dt)
        steady = np.array(ts[60:])                 # the window is full (T = 40) from frame 45 on
        s = io.StringIO()
        pstats.Stats(prof, stream=s).sort_stats("cumulative").print_stats(14)
        top = [l.strip() for l in s.getvalue().splitlines() if "/root/reference" in l or "scipy" in l or "numpy" in l][:12]
        out[name] = {"host_ms_per_frame_p50": float(np.median(steady) * 1e3), "host_ms_per_frame_mean": float(steady.mean() * 1e3),
                     "host_ms_per_frame_p95": float(np.percentile(steady, 95) * 1e3),
                     "frames_per_s_one_stream_host_bound": float(1.0 / np.median(steady)),
                     "model_calls": model.calls, "cprofile_top_cumulative_second_half": top}

    drive("RTRunnerMin.step (real_time_runner_minimal.py:114-200)",
          lambda m: RTRunnerMin(fk_capable_char(), m, 40, s_init, with_acc_sum=True),
          lambda r, imu, root, t: r.step(imu, root))
    try:
        drive("RTRunner.step (real_time_runner.py:384-496, five SBPs, terrain + multi-SBP correction)",
              lambda m: RTRunner(fk_capable_char(), m, 40, s_init, map_bound=20.0, grid_size=0.1, five_sbp=True, with_acc_sum=True,
                                 multi_sbp_terrain_and_correction=True),
              lambda r, imu, root, t: r.step(imu, root, t))
    except Exception as e:   # the full runner touches more of SimAgent than the stand-in models
        out["RTRunner.step"] = {"error": f"{type(e).__name__}: {e}"}
    out["hip_forward_ms_b1_driver_r01"] = 0.216
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

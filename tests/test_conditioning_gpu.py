"""Parity OUTSIDE the random-init regime (round-2 review, item 1a).

Every other GPU test draws `synth.make_weights(gain=1)`: O(1/sqrt(fan_in)) weights, attention logits of O(0.1), a tanh
recurrence far from saturation — where the fast `__expf` softmax and the `__expf`/`__fdividef` tanh of the HIP path are
benign.  Trained checkpoints are not available (no network), so the regime is widened synthetically: the conditioning
sweep of tests/golden/make_golden.py --cond scales the same weights by gain 1.5 ... 4 and LayerNorm gamma by 3, where the
REFERENCE's own fp32-vs-fp64 difference (/root/reference/simple_transformer_with_state.py:60-102 run in both precisions)
grows from 6e-7 to 1e-2.  The bar: for every plan AUTO can pick,

    |hip - y64| <= max(2e-5, 3 x the reference's own fp32 noise)     and     <= 1e-4 wherever that noise is < 2.5e-5.

The measured ratios go to gpurun_out/cond_ratio.json (docs/DESIGN_NOTES_r01-r03.md section 3 quotes them)."""
import json
import os

import numpy as np
import pytest
import torch

from tip_amd import synth
from test_host_cpu import make_model
from conftest import cond_other_case

pytestmark = pytest.mark.gpu

from tip_amd import lib as _tlib
PLANS = ["latency", "fusedh", "fused", "fused2", "general"]      # AUTO's candidates for the paper configuration
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(w):
    m = make_model(synth.PAPER)
    m.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    return m.cuda().eval()


def test_conditioning_sweep_all_plans(cond_golden):
    assert torch.cuda.is_available()
    cfg = synth.PAPER
    xi, xs = torch.tensor(cond_golden["x_imu"]).cuda(), torch.tensor(cond_golden["x_s"]).cuda()
    report = {}
    worst = 0.0
    for tag, c in sorted(cond_golden["cases"].items()):
        g, lg = (float(v) for v in c["gain_ln"])
        m = _model(synth.make_weights(cfg, seed=0, gain=g, ln_gamma=lg))
        noise = c["noise"]
        row = {"gain": g, "ln_gamma": lg, "ref_fp32_noise": noise, "ymax": float(np.abs(c["y64"]).max())}
        for plan in PLANS:
            m.set_plan(plan)
            n0 = m.hip_forward_count()
            with torch.no_grad():
                y = m(xi, xs)
                yl = m.forward_last(xi, xs)
            torch.cuda.synchronize()
            assert m.hip_forward_count() == n0 + 2
            y = y.cpu().numpy()
            assert np.isfinite(y).all(), (tag, plan)
            err = float(np.abs(y - c["y64"]).max())
            row[plan] = err
            bound = max(2e-5, 3.0 * noise)
            assert err <= bound, (tag, plan, err, noise)
            if noise < 2.5e-5:
                assert err <= 1e-4, (tag, plan, err)
            assert np.abs(yl.cpu().numpy() - c["y64"][:, -1]).max() <= bound, (tag, plan, "last row")
            worst = max(worst, err / max(noise, 6e-7))
        report[tag] = row
    out = os.path.join(ROOT, "gpurun_out")
    try:
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "cond_ratio.json"), "w") as f:
            json.dump(report, f, indent=1, sort_keys=True)
    except OSError:
        pass
    print("worst HIP-error / reference-noise ratio:", worst)


def test_conditioning_other_widths_general_plan(cond_golden):
    """The same bar for the configurations only the general plan serves (scaled widths d = 1024 / d_head 64 at 2 layers with
    T = 80 — the panel GEMM and the matrix-core attention —, and the small configuration with d_head 32 / rnn 192 at T = 33)."""
    assert torch.cuda.is_available()
    for tag, c in sorted(cond_golden["other"].items()):
        cfg, w, x_imu, x_s = cond_other_case(tag, c)
        m = make_model(cfg)
        m.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
        m = m.cuda().eval()
        n0 = m.hip_forward_count()
        with torch.no_grad():
            y = m(torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()).cpu().numpy()
            yl = m.forward_last(torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()).cpu().numpy()
        assert m.hip_forward_count() == n0 + 2
        err, noise = float(np.abs(y - c["y64"]).max()), c["noise"]
        bound = max(2e-5, 3.0 * noise)
        assert np.isfinite(y).all() and err <= bound, (tag, err, noise)
        assert np.abs(yl - c["y64"][:, -1]).max() <= bound, (tag, "last row")
        print(f"{tag}: HIP error / reference fp32 noise = {err / noise:.2f}")

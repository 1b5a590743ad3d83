"""GPU: plan "fused1s" (round 4) — ONE 40-frame window carried by TWO or FOUR co-resident workgroups of one XCD (csrc/tip_fused2.hip,
fused_encoder2s_kernel<1, 2 | 4>: the pair-split kernel's column split and partial-sum hand-offs at 48 rows; four parts while
4 B <= #CUs, TIP_OPT_F1S_PARTS pins the form), AUTO's choice for batches that would leave at least half of the CUs idle
(32 < B <= #CUs / 2; reference function simple_transformer_with_state.py:60-102): against the fp64 oracle and the reference's
goldens, both output forms, explicit keep mask, batch independence, determinism, and a lost hand-off (a partner workgroup never
arrives)."""
import numpy as np
import pytest
import torch

from tip_amd import synth
from tip_amd import lib as tlib
from oracle import oracle
from conftest import seed_for_tag
from test_host_cpu import make_model, load_synth

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _model(seed=0):
    m = make_model(synth.PAPER)
    w = load_synth(m, synth.PAPER, seed)
    return m.cuda().eval(), w


def _fwd(m, xi, xs, last=False):
    n0 = m.hip_forward_count()
    with torch.no_grad():
        y = (m.forward_last if last else m)(xi, xs)
    torch.cuda.synchronize()
    assert m.hip_forward_count() == n0 + 1
    return y.cpu().numpy()


@pytest.mark.parametrize("B,parts", [(1, 2), (2, 2), (7, 2), (49, 2), (100, 2), (128, 2), (1, 4), (3, 4), (33, 4), (64, 4), (40, 0), (70, 0)])
def test_fused1s_vs_oracle_and_properties(B, parts):
    """parts: workgroups per window (TIP_OPT_F1S_PARTS; 0 = the library's choice: four while 4 B <= #CUs)."""
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    if max(parts, 2) * B > ncu:
        pytest.skip("needs two / four CUs per window")
    m, w = _model(1)
    plan = {0: "fused1s", 2: "fused1s2", 4: "fused1s4"}[parts]
    m.set_plan(plan)
    x_imu, x_s = synth.make_inputs(synth.PAPER, B, 40, seed=500 + B, nan_frac=0.02)
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    y = _fwd(m, xi, xs)
    sel = np.unique(np.array([0, B // 2, B - 1]))
    yo = oracle.forward(synth.PAPER, w, x_imu[sel], x_s[sel], dtype=np.float64)
    assert np.isfinite(y).all() and np.abs(y[sel] - yo).max() < TOL
    assert np.array_equal(_fwd(m, xi, xs, last=True), y[:, -1])
    for _ in range(3):
        assert np.array_equal(_fwd(m, xi, xs), y)                              # deterministic (fixed partner / summation order)
    if B > 2:                                                                   # a window's result does not depend on its batch neighbours
        n = 2 if parts or B <= 64 else 65                                       # (same form: "fused1s" takes four parts up to 64 windows)
        assert np.array_equal(_fwd(m, xi[1:1 + n].contiguous(), xs[1:1 + n].contiguous()), y[1:1 + n])
    xs2 = torch.nan_to_num(xs, nan=0.0)
    xs2[:, :, 108:111] = 3.25                                                   # :65 NaN scrub, :75 root-velocity columns ignored
    assert np.array_equal(_fwd(m, xi, xs2), y)
    m.set_plan("fusedh")
    assert np.abs(_fwd(m, xi, xs) - y).max() < 5e-6                             # the other plans: summation order only
    with pytest.raises(RuntimeError):                                           # T != 40 is outside the plan: an error, not a fallback
        m.set_plan(plan)
        _fwd(m, xi[:, :17].contiguous(), xs[:, :17].contiguous())
    m.check_handoffs()


@pytest.mark.parametrize("parts", [2, 4])
def test_fused1s_goldens_and_keep_mask(golden, parts):
    models = {}
    for tag, case in golden.items():
        if not tag.startswith("paper") or case["x_imu"].shape[1] != 40:
            continue
        key = tag.split("_B")[0]
        if key not in models:
            models[key] = _model(seed_for_tag(tag))[0]
        m = models[key]
        h = m._ensure_handle()
        m.refresh_packed(torch.device("cuda:0"))
        xi, xs = torch.tensor(case["x_imu"]).cuda(), torch.tensor(case["x_s"]).cuda()
        B = xi.shape[0]
        y = torch.zeros(B, 40, 131, device="cuda")
        ws = torch.empty(h.workspace_bytes(B, 40), dtype=torch.uint8, device="cuda")
        flags, mp, sc = 0, None, 1.0
        if "mask" in tag:
            mask = torch.tensor(case["mask"]).cuda()
            flags, mp, sc = tlib.TIP_FWD_KEEP_MASK, mask.data_ptr(), 1.0 / (1.0 - float(case["p"][0]))
        h.set_option(tlib.TIP_OPT_PLAN, tlib.TIP_PLAN_FUSED1S)
        h.set_option(tlib.TIP_OPT_F1S_PARTS, parts)
        assert h.get_option(tlib.TIP_OPT_F1S_PARTS) == parts
        h.forward(xi.data_ptr(), xs.data_ptr(), y.data_ptr(), B, 40, flags, mp, sc, ws.data_ptr(), ws.numel(),
                  torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.abs(y.cpu().numpy() - case["y64"]).max() < TOL, tag


def test_fused1s_limits():
    m, _ = _model()
    m.set_plan("fused1s")
    x_imu, x_s = synth.make_inputs(synth.PAPER, 129, 40, seed=1)
    with pytest.raises(RuntimeError):                                           # more than #CUs / 2 windows: not co-resident
        _fwd(m, torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda())
    m.set_plan("fused1s4")
    with pytest.raises(RuntimeError):                                           # four workgroups per window asked for, 65 windows
        _fwd(m, torch.tensor(x_imu[:65]).cuda(), torch.tensor(x_s[:65]).cuda())
    with pytest.raises(Exception):
        m._ensure_handle().set_option(tlib.TIP_OPT_F1S_PARTS, 3)


@pytest.mark.handoff_fault
@pytest.mark.parametrize("plan", ["fused1s2", "fused1s4"])
def test_fused1s_lost_handoff_poisons_and_raises(plan):
    m, _ = _model()
    h = m._ensure_handle()
    h.set_option(tlib.TIP_OPT_AUTO_DEMOTE, 0)
    m.set_plan(plan)
    x_imu, x_s = synth.make_inputs(synth.PAPER, 20, 40, seed=6)
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    ref = _fwd(m, xi, xs)
    t0 = tlib.spin_timeouts()
    h.set_option(tlib.TIP_OPT_FAULT_INJECT, 1)                                  # workgroup (window 0, half 1) never arrives
    y = _fwd(m, xi, xs)
    h.set_option(tlib.TIP_OPT_FAULT_INJECT, 0)
    assert tlib.spin_timeouts() > t0
    bad = np.isnan(y)
    assert bad[0].all() and not bad[1:].any() and np.array_equal(y[1:], ref[1:])   # window 0 only; never finite-but-wrong
    with pytest.raises(tlib.TipHandoffError):
        m(xi, xs)
    h.check_clear()
    assert np.array_equal(_fwd(m, xi, xs), ref)
    m.check_handoffs()

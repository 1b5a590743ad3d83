"""GPU: plan "latency1" — the few-stream latency chain as ONE persistent kernel (csrc/tip_latency.hip, lat1_kernel: grid barriers
instead of 20 kernel boundaries, recurrence and output projection as its tail) — against the launch chain it replaces
("latency": bit-identical, same K splits and reduction orders), against the fp64 oracle, across window lengths, for both output
forms, with a keep mask, under HIP-graph replay, and with a lost hand-off (reference: real_time_runner_minimal.py:146-150 calls
the model with B = 1 and T growing 1 -> 40)."""
import numpy as np
import pytest
import torch

from tip_amd import synth
from tip_amd import lib as tlib
from oracle import oracle
from test_host_cpu import make_model, load_synth

pytestmark = pytest.mark.gpu
TOL = 2e-5


def _model(seed=0, **kw):
    m = make_model(synth.PAPER, **kw)
    w = load_synth(m, synth.PAPER, seed)
    return m.cuda().eval(), w


def _fwd(m, xi, xs, last):
    with torch.no_grad():
        y = (m.forward_last if last else m)(xi, xs)
    torch.cuda.synchronize()
    return y.cpu().numpy()


@pytest.mark.parametrize("B", [1, 2, 3, 5, 8])
def test_latency1_is_bit_identical_to_the_launch_chain(B):
    m, w = _model()
    x_imu, x_s = synth.make_inputs(synth.PAPER, B, 40, seed=70 + B)
    for T in (1, 2, 7, 16, 17, 33, 40):
        xi, xs = torch.tensor(x_imu[:, :T].copy()).cuda(), torch.tensor(x_s[:, :T].copy()).cuda()
        for last in (False, True):
            m.set_plan("latency")
            ref = _fwd(m, xi, xs, last)
            m.set_plan("latency1")
            got = _fwd(m, xi, xs, last)
            assert np.array_equal(got, ref), (B, T, last, np.abs(got - ref).max())
            for _ in range(3):                                  # flags only grow, the epoch word moves on: no reset between launches
                assert np.array_equal(_fwd(m, xi, xs, last), ref)
        if T == 40:
            yo = oracle.forward(synth.PAPER, w, x_imu, x_s, dtype=np.float64)
            assert np.abs(_fwd(m, xi, xs, False) - yo).max() < TOL
    m.check_handoffs()


def test_auto_keeps_the_launch_chain():
    """latency1 is opt-in (measured no faster than the chain): AUTO's few-stream plan is still "latency" + a separate output
    projection."""
    m, _ = _model()
    for B in (1, 8, 9, 32):                                  # (beyond 32 windows of 40 frames AUTO takes the window-split encoder)
        x_imu, x_s = synth.make_inputs(synth.PAPER, B, 40, seed=3)
        xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
        m.set_plan("auto", profile=1)
        a = _fwd(m, xi, xs, True)
        stages = {n: k for n, _, k in m.profile_read()}
        assert stages.get("latency_chain") == 1 and "out_linear" in stages, stages
        m.set_plan("latency", profile=0)
        assert np.array_equal(a, _fwd(m, xi, xs, True))


def test_latency1_keep_mask_and_nan_scrub(golden):
    case = golden["paper_mask_s0_B2_T40"]
    m, _ = _model(0)
    h = m._ensure_handle()
    m.refresh_packed(torch.device("cuda:0"))
    xi, xs = torch.tensor(case["x_imu"]).cuda(), torch.tensor(case["x_s"]).cuda()
    mask = torch.tensor(case["mask"]).cuda()
    p = float(case["p"][0])
    ws = torch.empty(h.workspace_bytes(2, 40), dtype=torch.uint8, device="cuda")
    outs = []
    for plan in (tlib.TIP_PLAN_LATENCY, tlib.TIP_PLAN_LATENCY1):
        y = torch.zeros(2, 40, 131, device="cuda")
        h.set_option(tlib.TIP_OPT_PLAN, plan)
        h.forward(xi.data_ptr(), xs.data_ptr(), y.data_ptr(), 2, 40, tlib.TIP_FWD_KEEP_MASK, mask.data_ptr(), 1.0 / (1.0 - p),
                  ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        outs.append(y.cpu().numpy())
    assert np.array_equal(outs[0], outs[1])
    assert np.abs(outs[1] - case["y64"]).max() < TOL


def test_latency1_under_hip_graph_replay():
    """Kernel arguments are frozen in a captured graph: the barrier protocol must carry its launch number on the device."""
    m, _ = _model()
    m.set_plan("latency1")
    x_imu, x_s = synth.make_inputs(synth.PAPER, 1, 40, seed=11)
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    ws = torch.empty(m.workspace_bytes(1, 40), dtype=torch.uint8, device="cuda")
    out = torch.empty(1, 131, device="cuda")
    with torch.no_grad():
        ref = m.forward_last(xi, xs).clone()
        m.forward_last(xi, xs, workspace=ws, out=out)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            m.forward_last(xi, xs, workspace=ws, out=out)
        for i in range(200):
            if i % 50 == 0:
                xi.add_(0.01)                                   # new inputs through the static buffers
                ref = None
            g.replay()
            if ref is None:
                torch.cuda.synchronize()
                m.set_plan("latency")
                ref = m.forward_last(xi, xs).clone()
                m.set_plan("latency1")
        torch.cuda.synchronize()
        assert torch.equal(out, ref)
    m.check_handoffs()


@pytest.mark.handoff_fault
def test_latency1_lost_handoff_poisons_and_recovers():
    m, _ = _model()
    h = m._ensure_handle()
    h.set_option(tlib.TIP_OPT_AUTO_DEMOTE, 0)
    m.set_plan("latency1")
    x_imu, x_s = synth.make_inputs(synth.PAPER, 3, 40, seed=5)
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    ref = _fwd(m, xi, xs, False)
    t0 = tlib.spin_timeouts()
    h.set_option(tlib.TIP_OPT_FAULT_INJECT, 4)                  # member 1 of stream 0's recurrence cluster never computes
    y = _fwd(m, xi, xs, False)
    h.set_option(tlib.TIP_OPT_FAULT_INJECT, 0)
    assert tlib.spin_timeouts() > t0
    bad = np.isnan(y)
    assert bad[0].all() and not bad[1:].any() and np.array_equal(y[1:], ref[1:])
    with pytest.raises(tlib.TipHandoffError):
        m(xi, xs)
    h.check_clear()
    assert np.array_equal(_fwd(m, xi, xs, False), ref)          # the flag words are re-zeroed after a reported failure
    m.check_handoffs()

"""GPU: self-demotion after a lost inter-workgroup hand-off (include/tip_hip.h: TIP_OPT_AUTO_DEMOTE / TIP_OPT_DEMOTED).

The default plans launch cooperating kernels; a co-tenant holding CUs can starve a partner workgroup.  The launch that loses a
hand-off yields NaN rows and a sticky flag (tests/test_handoff_fault_gpu.py).  What happens NEXT is tested here: the first
TipHandoffError of a handle makes the Python host clear the flag, switch the handle to the plans that need no co-resident
workgroups (hybrid one-window encoder + single-workgroup recurrence tiles: `fusedh` + `rnn_cluster=1`), re-issue the call and warn
once — with the fault STILL injected, every later forward is finite and bit-identical to the explicit non-cooperating plan."""
import warnings

import numpy as np
import pytest
import torch

from tip_amd import synth, streaming
from tip_amd import lib as tlib
from oracle import oracle
from test_host_cpu import make_model, load_synth

pytestmark = pytest.mark.gpu


def _model():
    m = make_model(synth.PAPER)
    w = load_synth(m, synth.PAPER, 0)
    return m.cuda().eval(), w


@pytest.mark.handoff_fault
@pytest.mark.parametrize("B,bits", [(100, 2), (3, 4), (300, 2)])   # clustered recurrence; latency plan's GEMV recurrence; two rounds
def test_lost_handoff_demotes_and_the_next_call_runs(B, bits):
    m, w = _model()
    h = m._ensure_handle()
    x_imu, x_s = synth.make_inputs(synth.PAPER, B, 40, seed=9)
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    with torch.no_grad():
        ref_auto = m(xi, xs).cpu().numpy()
        # the explicit non-cooperating plan AUTO falls back to: hybrid one-window encoder, or two windows per workgroup when that
        # needs fewer rounds (B = 300 on 256 CUs), with single-workgroup recurrence tiles
        m.set_plan("fused2" if B == 300 else "fusedh", rnn_cluster=1)
        ref_safe = m(xi, xs).cpu().numpy()
        m.set_plan("auto")
        yo = oracle.forward(synth.PAPER, w, x_imu[:2], x_s[:2], dtype=np.float64)
        assert np.abs(ref_safe[:2] - yo).max() < 2e-5 and np.abs(ref_auto - ref_safe).max() < 5e-6
        h.set_option(tlib.TIP_OPT_FAULT_INJECT, bits)         # from here on every cooperating launch loses a hand-off
        y_bad = m(xi, xs)
        torch.cuda.synchronize()
        assert bool(torch.isnan(y_bad).any()) and not m.is_demoted()
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            y1 = m(xi, xs)                                    # entry check trips -> demote -> this call runs
            torch.cuda.synchronize()
        assert any("lost an inter-workgroup hand-off" in str(r.message) for r in rec)
        assert m.is_demoted() and m.demotions == 1
        assert np.array_equal(y1.cpu().numpy(), ref_safe), "the demoted AUTO plan is not the explicit fusedh + cluster-1 plan"
        for _ in range(3):                                    # the fault is still injected: nothing cooperating runs any more
            assert np.array_equal(m(xi, xs).cpu().numpy(), ref_safe)
        assert np.array_equal(m.forward_last(xi, xs).cpu().numpy(), ref_safe[:, -1])
        m.check_handoffs()                                    # clean
        # back to the default plans once the GPU is ours again
        h.set_option(tlib.TIP_OPT_FAULT_INJECT, 0)
        m.undemote()
        assert np.array_equal(m(xi, xs).cpu().numpy(), ref_auto)
        m.check_handoffs()


@pytest.mark.handoff_fault
def test_auto_demote_off_reports_the_error():
    m, _ = _model()
    h = m._ensure_handle()
    h.set_option(tlib.TIP_OPT_AUTO_DEMOTE, 0)
    x_imu, x_s = synth.make_inputs(synth.PAPER, 100, 40, seed=9)    # AUTO: hybrid encoder + clustered recurrence
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    with torch.no_grad():
        m(xi, xs)
        h.set_option(tlib.TIP_OPT_FAULT_INJECT, 2)
        m(xi, xs)
        torch.cuda.synchronize()
        h.set_option(tlib.TIP_OPT_FAULT_INJECT, 0)
        with pytest.raises(tlib.TipHandoffError):
            m(xi, xs)
        assert not m.is_demoted()
        h.check_clear()
        assert bool(torch.isfinite(m(xi, xs)).all())


@pytest.mark.handoff_fault
def test_graph_mode_streaming_engine_polls_the_handoff_word():
    """a HIP-graph replay never re-enters tip_forward, so its entry check cannot report a lost hand-off.  The engine
    polls the pinned word before every replay: the frame after the lost one raises, the engine is re-primed and the handle
    demoted, and the stream then runs on (re-captured under the non-cooperating plan) with finite outputs."""
    m, _ = _model()
    h = m._ensure_handle()
    n = 2
    rng = np.random.RandomState(3)
    s_init = torch.zeros(n, 114)
    eng = streaming.StreamingEngine(m, s_init, use_graph=True)

    def frame():
        R = np.tile(np.eye(3).reshape(-1), (n, 6)).astype(np.float32)
        acc = rng.randn(n, 18).astype(np.float32)
        return torch.tensor(np.concatenate([R, acc], axis=1)).cuda()

    for _ in range(50):                                        # prime + capture + a few replays
        out = eng.step(frame())
    torch.cuda.synchronize()
    assert eng._graph is not None and bool(torch.isfinite(out["y_last"]).all())
    ws_ptr = eng._graph_refs[0].data_ptr()
    m.release_buffers()                                        # the module's evictable buffers are not what the graph points at
    out = eng.step(frame())
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out["y_last"]).all()) and eng._graph_refs[0].data_ptr() == ws_ptr
    # a launch of this handle loses a hand-off (kernel arguments of the captured graph are frozen, so the fault is injected
    # through a direct call on the same handle: the word the replays would set is the same word)
    x_imu, x_s = synth.make_inputs(synth.PAPER, n, 40, seed=2)
    h.set_option(tlib.TIP_OPT_FAULT_INJECT, 4)                # latency plan's GEMV recurrence (B = 2)
    with torch.no_grad():
        m.forward_last(torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda())
    torch.cuda.synchronize()
    with pytest.raises(tlib.TipHandoffError):
        eng.step(frame())                                      # the poll before the replay
    assert eng.frame == 0 and eng._graph is None and m.is_demoted()
    for _ in range(60):                                        # re-primed, demoted, fault still injected: runs clean
        out = eng.step(frame())
    torch.cuda.synchronize()
    assert out is not None and bool(torch.isfinite(out["y_last"]).all()) and eng._graph is not None
    m.check_handoffs()
    h.set_option(tlib.TIP_OPT_FAULT_INJECT, 0)


@pytest.mark.handoff_fault
def test_launch_by_launch_streaming_engine_raises_and_reprimes_after_a_lost_frame():
    """in the launch-by-launch engine the frame that lost a hand-off has already been consumed (its NaN y_last
    is in the history ring) when the NEXT forward's entry check demotes the model and serves the call.  step() must not carry on on
    top of that row: it resets the engine and raises, like the graph path; the stream then runs on demoted, finite."""
    m, _ = _model()
    h = m._ensure_handle()
    n = 2
    rng = np.random.RandomState(5)
    eng = streaming.StreamingEngine(m, torch.zeros(n, 114), use_graph=False)

    def frame():
        R = np.tile(np.eye(3).reshape(-1), (n, 6)).astype(np.float32)
        return torch.tensor(np.concatenate([R, rng.randn(n, 18).astype(np.float32)], axis=1)).cuda()

    for _ in range(50):
        out = eng.step(frame())
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out["y_last"]).all()) and not m.is_demoted()
    h.set_option(tlib.TIP_OPT_FAULT_INJECT, 4)                # latency plan's GEMV recurrence (B = 2): this frame is lost
    out = eng.step(frame())
    torch.cuda.synchronize()
    assert bool(torch.isnan(out["y_last"]).any())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with pytest.raises(tlib.TipHandoffError):
            eng.step(frame())                                  # entry check trips, the model demotes, the engine re-primes and raises
    assert eng.frame == 0 and m.is_demoted() and m.demotions == 1
    for _ in range(60):                                        # fault still injected: the demoted plans do not care
        out = eng.step(frame())
    torch.cuda.synchronize()
    assert out is not None and out["T"] == 40 and bool(torch.isfinite(out["y_last"]).all())
    m.check_handoffs()
    h.set_option(tlib.TIP_OPT_FAULT_INJECT, 0)


@pytest.mark.handoff_fault
def test_reuse_streaming_engine_raises_demotes_and_reprimes_after_a_lost_frame():
    """StreamingEngine(reuse=True): the encoder's reuse form has no hand-off, the clustered recurrence behind it has.  A lost frame's NaN
    row is in the history ring AND in the reuse ring: the next step demotes the handle, clears both (reset()) and raises; the stream
    then runs on — single-workgroup recurrence tiles — finite, with the fault still injected."""
    m, _ = _model()
    h = m._ensure_handle()
    n = 40
    rng = np.random.RandomState(6)
    eng = streaming.StreamingEngine(m, torch.zeros(n, 114), reuse=True)

    def frame():
        R = np.tile(np.eye(3).reshape(-1), (n, 6)).astype(np.float32)
        return torch.tensor(np.concatenate([R, rng.randn(n, 18).astype(np.float32)], axis=1)).cuda()

    for _ in range(50):
        out = eng.step(frame())
    torch.cuda.synchronize()
    assert out["T"] == 40 and bool(torch.isfinite(out["y_last"]).all()) and not m.is_demoted()
    h.set_option(tlib.TIP_OPT_FAULT_INJECT, 2)                # clustered recurrence: this frame is lost
    out = eng.step(frame())
    torch.cuda.synchronize()
    assert bool(torch.isnan(out["y_last"]).any())
    with pytest.raises(tlib.TipHandoffError):
        eng.step(frame())
    assert eng.frame == 0 and m.is_demoted() and m.demotions == 1
    for _ in range(60):
        out = eng.step(frame())
    torch.cuda.synchronize()
    assert out is not None and out["T"] == 40 and bool(torch.isfinite(out["y_last"]).all())
    m.check_handoffs()
    h.set_option(tlib.TIP_OPT_FAULT_INJECT, 0)


@pytest.mark.handoff_fault
def test_train_mode_runner_call_demotes_after_a_lost_handoff():
    """the unedited runner's call (.train() mode, B = 1: tip_forward_dropout on the cooperating latency kernels) had no
    demote flow — after a lost hand-off every following frame raised.  Now as in _forward_hip: the first TipHandoffError clears the
    word, demotes the handle and the call falls back to tip_train_forward (one-workgroup recurrence tiles); later frames are finite."""
    m = make_model(synth.PAPER, p_state=0.8)
    load_synth(m, synth.PAPER, 0)
    m = m.cuda()                                           # stays in .train() mode
    h = m._ensure_handle()
    x_imu, x_s = synth.make_inputs(synth.PAPER, 1, 40, seed=9)
    xi, xs = torch.tensor(x_imu).cuda(), torch.nan_to_num(torch.tensor(x_s)).cuda()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(3):                                 # slow path, then the fast path twice
            assert bool(torch.isfinite(m(xi, xs)).all())
        assert m._fast_state is not None
        h.set_option(tlib.TIP_OPT_FAULT_INJECT, 4)         # the GEMV recurrence loses member 1 of stream 0 from here on
        y_bad = m(xi, xs)
        torch.cuda.synchronize()
        assert bool(torch.isnan(y_bad).any()) and not m.is_demoted()
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        y1 = m(xi, xs)                                     # entry check trips -> demote -> tip_train_forward serves this call
        torch.cuda.synchronize()
    assert any("lost an inter-workgroup hand-off" in str(r.message) for r in rec)
    assert m.is_demoted() and m.demotions == 1 and bool(torch.isfinite(y1).all())
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(3):                                 # the fault is still injected: nothing cooperating runs any more
            assert bool(torch.isfinite(m(xi, xs)).all())
        y1.sum().backward()                                # and the demoted step differentiates
        torch.cuda.synchronize()
        assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())
        m.check_handoffs()
        h.set_option(tlib.TIP_OPT_FAULT_INJECT, 0)
        m.undemote()
        m._backward_seen = False
        assert bool(torch.isfinite(m(xi, xs)).all())
        m.check_handoffs()
    # AUTO_DEMOTE off: the error is reported
    m2 = make_model(synth.PAPER, p_state=0.8)
    load_synth(m2, synth.PAPER, 0)
    m2 = m2.cuda()
    h2 = m2._ensure_handle()
    h2.set_option(tlib.TIP_OPT_AUTO_DEMOTE, 0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m2(xi, xs)
        h2.set_option(tlib.TIP_OPT_FAULT_INJECT, 4)
        m2(xi, xs)
        torch.cuda.synchronize()
        h2.set_option(tlib.TIP_OPT_FAULT_INJECT, 0)
        with pytest.raises(tlib.TipHandoffError):
            m2(xi, xs)


@pytest.mark.handoff_fault
@pytest.mark.parametrize("B", [1, 9])
def test_placement_loss_of_the_one_launch_form_falls_back_to_the_launch_chain(B):
    """The one-launch few-stream form needs every workgroup of a window on the window's XCD; kernels of another stream dispatched
    beside it can break that (tests/test_hip_parity.py: test_one_launch_form_under_foreign_stream_load).  Simulated deterministically
    (TIP_OPT_FAULT_INJECT bit 4: one producer stamps its flag as another XCD's): the launch is NaN + reported with
    TIP_OPT_HANDOFF_KIND = 2, and the host's answer is the MILD one — TIP_OPT_NO_FLOW, the same plan as a launch chain (bit-identical
    outputs, co-residency only) — not the plans without any hand-off; a later loss of another kind still demotes fully."""
    m, w = _model()
    h = m._ensure_handle()
    x_imu, x_s = synth.make_inputs(synth.PAPER, B, 40, seed=11)
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    with torch.no_grad():
        ref = m(xi, xs).cpu().numpy()                          # one-launch form
        h.set_option(tlib.TIP_OPT_NO_FLOW, 1)
        ref_chain = m(xi, xs).cpu().numpy()
        h.set_option(tlib.TIP_OPT_NO_FLOW, 0)
        assert np.array_equal(ref, ref_chain), "the launch chain and the one-launch form share their stage bodies: same bits"
        assert h.get_option(tlib.TIP_OPT_HANDOFF_KIND) == 0
        h.set_option(tlib.TIP_OPT_FAULT_INJECT, 16)
        y_bad = m(xi, xs)
        torch.cuda.synchronize()
        assert bool(torch.isnan(y_bad).any())
        assert h.get_option(tlib.TIP_OPT_HANDOFF_KIND) == 2
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            y1 = m(xi, xs)                                     # entry check trips -> launch chain -> this call runs
            torch.cuda.synchronize()
        assert any("spread over several XCDs" in str(r.message) for r in rec)
        assert m.flow_demotions == 1 and m.demotions == 0 and not m.is_demoted() and h.get_option(tlib.TIP_OPT_NO_FLOW) == 1
        assert np.array_equal(y1.cpu().numpy(), ref)
        for _ in range(3):                                     # the fault is still injected: the chain does not care
            assert np.array_equal(m.forward_last(xi, xs).cpu().numpy(), ref[:, -1])
        m.check_handoffs()
        # a loss of the OTHER kind on top (the chain's GEMV recurrence drops a member): now the full demotion
        h.set_option(tlib.TIP_OPT_FAULT_INJECT, 4)
        y_bad = m(xi, xs)
        torch.cuda.synchronize()
        assert bool(torch.isnan(y_bad).any()) and h.get_option(tlib.TIP_OPT_HANDOFF_KIND) == 1
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            y2 = m(xi, xs)
            torch.cuda.synchronize()
        assert m.is_demoted() and m.demotions == 1 and bool(torch.isfinite(y2).all())
        h.set_option(tlib.TIP_OPT_FAULT_INJECT, 0)
        m.undemote()
        assert h.get_option(tlib.TIP_OPT_NO_FLOW) == 0 and not m.is_demoted()
        assert np.array_equal(m(xi, xs).cpu().numpy(), ref)
        m.check_handoffs()
    # the unedited runner's call (.train() mode: tip_forward_dropout) takes the same mild step
    mt = make_model(synth.PAPER, p_state=0.8)
    load_synth(mt, synth.PAPER, 0)
    mt = mt.cuda()
    ht = mt._ensure_handle()
    xs1 = torch.nan_to_num(xs[:1])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(3):
            assert bool(torch.isfinite(mt(xi[:1], xs1)).all())
        ht.set_option(tlib.TIP_OPT_FAULT_INJECT, 16)
        y_bad = mt(xi[:1], xs1)
        torch.cuda.synchronize()
        assert bool(torch.isnan(y_bad).any())
        y1 = mt(xi[:1], xs1)
        torch.cuda.synchronize()
        assert mt.flow_demotions == 1 and not mt.is_demoted() and bool(torch.isfinite(y1).all())
        for _ in range(3):
            assert bool(torch.isfinite(mt(xi[:1], xs1)).all())
        mt.check_handoffs()
        ht.set_option(tlib.TIP_OPT_FAULT_INJECT, 0)

"""GPU: a hand-off wait that gives up is an ERROR, never a number.

The cooperating plans (pair-split encoder, clustered / GEMV RNN) wait for partner workgroups with bounded spins.  When a
partner never arrives — simulated deterministically with TIP_OPT_FAULT_INJECT, which drops one cooperating workgroup at kernel
entry — the launch must (a) leave NaN, not stale finite values, in every output row the missing data could have reached and
bit-exact values everywhere else, and (b) make the NEXT call raise TipHandoffError (sticky until cleared).  A CU-masked stream
(the co-tenant case) must give either bit-exact results or that error."""
import ctypes

import numpy as np
import pytest
import torch

import tip_amd
from tip_amd import synth
from tip_amd import lib as tlib
from test_host_cpu import make_model, load_synth

pytestmark = pytest.mark.gpu


def _model():
    """(TIP_OPT_AUTO_DEMOTE off: these tests are about the raw error path — NaN rows, sticky TIP_ERR_HANDOFF; what the host does
    with the first error by default is tests/test_demote_gpu.py)"""
    m = make_model(synth.PAPER)
    load_synth(m, synth.PAPER, 0)
    m = m.cuda().eval()
    m._ensure_handle().set_option(tlib.TIP_OPT_AUTO_DEMOTE, 0)
    return m


def _fault(m, bits):
    m._ensure_handle().set_option(tlib.TIP_OPT_FAULT_INJECT, bits)


@pytest.mark.handoff_fault
@pytest.mark.parametrize("plan,B,bits,hit,cluster", [
    ("fused1s2", 40, 1, "win0", 0),     # window-split encoder (AUTO's choice for 33..128 windows): workgroup (window 0, part 1) never arrives
    ("fused", 40, 2, "tile0", 0),       # RNN clusters (AUTO: 4 workgroups per 4-window tile): member 1 of cluster 0 never arrives
    ("fusedh", 40, 2, "tile0", 16),     # the 16-workgroup clusters on 16-window tiles (sentinel hand-off)
    ("general", 37, 2, "tile0", 0),     # AUTO through the general plan
    ("general", 37, 2, "tile0", 8),     # 8-workgroup clusters
    ("latency", 3, 4, "win0", 0),       # GEMV RNN of the latency plan: member 1 of stream 0 never arrives
    ("latency", 3, 1, "win0", 0),       # one-launch form of the latency plan (B <= 8): the out-projection workgroup of column block 1 of
                                        # window 0, layer 0, never arrives — its consumers' flag waits give up and poison the window
    ("latency", 20, 4, "win0", 0),      # the launch chain (B > 8) with the 8-member GEMV recurrence
])
def test_lost_handoff_poisons_and_raises(plan, B, bits, hit, cluster):
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    m = _model()
    m.set_plan(plan, rnn_cluster=cluster)
    x_imu, x_s = synth.make_inputs(synth.PAPER, B, 40, seed=5)
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    with torch.no_grad():
        ref = m(xi, xs).cpu().numpy()
        m.check_handoffs()                                   # healthy so far
        t0 = tlib.spin_timeouts()
        _fault(m, bits)
        y = m(xi, xs)                                        # this launch loses a hand-off: the call itself returns normally
        torch.cuda.synchronize()
        y = y.cpu().numpy()
        _fault(m, 0)
        assert tlib.spin_timeouts() > t0, "the injected fault did not make any wait give up"
        # never finite-but-wrong: every element is either NaN or exactly the healthy value
        bad = np.isnan(y)
        assert bad.any(), "a lost hand-off left no trace in the output"
        assert np.array_equal(y[~bad], ref[~bad])
        if hit == "pair0":
            assert bad[0].all() and bad[1].all() and not bad[2:].any()      # both windows of pair 0, nothing else
        elif hit == "tile0":
            nt = min(4 if cluster == 0 else 16, B)                            # windows per RNN tile (tip_general.hip: rnn_rows4_kernel / rnn_resident_kernel)
            assert bad[:nt].all()                                            # the whole tile (row 0 too: the slice of h_0 that never came)
            assert not bad[nt:].any()                                        # other window tiles are untouched
        else:
            assert bad[0].all() and not bad[1:].any()
        # the NEXT call reports it, and keeps reporting it until cleared
        with pytest.raises(tlib.TipHandoffError):
            m(xi, xs)
        with pytest.raises(tlib.TipHandoffError):
            m.check_handoffs()
        try:
            m._ensure_handle().check(clear=True)
        except tlib.TipHandoffError:
            pass
        m.check_handoffs()                                   # cleared
        y2 = m(xi, xs).cpu().numpy()                          # and the handle works again
        assert np.array_equal(y2, ref)


@pytest.mark.handoff_fault
def test_training_step_reports_a_lost_handoff():
    """tip_train_forward shares the clustered RNN: a lost hand-off there poisons y and makes the next step raise."""
    m = _model().train()
    x_imu, x_s = synth.make_inputs(synth.PAPER, 36, 40, seed=6)   # (> LAZY_STASH_MAX_BATCH: fewer windows take tip_forward_dropout)
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    _fault(m, 2)
    y = m(xi, xs)
    torch.cuda.synchronize()
    _fault(m, 0)
    assert torch.isnan(y[:4, 1:]).all() and torch.isfinite(y[4:]).all()      # the 4-window tile of the cluster that lost a member
    with pytest.raises(tlib.TipHandoffError):
        m(xi, xs)
    try:
        m._ensure_handle().check(clear=True)
    except tlib.TipHandoffError:
        pass
    y = m(xi, xs)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()


def _masked_stream(hip, ncus_enabled):
    mask = (ctypes.c_uint32 * 8)()
    per_xcd = ncus_enabled // 8                                # CU bit i -> (XCD i % 8, CU i / 8): keep the XCDs balanced
    for i in range(256):
        if (i // 8) < per_xcd:
            mask[i // 32] |= 1 << (i % 32)
    stream = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(stream), 8, mask)
    if rc != 0:
        pytest.skip(f"hipExtStreamCreateWithCUMask failed ({rc})")
    return stream


@pytest.mark.parametrize("ncus_enabled", [200, 128, 64])
def test_cu_masked_stream_is_exact(ncus_enabled):
    """The co-tenant / partitioned-GPU case: the forward runs on a stream that may use only some of the CUs.  The library reads
    the stream's CU mask (effective_cus, tip_internal.h) and sizes plan selection, grids and — above all — the cooperating
    clusters of the recurrence for THAT many CUs, so every member of a cluster is resident and the result is bit-exact (round
    2 sized them for the whole device: the outcome then was "exact, or NaN rows + TipHandoffError").  With 128 CUs AUTO must
    also re-cost its encoder choice: with 128 CUs 256 windows are ONE round of the two-window kernel (2 x 128) against two
    rounds of the one-window kernel — bit-identical to the explicit "fused2" plan, where the unmasked stream takes "fusedh"."""
    hip = ctypes.CDLL("libamdhip64.so")
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    if ncu < 256:
        pytest.skip("written for the 256-CU part")
    stream = _masked_stream(hip, ncus_enabled)
    try:
        m = _model()
        m.set_plan("auto")
        B = 256
        x_imu, x_s = synth.make_inputs(synth.PAPER, B, 40, seed=8)
        xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
        with torch.no_grad():
            ref = m(xi, xs)                       # unmasked: fusedh
            m.set_plan("fused2")
            ref2 = m(xi, xs)
            m.set_plan("auto")
            torch.cuda.synchronize()
            ext = torch.cuda.ExternalStream(stream.value)
            for _ in range(3):
                with torch.cuda.stream(ext):
                    y = m(xi, xs)
                ext.synchronize()
                assert bool(torch.isfinite(y).all()), "a cluster member was not resident under the CU mask"
                # 200 / 128 / 64 usable CUs: 1 / 1 / 2 rounds of the two-window kernel against 2 / 2 / 4 of the one-window kernel
                assert torch.equal(y, ref2), "AUTO did not re-cost its plan for the masked CU count"
                assert not torch.equal(y, ref)
            m.check_handoffs()
    finally:
        torch.cuda.synchronize()
        hip.hipStreamDestroy(stream)


@pytest.mark.parametrize("plan,B", [("auto", 60), ("fused1s4", 30), ("fused1s2", 64)])
def test_cu_masked_stream_window_split(plan, B):
    """The window-split encoder (one window on two / four workgroups) on a stream limited to 128 CUs: the partners of a window must
    all be resident (the launchers size the check for the stream's CUs) and are not necessarily on one XCD any more (checked at run
    time: the cross-XCD path then carries the partial sums) — bit-identical to the unmasked run either way.  AUTO at 60 windows:
    four workgroups per window do not fit 128 CUs, two do."""
    hip = ctypes.CDLL("libamdhip64.so")
    if torch.cuda.get_device_properties(0).multi_processor_count < 256:
        pytest.skip("written for the 256-CU part")
    stream = _masked_stream(hip, 128)
    try:
        m = _model()
        x_imu, x_s = synth.make_inputs(synth.PAPER, B, 40, seed=18)
        xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
        with torch.no_grad():
            m.set_plan("fused1s2" if plan == "auto" else plan)
            ref = m(xi, xs)
            torch.cuda.synchronize()
            m.set_plan(plan)
            ext = torch.cuda.ExternalStream(stream.value)
            for _ in range(3):
                with torch.cuda.stream(ext):
                    y = m(xi, xs)
                ext.synchronize()
                assert bool(torch.isfinite(y).all())
                assert torch.equal(y, ref)
            m.check_handoffs()
    finally:
        torch.cuda.synchronize()
        hip.hipStreamDestroy(stream)


@pytest.mark.parametrize("plan,B,cluster", [("fused1s4", 20, 0), ("fused1s2", 40, 0), ("fusedh", 256, 0), ("fusedh", 100, 0),
                                            ("fused", 64, 16), ("fused", 64, 4), ("latency", 5, 0)])
def test_cross_xcd_paths_are_bit_identical(plan, B, cluster):
    """TIP_OPT_FAULT_INJECT bit 3: every cooperating kernel (window-split encoder, four-window and 16-window recurrence
    clusters, the latency plan's GEMV cluster) treats its partners as sitting on different XCDs — agent-scope stores, L1-bypassing loads,
    paced polls — wherever they really are.  That is the path a placement across XCDs takes (a CU-masked stream, a partitioned part);
    on an idle full part the dispatcher never produces it, so it is forced here: bit-identical results, no time-out."""
    m = _model()
    h = m._ensure_handle()
    x_imu, x_s = synth.make_inputs(synth.PAPER, B, 40, seed=28)
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    t0 = tlib.spin_timeouts()
    with torch.no_grad():
        m.set_plan(plan, rnn_cluster=cluster)
        ref = m(xi, xs)
        refl = m.forward_last(xi, xs)
        torch.cuda.synchronize()
        h.set_option(tlib.TIP_OPT_FAULT_INJECT, 8)
        try:
            for _ in range(3):
                y = m(xi, xs)
                yl = m.forward_last(xi, xs)
                torch.cuda.synchronize()
                assert torch.equal(y, ref) and torch.equal(yl, refl)
        finally:
            h.set_option(tlib.TIP_OPT_FAULT_INJECT, 0)
    assert tlib.spin_timeouts() == t0
    m.check_handoffs()

"""GPU tests of the exact streaming reuse (include/tip_hip.h: tip_forward_reuse, csrc/tip_fused2.hip:
reuse_update_kernel + the ring-reading instantiation of fused_encoder2_kernel; model.forward_last_reuse, StreamingEngine(reuse=True)).

What the reference guarantees (real_time_runner_minimal.py:74,85,137): a frame's model inputs never change once recorded, so its
in_linear row and layer-0 Q / K / V rows are the same in each of the 40 windows it appears in.  The bar here is therefore
BIT-IDENTITY with the engine that recomputes them (two-window encoder, TIP_PLAN_FUSED2 — AUTO's own plan at 1024 streams), over the
trace of the real reference runner and over random closed loops; a ring that does not hold the window's frames must give NaN."""
import ctypes
import os

import numpy as np
import pytest
import torch

import tip_amd
from tip_amd import synth
from tip_amd import lib as tlib
from conftest import ROOT
from test_host_cpu import make_model, load_synth

pytestmark = pytest.mark.gpu
RUNNER_GOLDEN = os.path.join(ROOT, "tests", "golden", "tip_runner_golden.npz")


def _model():
    cfg = synth.PAPER
    m = make_model(cfg)
    load_synth(m, cfg, 0)
    return m.cuda().eval()


def _raw_frames(B, F, seed):
    from scipy.spatial.transform import Rotation
    rng = np.random.RandomState(seed)
    raw = np.zeros((F, B, 72), dtype=np.float32)
    for f in range(F):
        raw[f, :, :54] = Rotation.random(B * 6, random_state=1000 * seed + f).as_matrix().reshape(B, 54)
        raw[f, :, 54:] = rng.randn(B, 18)
    return raw, rng.randn(B, 114).astype(np.float32) * 0.2


def _lockstep(m, ref, eng, raw, plan_full="fused2"):
    """ref recomputes every window (plan AUTO while the windows grow, `plan_full` once they are full: the reuse form is the two-window
    encoder's whatever the batch); eng reads the ring.  Every output of every frame must be the same bits."""
    lib = tlib.load()
    full = 0
    for f in range(raw.shape[0]):
        m.set_plan(plan_full if lib.tip_stream_window_len(f) == 40 else "auto")
        a, b = ref.step(raw[f]), eng.step(raw[f])
        assert (a is None) == (b is None)
        if a is None:
            continue
        for k in ("s_rest", "c_t", "y_last"):
            assert torch.equal(a[k], b[k]), (f, k, float((a[k] - b[k]).abs().max()))
        assert torch.isfinite(b["y_last"]).all()
        full += a["T"] == 40
    m.set_plan("auto")
    return full


def test_reuse_engine_is_bit_identical_over_the_reference_runner_trace():
    z = np.load(RUNNER_GOLDEN)
    tr = [{k.split("/")[1]: z[k] for k in z.files if k.startswith(f"stream{i}/")} for i in range(2)]
    m = _model()
    s_init = np.stack([t["s_init"] for t in tr])
    raw = np.stack([np.stack([t["raw_imu"][f] for t in tr]) for f in range(70)]).astype(np.float32)
    ref = tip_amd.streaming.StreamingEngine(m, s_init)
    eng = tip_amd.streaming.StreamingEngine(m, s_init, reuse=True)
    n0 = m.hip_forward_count()
    assert _lockstep(m, ref, eng, raw) == 26          # frames 44 .. 69 run on full windows
    assert m.hip_forward_count() == n0 + 2 * 65
    # ... and the closed loop still tracks the reference runner (same bound as tests/test_streaming_gpu.py)
    eng.reset()
    for f in range(70):
        out = eng.step(raw[f])
        if out is not None:
            for b, t in enumerate(tr):
                assert np.abs(out["s_rest"][b].cpu().numpy() - t["qdq"][f][3:]).max() < 5e-4, (f, b)
    m.check_handoffs()


@pytest.mark.parametrize("B", [1, 5, 64, 515])   # odd counts: the last workgroup carries ONE window; 515 > 2 x 256: workgroups loop over pairs
def test_reuse_engine_equals_recomputation_random_closed_loop(B):
    m = _model()
    raw, s_init = _raw_frames(B, 100, 7 + B)
    ref = tip_amd.streaming.StreamingEngine(m, s_init)
    eng = tip_amd.streaming.StreamingEngine(m, s_init, reuse=True)
    assert _lockstep(m, ref, eng, raw) == 56          # more than one trip round the 40-slot ring
    # reset() forgets the ring: the second pass is the first one again
    ref.reset()
    eng.reset()
    assert _lockstep(m, ref, eng, raw[:50]) == 6
    m.check_handoffs()


def test_reuse_at_1024_streams_equals_the_auto_engine():
    """BASELINE configs[2]: at 1024 streams AUTO itself takes the two-window encoder for full windows, so the reuse engine must
    reproduce the DEFAULT engine bit for bit with no plan forced anywhere."""
    m = _model()
    raw, s_init = _raw_frames(1024, 60, 3)
    ref = tip_amd.streaming.StreamingEngine(m, s_init)
    eng = tip_amd.streaming.StreamingEngine(m, s_init, reuse=True)
    assert _lockstep(m, ref, eng, raw, plan_full="auto") == 16


@pytest.mark.parametrize("B", [3, 512])
def test_reuse_graph_mode_equals_launch_by_launch(B):
    """use_graph=True + reuse=True: the frame index of the ring comes from the counter the ingest kernel keeps in the state buffer
    (kernel arguments are frozen in the captured graph)."""
    m = _model()
    raw, s_init = _raw_frames(B, 110, 11)
    ref = tip_amd.streaming.StreamingEngine(m, s_init, reuse=True)
    eng = tip_amd.streaming.StreamingEngine(m, s_init, use_graph=True, reuse=True)
    for f in range(raw.shape[0]):
        a, b = ref.step(raw[f]), eng.step(raw[f])
        if a is None:
            assert b is None
            continue
        for k in ("s_rest", "c_t", "y_last"):
            assert torch.equal(a[k], b[k]), (f, k)
    assert eng._graph is not None
    m.check_handoffs()


def test_ring_that_does_not_hold_the_window_gives_nan_not_numbers():
    m = _model()
    B = 6
    g = torch.Generator().manual_seed(5)
    frames_i = torch.randn(60, B, 90, generator=g).cuda()
    frames_s = torch.randn(60, B, 131, generator=g).cuda()

    def window(c):                      # frames c-39 .. c
        return frames_i[c - 39: c + 1].transpose(0, 1).contiguous(), frames_s[c - 39: c + 1].transpose(0, 1).contiguous()

    m.set_plan("fused2")
    ring = m.reuse_cache(B)
    with torch.no_grad():
        # prime: frames 0 .. 38 as growing windows, then full windows
        for c in range(39):
            xi, xs = frames_i[: c + 1].transpose(0, 1).contiguous(), frames_s[: c + 1].transpose(0, 1).contiguous()
            m.set_plan("auto")
            m.forward_last_reuse(xi, xs, ring, c)
        m.set_plan("fused2")
        for c in range(39, 45):
            y = m.forward_last_reuse(*window(c), ring, c)
            assert torch.equal(y, m.forward_last(*window(c))), c
        # a skipped frame: slot (46 mod 40) was written, frame 45 never was -> its slot still holds frame 5
        y = m.forward_last_reuse(*window(46), ring, 46)
        assert torch.isnan(y).all()
        # ... and it stays wrong until 40 consecutive frames are in again
        for c in range(47, 60):
            assert torch.isnan(m.forward_last_reuse(*window(c), ring, c)).all()
        # a cleared ring
        m.reuse_reset(ring)
        assert torch.isnan(m.forward_last_reuse(*window(50), ring, 50)).all()
    m.set_plan("auto")
    # bad arguments are errors, not launches
    lib = tlib.load()
    h = m._ensure_handle()
    xi, xs = window(50)
    y = torch.empty(B, 131, device="cuda")
    ws = torch.empty(m.workspace_bytes(B, 40), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def call(T=40, flags=tlib.TIP_FWD_LAST_ROW_ONLY, frame=50, cache_bytes=None):
        return lib.tip_forward_reuse(h._h, xi.data_ptr(), xs.data_ptr(), y.data_ptr(), B, T, flags, ring.data_ptr(),
                                     ring.numel() if cache_bytes is None else cache_bytes, frame, None, ws.data_ptr(), ws.numel(), st)

    assert call() == 0
    assert call(flags=tlib.TIP_FWD_LAST_ROW_ONLY | tlib.TIP_FWD_KEEP_MASK) == -1      # TIP_ERR_INVALID_ARG: no keep mask with reuse
    assert call(frame=38) == -1                                                       # a full window has 39 earlier frames
    assert call(frame=-3) == -1
    assert call(cache_bytes=ring.numel() - 4096) == -4                                # TIP_ERR_WORKSPACE
    assert call(T=41) == tlib.TIP_ERR_UNSUPPORTED_CONFIG
    torch.cuda.synchronize()


def test_reuse_is_refused_where_it_would_not_be_exact():
    cfg = synth.PAPER
    m = make_model(cfg, p_state=0.8)        # the shipped loaders' past_state_dropout (offline_testing_simple.py:93): rows differ per window
    load_synth(m, cfg, 0)
    m = m.cuda().eval()
    with pytest.raises(RuntimeError, match="past_state_dropout"):
        tip_amd.streaming.StreamingEngine(m, np.zeros((2, 114), np.float32), reuse=True)
    m2 = _model().train()
    with pytest.raises(RuntimeError, match="eval"):
        tip_amd.streaming.StreamingEngine(m2, np.zeros((2, 114), np.float32), reuse=True)
    m3 = _model()
    ring = m3.reuse_cache(2)
    xi, xs = torch.zeros(2, 40, 90, device="cuda"), torch.zeros(2, 40, 131, device="cuda")
    with pytest.raises(RuntimeError, match="no_grad"):
        m3.forward_last_reuse(xi, xs, ring, 39)          # autograd recording
    # a parameter update while a ring is live: the rows in it belong to the old weights
    with torch.no_grad():
        m3.forward_last_reuse(xi[:, :1], xs[:, :1], ring, 0)
        m3.in_linear.weight.mul_(1.5)
        with pytest.raises(RuntimeError, match="parameters changed"):
            m3.forward_last_reuse(xi[:, :2], xs[:, :2], ring, 1)
        # ... also when somebody else re-packed the image in between (a plain forward after the update): the ring is tied to the
        # packed image it was filled under, not to this call noticing the change itself
        m3.forward_last_reuse(xi[:, :1], xs[:, :1], ring, 0)          # (cleared ring, current image: fine)
        m3.in_linear.weight.mul_(0.5)
        m3(xi, xs)                                                     # re-packs
        with pytest.raises(RuntimeError, match="parameters changed"):
            m3.forward_last_reuse(xi[:, :2], xs[:, :2], ring, 1)
        with pytest.raises(RuntimeError, match="feature widths"):
            m3.forward_last_reuse(xi[:, :1, :80], xs[:, :1], ring, 0)


def test_reuse_auto_engages_only_where_it_pays_and_is_exact():
    """reuse="auto": on where the two-window encoder is AUTO's plan anyway (>= two windows per CU) and the model has no stochastic
    part; off (silently: it is an optimisation, not a request) everywhere else."""
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    m = _model()
    z = lambda n: np.zeros((n, 114), np.float32)   # noqa: E731
    assert tip_amd.streaming.StreamingEngine(m, z(4 * cus), reuse="auto").reuse is True
    assert tip_amd.streaming.StreamingEngine(m, z(2 * cus), reuse="auto").reuse is True
    assert tip_amd.streaming.StreamingEngine(m, z(3 * cus), reuse="auto").reuse is False    # three rounds of the one-window kernel win
    assert tip_amd.streaming.StreamingEngine(m, z(cus), reuse="auto").reuse is False
    assert tip_amd.streaming.StreamingEngine(m, z(3), reuse="auto").reuse is False
    cfg = synth.PAPER
    mp = make_model(cfg, p_state=0.8)
    load_synth(mp, cfg, 0)
    assert tip_amd.streaming.StreamingEngine(mp.cuda().eval(), z(4 * cus), reuse="auto").reuse is False


def test_reuse_on_a_demoted_handle_and_with_an_explicit_rnn_cluster():
    """The reuse form only replaces the encoder: whatever the handle says about the recurrence (TIP_OPT_DEMOTED after a lost hand-off:
    single-workgroup tiles; an explicit cluster size) applies to both engines alike, and they stay bit-identical."""
    m = _model()
    raw, s_init = _raw_frames(40, 60, 21)
    h = m._ensure_handle()
    for setup in ("demoted", "cluster2"):
        if setup == "demoted":
            h.set_option(tlib.TIP_OPT_DEMOTED, 1)
        ref = tip_amd.streaming.StreamingEngine(m, s_init)
        eng = tip_amd.streaming.StreamingEngine(m, s_init, reuse=True)
        lib = tlib.load()
        for f in range(raw.shape[0]):
            full = lib.tip_stream_window_len(f) == 40
            m.set_plan("fused2" if full else "auto", rnn_cluster=2 if setup == "cluster2" else 0)
            a, b = ref.step(raw[f]), eng.step(raw[f])
            if a is not None:
                assert torch.equal(a["y_last"], b["y_last"]) and torch.equal(a["s_rest"], b["s_rest"]), (setup, f)
        h.set_option(tlib.TIP_OPT_DEMOTED, 0)
        m.set_plan("auto")
    m.check_handoffs()

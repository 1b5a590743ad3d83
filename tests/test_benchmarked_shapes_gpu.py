"""The exact launches bench.py's `extra.configs` time, against the oracle (round-2 review, item 1b).

`b1024_last_row` / `streams1024_closed_loop` run AUTO at B = 1024 (fused_encoder2_kernel + rnn_rows4_kernel<4> + the last-row
head); `b8192_full` runs AUTO at B = 8192 (several rounds per CU, 2 048 recurrence tiles).  Until now these shapes were
covered only through chains of bit-identities between plans; here they are run as AUTO picks them and compared with the
fp64 oracle (pinned to the reference, /root/reference/simple_transformer_with_state.py:60-102) on sampled windows — every
window is independent of its batch neighbours, which is itself asserted bit for bit."""
import numpy as np
import pytest
import torch

from tip_amd import synth
from tip_amd import lib as tlib
from oracle import oracle
from test_host_cpu import make_model, load_synth

pytestmark = pytest.mark.gpu
TOL_TIGHT = 2e-5


def _gpu_model(seed):
    m = make_model(synth.PAPER)
    w = load_synth(m, synth.PAPER, seed)
    return m.cuda().eval(), w


def _fwd(m, xi, xs, last=False):
    n0 = m.hip_forward_count()
    with torch.no_grad():
        y = (m.forward_last if last else m)(xi, xs)
    torch.cuda.synchronize()
    assert m.hip_forward_count() == n0 + 1, "the HIP path did not run"
    return y


def _inputs(B, seed):
    """B windows as independent 1024-window draws (synth.make_inputs is O(B) numpy work: keep each draw small)."""
    xi, xs = [], []
    for k in range((B + 1023) // 1024):
        a, b = synth.make_inputs(synth.PAPER, min(1024, B - 1024 * k), 40, seed=seed + k)
        xi.append(a)
        xs.append(b)
    return np.concatenate(xi), np.concatenate(xs)


@pytest.mark.parametrize("plan", ["auto", "fused2"])
def test_b1024_full_and_last_row_vs_oracle(plan):
    cfg = synth.PAPER
    m, w = _gpu_model(0)
    m.set_plan(plan)
    B = 1024
    x_imu, x_s = _inputs(B, 4100)
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    y = _fwd(m, xi, xs).cpu().numpy()
    yl = _fwd(m, xi, xs, last=True).cpu().numpy()
    assert y.shape == (B, 40, 131) and yl.shape == (B, 131)
    assert np.isfinite(y).all()
    assert np.array_equal(yl, y[:, -1]), "last-row output differs from row T-1 of the full output"
    sel = np.array([0, 1, 2, 3, 255, 256, 257, 511, 512, 513, 767, 768, 1000, 1021, 1022, 1023])
    yo = oracle.forward(cfg, w, x_imu[sel], x_s[sel], dtype=np.float64)
    e = np.abs(y[sel] - yo).max()
    assert e < TOL_TIGHT, e
    # batch independence, bit for bit: the sampled windows alone (another plan shape: 16 windows), and in halves
    m.set_plan("fused2")
    ysub = _fwd(m, xi[torch.tensor(sel).cuda()], xs[torch.tensor(sel).cuda()]).cpu().numpy()
    assert np.array_equal(ysub, y[sel])
    m.set_plan(plan)
    ya, yb = _fwd(m, xi[:512], xs[:512], last=True), _fwd(m, xi[512:], xs[512:], last=True)
    assert np.array_equal(torch.cat([ya, yb]).cpu().numpy(), yl)
    assert np.array_equal(_fwd(m, xi, xs, last=True).cpu().numpy(), yl), "run-to-run difference"


def test_b8192_auto_vs_oracle_and_shards():
    """BASELINE configs[3] on ONE GPU (bench `b8192_full`) and as its 8 per-GPU shards of 1024 (what `--gpus 8 --config
    streams1024` runs): the concatenation of the shards is the full batch bit for bit, and sampled windows match the oracle."""
    cfg = synth.PAPER
    m, w = _gpu_model(1)
    m.set_plan("auto")
    B = 8192
    x_imu, x_s = _inputs(B, 8100)
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    y = _fwd(m, xi, xs)
    assert bool(torch.isfinite(y).all())
    yl = _fwd(m, xi, xs, last=True)
    assert torch.equal(yl, y[:, -1])
    for r in range(8):
        sl = slice(1024 * r, 1024 * (r + 1))
        assert torch.equal(_fwd(m, xi[sl], xs[sl]), y[sl]), f"shard {r} differs from the full batch"
    sel = np.array([0, 1, 511, 1023, 1024, 2047, 2048, 3000, 4095, 4096, 5000, 6143, 6144, 7777, 8190, 8191])
    yo = oracle.forward(cfg, w, x_imu[sel], x_s[sel], dtype=np.float64)
    e = np.abs(y[torch.tensor(sel).cuda()].cpu().numpy() - yo).max()
    assert e < TOL_TIGHT, e


def test_batch_beyond_the_32bit_descriptor_limit_is_chunked():
    """B * T * widest row * 4 bytes >= 2^31 (B > 13 107 at T = 40) is beyond what one tip_forward call addresses
    (TIP_ERR_UNSUPPORTED_CONFIG; tip_max_batch reports the limit); the reference accepts any batch, so the host runs it in chunks
    of chunk_batch() windows (the limit rounded down to whole rounds of 256) — bit-identical to running the pieces by hand."""
    m, _ = _gpu_model(0)
    m.set_plan("auto")
    B = 13107 + 150
    x_imu, x_s = synth.make_inputs(synth.PAPER, 300, 40, seed=77)
    reps = (B + 299) // 300
    xi = torch.tensor(np.tile(x_imu, (reps, 1, 1))[:B]).cuda()
    xs = torch.tensor(np.tile(x_s, (reps, 1, 1))[:B]).cuda()
    n0 = m.hip_forward_count()
    with torch.no_grad():
        yl = m.forward_last(xi, xs)
        assert m.hip_forward_count() == n0 + 2, "expected two chunks"
        cb = m.chunk_batch(40)
        assert m._ensure_handle().max_batch(40) == 13107 and cb == 13056
        a, b = m.forward_last(xi[:cb], xs[:cb]), m.forward_last(xi[cb:], xs[cb:])
    torch.cuda.synchronize()
    assert yl.shape == (B, 131) and bool(torch.isfinite(yl).all())
    assert torch.equal(yl, torch.cat([a, b]))
    assert torch.equal(yl[:300], yl[300:600])          # the same windows, 300 streams further on
    m.release_buffers()


@pytest.mark.parametrize("B", [256 + 1, 256 + 16, 256 + 44, 256 + 64, 256 + 128, 512 + 7, 512 + 44, 768 + 64])
def test_auto_splits_whole_rounds_and_a_small_remainder(B):
    """VERDICT r03 weak #6 (batch quantisation): AUTO runs a batch of whole rounds of #CUs windows plus a small remainder as two
    launch sequences — the rounds on the one-/two-window encoder, the remainder on the few-stream latency plan (up to 32 windows) or the window-split encoder — when its cost model
    says that beats one more full round (round 5: a window-split remainder shares the whole rounds' recurrence and output projection
    where that does not cost the recurrence a tile step: 256 + 44 ... 256 + 128 and 768 + 64 here, not 512 + 44).  One forward for the caller; every window bit-identical to what its part gives when
    it is run on its own; both output forms; keep mask carried to both parts."""
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    if ncu != 256:
        pytest.skip("the split is sized for a full 256-CU part")
    m, w = _gpu_model(0)
    x_imu, x_s = synth.make_inputs(synth.PAPER, B, 40, seed=300 + B)
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    bm = B - B % 256
    with torch.no_grad():
        m.set_plan("auto", profile=1)
        n0 = m.hip_forward_count()
        y = m(xi, xs)
        torch.cuda.synchronize()
        assert m.hip_forward_count() == n0 + 1
        stages = {n: k for n, _, k in m.profile_read()}
        if B % 256 <= 32:
            assert stages.get("fused_encoder") == 1 and stages.get("latency_chain") == 1, stages   # both parts ran
        else:
            assert stages.get("fused_encoder") == 2 and "latency_chain" not in stages, stages
        m.set_plan("auto", profile=0)
        a, b = m(xi[:bm], xs[:bm]), m(xi[bm:], xs[bm:])
        assert torch.equal(y, torch.cat([a, b]))
        yl = m.forward_last(xi, xs)
        assert torch.equal(yl, y[:, -1])
        mask = (torch.rand_like(xs) > 0.3).float()
        h = m._ensure_handle()
        ws = torch.empty(h.workspace_bytes(B, 40), dtype=torch.uint8, device="cuda")
        outs = []
        for lo, hi in ((0, B), (0, bm), (bm, B)):
            o = torch.empty(hi - lo, 40, 131, device="cuda")
            h.forward(xi[lo:hi].contiguous().data_ptr(), xs[lo:hi].contiguous().data_ptr(), o.data_ptr(), hi - lo, 40, tlib.TIP_FWD_KEEP_MASK,
                      mask[lo:hi].contiguous().data_ptr(), 1.25, ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
            outs.append(o)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], torch.cat(outs[1:]))
    sel = np.array([0, bm - 1, bm, B - 1])
    yo = oracle.forward(synth.PAPER, w, x_imu[sel], x_s[sel], dtype=np.float64)
    assert np.abs(y[torch.tensor(sel).cuda()].cpu().numpy() - yo).max() < TOL_TIGHT
    m.check_handoffs()


def test_auto_over_a_batch_sweep_matches_the_one_window_plan():
    """Every AUTO decision boundary in one sweep (latency plan / one window on four CUs / on two / one or two windows per CU / whole
    rounds + remainder on each of the few-stream plans, below and above the 1024-window mark where the workspace layout used to
    change): finite, hand-offs clean, within summation-order distance of the explicit one-window plan, and — the caller's workspace
    being sized by tip_workspace_bytes of the WHOLE batch — no part ever short of workspace (round 4: B = 1064 was)."""
    m, _ = _gpu_model(0)
    x_imu, x_s = synth.make_inputs(synth.PAPER, 256, 40, seed=4242)
    sizes = sorted(set([1, 2, 8, 9, 31, 32, 33, 47, 48, 49, 63, 64, 65, 100, 127, 128, 129, 200, 255, 256, 257, 258, 287, 288, 289, 300, 319, 320, 321,
                        383, 384, 385, 400, 511, 512, 513, 545, 600, 767, 768, 769, 801, 1000, 1023, 1024, 1025, 1056, 1057, 1064, 1088, 1089,
                        1152, 1153, 1279, 1280, 1281, 1500, 2047, 2048, 2049, 2081, 2112, 2113, 2200]))
    with torch.no_grad():
        for B in sizes:
            reps = (B + 255) // 256
            xi = torch.tensor(np.tile(x_imu, (reps, 1, 1))[:B]).cuda()
            xs = torch.tensor(np.tile(x_s, (reps, 1, 1))[:B]).cuda()
            m.set_plan("auto")
            n0 = m.hip_forward_count()
            y = m(xi, xs)
            torch.cuda.synchronize()
            assert m.hip_forward_count() == n0 + 1, B
            assert bool(torch.isfinite(y).all()), B
            m.set_plan("fusedh")
            ref = m(xi, xs)
            assert float((y - ref).abs().max()) < 5e-6, B
            yl = None
            m.set_plan("auto")
            yl = m.forward_last(xi, xs)
            assert torch.equal(yl, y[:, -1]), B
    m.check_handoffs()


def test_few_stream_plan_over_changing_layouts():
    """The one-launch form of the few-stream plan (B <= 24: stages, recurrence and output projection as roles of ONE launch, flags and
    launch counters at the front of the workspace) over a schedule whose batch size and window length change from call to call — every
    change moves the activations inside the workspace; the flag area must not move with them (round 6: it did at first, and a window
    found its own earlier stamps current: finite-but-wrong rows, tools/flow_soak.py).  Bit-identical results pass after pass, against
    the launch chain's numbers at B > 24 for the same windows (same kernels' bodies, same summation orders), no wait gives up."""
    cfg = synth.PAPER
    m = make_model(cfg)
    load_synth(m, cfg, 0)
    m = m.cuda().eval()
    h = m._ensure_handle()
    h.set_option(tlib.TIP_OPT_AUTO_DEMOTE, 0)
    assert h.workspace_bytes(1, 40) > 34 * 64 * 64 * 8      # the flag area is part of every paper-configuration workspace
    rng = np.random.RandomState(11)
    sched = [(int(rng.choice([1, 2, 3, 5, 8])), int(rng.randint(1, 41)), bool(rng.randint(2))) for _ in range(60)]
    sched += [(1, t, True) for t in range(1, 41)] + [(8, 39, True), (2, 40, True), (8, 39, True)]
    data = {}
    for B, T, _ in sched:
        if (B, T) not in data:
            x_imu, x_s = synth.make_inputs(cfg, B, T, seed=100 * B + T)
            data[(B, T)] = (torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda())
    t0 = tlib.spin_timeouts()
    ref = []
    with torch.no_grad():
        for p in range(6):
            for i, (B, T, last) in enumerate(sched):
                xi, xs = data[(B, T)]
                y = (m.forward_last(xi, xs) if last else m(xi, xs)).clone()
                if p == 0:
                    assert bool(torch.isfinite(y).all())
                    ref.append(y)
                else:
                    assert torch.equal(y, ref[i]), (p, i, B, T, last)
        # the same windows through the launch chain (B = 9 <= 24 takes the one-launch form too: batch neighbours do not matter): same bits
        xi8, xs8 = data[(8, 39)]
        xi9, xs9 = torch.cat([xi8, xi8[:1]]), torch.cat([xs8, xs8[:1]])
        y9 = m.forward_last(xi9, xs9)
        y8 = m.forward_last(xi8, xs8)
        assert torch.equal(y9[:8], y8) and torch.equal(y9[8], y8[0])
        # two and three windows per XCD (B = 16, 24: the one-launch form's limit) against the launch chain at B = 25, full output
        x25i, x25s = synth.make_inputs(cfg, 25, 40, seed=77)
        x25i, x25s = torch.tensor(x25i).cuda(), torch.tensor(x25s).cuda()
        y25 = m(x25i, x25s)
        for Bf in (9, 16, 24):
            yf = m(x25i[:Bf].contiguous(), x25s[:Bf].contiguous())
            assert torch.equal(yf, y25[:Bf]), Bf
            assert torch.equal(m.forward_last(x25i[:Bf].contiguous(), x25s[:Bf].contiguous()), y25[:Bf, -1]), Bf
    torch.cuda.synchronize()
    m.check_handoffs()
    assert tlib.spin_timeouts() == t0

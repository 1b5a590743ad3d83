"""Host-side logic that needs no GPU: module surface, state-dict contract, packed-image folds, C-ABI exports,
torch-op composite vs the golden vectors."""
import ctypes
import io
import os
import re

import numpy as np
import pytest
import torch

import tip_amd
from tip_amd import synth
from tip_amd import lib as tlib
from conftest import ROOT, cfg_for_tag, seed_for_tag


def make_model(cfg, p_state=0.0, dropout=0.0):
    return tip_amd.TF_RNN_Past_State(
        cfg["input_size_imu"], cfg["size_s"], rnn_hid_size=cfg["rnn_hid_size"], tf_hid_size=cfg["tf_hid_size"],
        tf_in_dim=cfg["tf_in_dim"], n_heads=cfg["n_heads"], tf_layers=cfg["tf_layers"], dropout=dropout,
        in_dropout=0.0, past_state_dropout=p_state, with_rnn=cfg.get("with_rnn", True),
        with_acc_sum=cfg.get("with_acc_sum", False))


def load_synth(model, cfg, seed):
    w = synth.make_weights(cfg, seed=seed)
    model.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    return w


@pytest.mark.parametrize("cfg", [synth.PAPER, synth.TINY, dict(synth.TINY, with_rnn=False),
                                 dict(synth.TINY, with_acc_sum=False)])
def test_state_dict_contract(cfg):
    m = make_model(cfg)
    sd = m.state_dict()
    lay = synth.state_dict_layout(cfg)
    assert list(sd.keys()) == list(lay.keys())
    for k, shape in lay.items():
        assert tuple(sd[k].shape) == tuple(shape), k
    if cfg is synth.PAPER:
        assert len(sd) == 56 and sum(v.numel() for v in sd.values()) == 3677315


def test_torch_save_round_trip():
    m = make_model(synth.TINY)
    load_synth(m, synth.TINY, 3)
    buf = io.BytesIO()
    torch.save(m.state_dict(), buf)   # train_model.py:220-225
    buf.seek(0)
    m2 = make_model(synth.TINY)
    m2.load_state_dict(torch.load(buf))  # offline_testing_simple.py:96
    for (k1, v1), (k2, v2) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


def test_autograd_composite_matches_golden(golden):
    """train_model.py:175,192 — in grad mode the module runs torch ops; must equal the reference forward."""
    for tag, case in golden.items():
        if "mask" in tag:
            continue
        cfg = cfg_for_tag(tag)
        m = make_model(cfg)
        load_synth(m, cfg, seed_for_tag(tag))
        m.eval()
        with pytest.warns(UserWarning):
            y = m(torch.tensor(case["x_imu"]), torch.tensor(case["x_s"]))
        assert y.requires_grad
        err = np.abs(y.detach().numpy() - case["y32"]).max()
        assert err < 1e-5, (tag, err)
    y.sum().backward()
    assert m.in_linear.weight.grad is not None and torch.isfinite(m.in_linear.weight.grad).all()


def test_inference_on_cpu_fails_loudly():
    m = make_model(synth.TINY)
    x_imu, x_s = synth.make_inputs(synth.TINY, 1, 4)
    with torch.no_grad(), pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.tensor(x_imu), torch.tensor(x_s))


def test_c_abi_exports_every_declared_symbol():
    """The dynamic symbol table of libtip_hip.so IS the two headers: include/tip_hip.h (the drop-in boundary) and
    include/tip_hip_debug.h (measurement hooks) — nothing else is visible (-fvisibility=hidden + csrc/tip_exports.map)."""
    import subprocess
    hdr = open(os.path.join(ROOT, "include", "tip_hip.h")).read()
    declared = set(re.findall(r"\b(tip_[a-z0-9_]+)\s*\(", hdr)) - {"tip_stream_t"}
    assert declared == set(tlib.EXPORTS), declared ^ set(tlib.EXPORTS)
    dbg = set(re.findall(r"\b(tip_debug_[a-z0-9_]+)\s*\(", open(os.path.join(ROOT, "include", "tip_hip_debug.h")).read()))
    assert len(dbg) == 12
    lib = ctypes.CDLL(tlib.LIB_PATH)
    for name in declared | dbg:
        assert hasattr(lib, name), name
    out = subprocess.run(["nm", "-D", "--defined-only", tlib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported == declared | dbg, exported ^ (declared | dbg)
    assert tlib.load().tip_abi_version() == tlib.TIP_ABI_VERSION == 5
    assert int(re.search(r"#define TIP_ABI_VERSION (\d+)", hdr).group(1)) == 5


def test_max_batch_and_pack_options_without_a_gpu():
    """tip_max_batch (the host chunks by the library's own limit), the retired plan / option values, TIP_OPT_AUTO_DEMOTE /
    TIP_OPT_DEMOTED, TIP_OPT_F1S_PARTS."""
    m = make_model(synth.PAPER)
    h = m._ensure_handle()
    assert h.max_batch(40) == (2 ** 31 - 1) // (4 * 1024 * 40) == 13107          # widest row: the 1024-wide FFN hidden
    assert h.max_batch(40, fp64=True) == min((2 ** 31 - 1) // (1024 * 40), 65535 * 64 // 40)
    assert h.max_batch(4096) == (2 ** 31 - 1) // (4 * 1024 * 4096)
    with pytest.raises(tlib.TipStatusError):
        h.max_batch(0)
    hs = make_model(synth.SCALED)._ensure_handle()
    assert hs.max_batch(80) == (2 ** 31 - 1) // (4 * 4096 * 80) == 1638            # BASELINE config 5 (B = 4096) runs as >= 3 chunks
    base = h.packed_bytes()
    # round 6: the split-fp16 sections / plans are retired in every build — no such option, no such plan, image unchanged
    for v in (0, 1, 2):
        with pytest.raises(tlib.TipStatusError):
            h.set_option(6, v)
    for pl in (5, 7, 8):
        with pytest.raises(tlib.TipStatusError) as ei:
            h.set_option(tlib.TIP_OPT_PLAN, pl)
        assert ei.value.status == tlib.TIP_ERR_UNSUPPORTED_CONFIG
    assert h.packed_bytes() == base
    with pytest.raises(RuntimeError, match="retired"):
        m.set_plan("fused16")
    assert h.get_option(tlib.TIP_OPT_AUTO_DEMOTE) == 1 and h.get_option(tlib.TIP_OPT_DEMOTED) == 0
    h.set_option(tlib.TIP_OPT_DEMOTED, 1)
    assert m.is_demoted()
    m.undemote()
    assert not m.is_demoted()
    # TIP_OPT_F1S_PARTS: workgroups per window of the window-split plan (0 = the library's choice); the plan names pin it
    assert h.get_option(tlib.TIP_OPT_F1S_PARTS) == 0
    m.set_plan("fused1s4")
    assert h.get_option(tlib.TIP_OPT_F1S_PARTS) == 4 and h.get_option(tlib.TIP_OPT_PLAN) == tlib.TIP_PLAN_FUSED1S
    m.set_plan("fused1s2")
    assert h.get_option(tlib.TIP_OPT_F1S_PARTS) == 2
    m.set_plan("auto")
    assert h.get_option(tlib.TIP_OPT_F1S_PARTS) == 0
    with pytest.raises(tlib.TipStatusError):
        h.set_option(tlib.TIP_OPT_F1S_PARTS, 3)


def test_handle_table_status_and_errors():
    cfg = synth.PAPER
    m = make_model(cfg)
    h = m._ensure_handle()
    table = h.tensor_table()
    assert [n for n, _ in table] == list(synth.state_dict_layout(cfg).keys())
    assert [tuple(s) for _, s in table] == [tuple(s) for s in synth.state_dict_layout(cfg).values()]
    assert h.packed_bytes() % 256 == 0 and h.packed_bytes() >= 3677315 * 4
    assert h.workspace_bytes(256, 40) > 256 * 40 * 256 * 4
    lib = tlib.load()
    assert lib.tip_strerror(0) == b"ok" and b"workspace" in lib.tip_strerror(-4)
    bad = tlib.TipConfig(72, 131, 512, 1024, 250, 16, 4, 1, 1, 40)   # 250 % 16 != 0
    with pytest.raises(tlib.TipStatusError):
        tlib.Handle(bad)
    # forward before weights are attached -> TIP_ERR_NOT_READY (no kernel launched, safe without a GPU)
    h2 = tlib.Handle(m._tip_config())
    with pytest.raises(tlib.TipStatusError) as ei:
        h2.forward(1, 1, 1, 1, 1, 0, None, 1.0, 0, 0, 0)
    assert ei.value.status == -3
    with pytest.raises(tlib.TipStatusError):
        h2.forward(0, 1, 1, 1, 1, 0, None, 1.0, 0, 0, 0)


def test_new_entry_points_check_their_arguments_without_a_gpu():
    """Round-3 entry points, the paths that return before any kernel is launched (safe on the CPU box): tip_forward_f64 /
    tip_forward_f64_bytes, TIP_OPT_FUSE_HEAD, TIP_STREAM_FRAME_AUTO."""
    import ctypes
    cfg = synth.PAPER
    m = make_model(cfg)
    h = m._ensure_handle()
    lib = tlib.load()
    # fp64 forward: workspace size grows with B * T and covers every activation of the layer-by-layer path in doubles
    n1, n2 = h.forward_f64_bytes(1, 40), h.forward_f64_bytes(256, 40)
    per_row = 8 * (221 + 256 + 768 + 256 + 1024 + 512 + 512)
    assert n2 > n1 >= 40 * per_row and n2 >= 256 * 40 * per_row
    n_t = len(h.tensor_table())
    ptrs = [8] * n_t
    h.forward_f64(ptrs, 8, 8, 8, 0, 40, 0, None, 1.0, 0, 0, 0)            # B = 0: nothing to do, TIP_OK
    with pytest.raises(tlib.TipStatusError) as ei:                          # wrong tensor count
        h.forward_f64(ptrs[:-1], 8, 8, 8, 1, 40, 0, None, 1.0, 256, 1 << 30, 0)
    assert ei.value.status == -1
    with pytest.raises(tlib.TipStatusError) as ei:                          # workspace too small
        h.forward_f64(ptrs, 8, 8, 8, 1, 40, 0, None, 1.0, 256, 1024, 0)
    assert ei.value.status == -4
    with pytest.raises(tlib.TipStatusError) as ei:                          # misaligned workspace
        h.forward_f64(ptrs, 8, 8, 8, 1, 40, 0, None, 1.0, 264, 1 << 30, 0)
    assert ei.value.status == -4
    # options
    assert h.get_option(tlib.TIP_OPT_FUSE_HEAD) == 0   # reserved since round 5: accepted, without effect
    h.set_option(tlib.TIP_OPT_FUSE_HEAD, 1)
    assert h.get_option(tlib.TIP_OPT_FUSE_HEAD) == 1
    with pytest.raises(tlib.TipStatusError):
        h.set_option(tlib.TIP_OPT_FUSE_HEAD, 2)
    h.set_option(tlib.TIP_OPT_FUSE_HEAD, 0)
    # streaming: AUTO is -1, anything below is refused; zero streams is a no-op
    assert tlib.TIP_STREAM_FRAME_AUTO == -1
    assert lib.tip_stream_ingest(ctypes.c_void_p(8), ctypes.c_void_p(8), 0, -1, ctypes.c_void_p(8), ctypes.c_void_p(8), None) == 0
    assert lib.tip_stream_ingest(ctypes.c_void_p(8), ctypes.c_void_p(8), 1, -2, ctypes.c_void_p(8), ctypes.c_void_p(8), None) == -1
    assert lib.tip_stream_consume(ctypes.c_void_p(8), ctypes.c_void_p(8), 0, -1, ctypes.c_void_p(8), ctypes.c_void_p(8), None) == 0
    assert lib.tip_stream_consume(ctypes.c_void_p(8), ctypes.c_void_p(8), 1, -2, ctypes.c_void_p(8), ctypes.c_void_p(8), None) == -1
    assert [lib.tip_stream_window_len(f) for f in (0, 4, 5, 43, 44, 45, 1000)] == [0, 0, 1, 39, 40, 40, 40]


def _unpack_linear(img, off_w, off_b, N, K, Npad, Kpad):
    W = img[off_w: off_w + Npad * Kpad].reshape(Npad, Kpad)
    return W[:N, :K], img[off_b: off_b + N], W


def test_packed_image_folds():
    """The packed image must encode: shuffle :88-89 in in_linear rows, root-vel zero :75 in its columns,
    0.25 in W_q, b_ih+b_hh, and W_hh in 16x16x4 B-fragment order."""
    cfg = synth.PAPER
    m = make_model(cfg)
    w = load_synth(m, cfg, 0)
    img = m.pack_host().numpy().view(np.float32)
    D, H, In = 256, 16, 221
    dh = D // H
    KB = 224
    Win = img[0: 256 * KB].reshape(256, KB)
    ref = w["in_linear.weight"].copy()
    ref[:, 90 + 108: 90 + 111] = 0.0
    perm = np.array([(n % H) * dh + n // H for n in range(D)])   # packed row a*H+b <- reference row b*dh+a
    assert np.array_equal(Win[:, :In], ref[perm])
    assert np.all(Win[:, In:] == 0)
    # semantic check of the fold against the oracle's in-linear tap
    from oracle import oracle
    x_imu, x_s = synth.make_inputs(cfg, 1, 3)
    _, taps = oracle.forward(cfg, w, x_imu, x_s, dtype=np.float64, taps=True)
    u = np.concatenate([x_imu, np.nan_to_num(x_s, nan=0.0)], axis=2)[0].astype(np.float64)
    b_in = img[256 * KB: 256 * KB + 256]
    z = u @ Win[:, :In].astype(np.float64).T + b_in
    assert np.abs(z - taps["in"][0]).max() < 1e-6
    # W_hh fragments
    R = 512
    flat = img
    whh = w["rnn.weight_hh_l0"]
    # locate the fragment section by searching for its first element pattern: nb=0,kb=0,lane=0 -> W[0][0..3]
    frag0 = np.array([whh[l & 15, 4 * (l >> 4) + s] for l in range(64) for s in range(4)], dtype=np.float32)
    hits = [i for i in range(0, flat.size - 256, 64) if flat[i] == frag0[0] and np.array_equal(flat[i:i + 256], frag0)]
    assert len(hits) >= 1
    off = hits[0]
    nb, kb = 5, 17
    blk = flat[off + (nb * 32 + kb) * 256: off + (nb * 32 + kb + 1) * 256].reshape(64, 4)
    exp = np.array([[whh[nb * 16 + (l & 15), kb * 16 + 4 * (l >> 4) + s] for s in range(4)] for l in range(64)])
    assert np.array_equal(blk, exp)


def test_zero_edit_drop_in_import_path():
    """`from simple_transformer_with_state import TF_RNN_Past_State` (train_model.py:14) must resolve to our module
    when the drop-in directory is put first on PYTHONPATH (the full recipe: tests/test_dropin_cpu.py)."""
    import subprocess
    import sys
    code = ("from simple_transformer_with_state import TF_RNN_Past_State as M; import inspect;"
            "m = M(72, 131, rnn_hid_size=64, tf_hid_size=32, tf_in_dim=32, n_heads=4, tf_layers=1, dropout=0.0,"
            "in_dropout=0.0, past_state_dropout=0.8, with_acc_sum=True);"
            "print(len(m.state_dict()), 'tip_amd' in inspect.getsourcefile(M) or 'inertial-poser_amd' in inspect.getsourcefile(M))")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "transformer-inertial-poser_amd", "dropin"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd="/tmp")
    assert out.returncode == 0, out.stderr
    assert out.stdout.strip().splitlines()[-1] == "20 True"


def test_stock_nn_composite_matches_golden(golden):
    """oracle/torch_stock.py (stock nn.Linear / nn.TransformerEncoder / nn.RNN, the CPU dispatch the reference takes and the
    bench's cpu_baseline) against the vectors captured from the reference: same modules, same kernels -> same bits."""
    from oracle import torch_stock
    for tag, case in golden.items():
        cfg = cfg_for_tag(tag)
        m = torch_stock.build(cfg, synth.make_weights(cfg, seed=seed_for_tag(tag)))
        km = torch.tensor(case["mask"]) if "mask" in tag else None
        ks = 1.0 / (1.0 - float(case["p"][0])) if "mask" in tag else 1.0
        with torch.no_grad():
            y = m(torch.tensor(case["x_imu"]), torch.tensor(case["x_s"]), km, ks).numpy()
        if km is None:
            assert np.array_equal(y, case["y32"]), (tag, np.abs(y - case["y32"]).max())
        else:
            assert np.abs(y - case["y32"]).max() < 2e-6, tag     # x * mask * 1.25 vs torch's fused dropout scaling


@pytest.mark.parametrize("cfg", [synth.PAPER, dict(synth.TINY, with_rnn=False)])
def test_reset_parameters_matches_torch_initialisers(cfg):
    """Training from scratch under train_model.py depends on the initial distributions: every tensor of the drop-in must be
    drawn like the stock module's (nn.Linear kaiming-uniform(a=sqrt 5) = U(+-1/sqrt(fan_in)) for weight and bias,
    nn.MultiheadAttention xavier-uniform in_proj + zero biases, nn.LayerNorm ones/zeros, nn.RNN U(+-1/sqrt(hidden)))."""
    from oracle import torch_stock
    torch.manual_seed(7)
    ours = make_model(cfg)
    theirs = torch_stock.StockTIP(cfg)
    sd_o, sd_t = ours.state_dict(), theirs.state_dict()
    assert list(sd_o.keys()) == list(sd_t.keys())
    for k in sd_o:
        a, b = sd_o[k].double(), sd_t[k].double()
        assert a.shape == b.shape, k
        if float(b.abs().max()) == 0.0 or float((b - 1.0).abs().max()) == 0.0:
            assert torch.equal(a, b), k                      # constants: zeros (attention biases, LN beta) / ones (LN gamma)
            continue
        # uniform on [-bound, bound]: torch's own draw estimates the bound; ours must fill the same interval
        n = a.numel()
        bound = float(b.abs().max()) * (n + 1) / n           # unbiased max estimate of a uniform
        assert float(a.abs().max()) <= bound * (1.0 + 8.0 / n) + 1e-12, k
        assert float(a.abs().max()) >= bound * (1.0 - 8.0 / n) - 1e-12 or n < 64, k
        tol = 4.0 / np.sqrt(n)                               # ~4 sigma of the sample statistics of a uniform
        assert abs(float(a.mean())) <= bound * tol, k
        assert abs(float(a.std()) / (bound / np.sqrt(3.0)) - 1.0) <= 1.5 * tol + 0.01, k


def test_f64_training_and_loss_entry_points_check_their_arguments_without_a_gpu():
    """Round-4 entry points, the paths that return before any kernel is launched: tip_train_*_f64, tip_loss_*_f64."""
    import ctypes
    m = make_model(synth.PAPER)
    h = m._ensure_handle()
    lib = tlib.load()
    s1, x1 = h.train_bytes(4, 40, fp64=True)
    s2, x2 = h.train_bytes(8, 40, fp64=True)
    per_row = 8 * (221 + 256 + 4 * (768 + 256 * 5 + 1024) + 512 * 2)        # U, X0, per layer qkv + att/z1/x1/z2/xo + hid, IH, HALL
    assert s2 > s1 >= 4 * 40 * per_row and x2 > x1 > 0
    with pytest.raises(tlib.TipStatusError) as ei:
        h.train_bytes(4, 200, fp64=True)                                     # attention backward keeps two T x T tiles in LDS
    assert ei.value.status == -2
    assert make_model(dict(synth.TINY, with_rnn=False))._ensure_handle().train_bytes(2, 9, fp64=True)[0] > 0    # any configuration
    n_t = len(h.tensor_table())
    ptrs = [8] * n_t
    with pytest.raises(tlib.TipStatusError) as ei:                          # wrong tensor count
        h.train_forward(ptrs[:-1], 8, 8, None, 1.0, 0.1, 1, 8, 256, 1 << 40, 1, 40, 0, fp64=True)
    assert ei.value.status == -1
    with pytest.raises(tlib.TipStatusError) as ei:                          # stash too small
        h.train_forward(ptrs, 8, 8, None, 1.0, 0.1, 1, 8, 256, 1024, 1, 40, 0, fp64=True)
    assert ei.value.status == -4
    with pytest.raises(tlib.TipStatusError) as ei:                          # p_drop out of range
        h.train_forward(ptrs, 8, 8, None, 1.0, 1.0, 1, 8, 256, 1 << 40, 1, 40, 0, fp64=True)
    assert ei.value.status == -1
    with pytest.raises(tlib.TipStatusError) as ei:                          # gradient buffer too small
        h.train_backward(ptrs, 8, 256, 1 << 40, 256, 1 << 40, 8, 10, 0.1, 1, 1, 40, 0, fp64=True)
    assert ei.value.status == -1
    vp = ctypes.c_void_p
    assert lib.tip_loss_forward_f64(vp(8), 131, vp(8), 131, 2, 5, 108, 2, 5, 7, vp(8), vp(8), 512, None) == -1     # n_vel must be 0 or 3
    assert lib.tip_loss_forward_f64(vp(8), 131, vp(8), 131, 2, 5, 108, 3, 5, 7, vp(8), vp(8), 8, None) == -4       # workspace too small
    assert lib.tip_loss_backward_f64(vp(8), 131, vp(8), 131, 2, 5, 108, 3, 5, 7, vp(8), None, vp(8), 100, None) == -1   # ld_dpred < W


def test_workspace_of_a_part_never_exceeds_the_whole():
    """AUTO runs a batch of whole rounds of 256 windows + a remainder as two forwards over the SAME workspace: every part's
    tip_workspace_bytes must fit the whole's (round 4: the exchange section of the window-split plans was sized for B <= 1024 and
    absent above, so the 1024-window part of a 1064-window batch asked for more than the whole — TIP_ERR_WORKSPACE)."""
    m = make_model(synth.PAPER)
    h = m._ensure_handle()
    for T in (40, 17):
        sizes = {B: h.workspace_bytes(B, T) for B in list(range(1, 300)) + list(range(300, 2400, 7)) + [1024, 1025, 1064, 1279, 1280, 2048, 4096, 8192]}
        for B, total in sizes.items():
            r = B % 256
            if B > 256 and r:
                assert h.workspace_bytes(B - r, T) <= total and h.workspace_bytes(r, T) <= total, (B, T)
        big = [sizes[B] for B in sorted(sizes) if B >= 256]
        assert all(a <= b for a, b in zip(big, big[1:])), "monotone from one round up"


def test_default_library_does_not_read_the_environment():
    """kernel selection of a production handle must not depend on the process environment.  The launchers'
    TIP_* measurement switches go through tip_env() (csrc/tip_internal.h), which is a constant nullptr unless the library is built
    with -DTIP_MEASURE (`make measure` -> libtip_hip_measure.so): the default library does not even import getenv."""
    import glob
    import subprocess
    und = subprocess.run(["nm", "-D", "--undefined-only", tlib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert os.path.basename(tlib.LIB_PATH) == "libtip_hip.so"
    assert "getenv" not in und
    hits = []
    for f in glob.glob(os.path.join(ROOT, "transformer-inertial-poser_amd", "csrc", "*.hip")) + \
            glob.glob(os.path.join(ROOT, "transformer-inertial-poser_amd", "csrc", "*.h")):
        n = len(re.findall(r"\bgetenv\s*\(", open(f).read()))
        if n:
            hits.append((os.path.basename(f), n))
    assert hits == [("tip_internal.h", 1)], hits


def test_parameter_list_cache_follows_swapped_parameters():
    """the module keeps list(self.parameters()) between calls (the tree walk costs ~40 us per forward) and must notice
    every way a Parameter or a submodule can be swapped WITHOUT going through its own _apply: torch.func.functional_call (writes
    `_parameters[name]` directly), load_state_dict(assign=True), re-assigning a leaf's weight, replacing a submodule, a submodule's own
    .double().  The cached list is validated by identity on every call; the dtype signature follows the storage pointers."""
    import copy
    cfg = synth.TINY
    m = make_model(cfg)
    load_synth(m, cfg, 0)
    ids = lambda ps: [id(p) for p in ps]   # noqa: E731
    pl = m._plist()
    assert ids(pl) == ids(m.parameters()) and m._plist() is pl                       # cached, in state-dict order
    assert [n for n, _ in m.named_parameters()] == list(m.state_dict().keys())
    # (1) re-assigning a leaf's parameter
    m.rnn.weight_hh_l0 = torch.nn.Parameter(torch.zeros_like(m.rnn.weight_hh_l0))
    assert ids(m._plist()) == ids(m.parameters())
    # (2) load_state_dict(assign=True): every Parameter object is replaced
    before = ids(m._plist())
    m.load_state_dict({k: v.clone() for k, v in m.state_dict().items()}, assign=True)
    assert ids(m._plist()) == ids(m.parameters()) and ids(m._plist()) != before
    # (3) replacing a submodule
    m.tf_encode.layers[1] = copy.deepcopy(m.tf_encode.layers[0])
    assert ids(m._plist()) == ids(m.parameters())
    # (4) torch.func.functional_call: inside the call the module must see the substituted tensors (gradients reach THEM), and the
    #     originals again afterwards
    x_imu, x_s = synth.make_inputs(cfg, 2, 5, seed=3)
    xi, xs = torch.tensor(x_imu), torch.nan_to_num(torch.tensor(x_s))
    m.eval()
    subst = {k: (v.detach().clone() * 0.5).requires_grad_(True) for k, v in m.named_parameters()}
    y_sub = torch.func.functional_call(m, subst, (xi, xs))
    y_own = m(xi, xs)
    assert not torch.allclose(y_sub, y_own)
    y_sub.sum().backward()
    assert all(v.grad is not None for v in subst.values()) and all(p.grad is None for p in m.parameters())
    assert ids(m._plist()) == ids(m.parameters())
    # (5) a submodule's own .double(): same Parameter objects, new storage and dtype -> the HIP training path must say no (mixed dtypes)
    m.rnn.double()
    assert ids(m._plist()) == ids(m.parameters())
    assert m._hip_train_ok(xi, xs) is False
    # (6) deleting a parameter / a submodule does not crash the validation
    m2 = make_model(cfg)
    m2._plist()
    del m2.tf_encode.layers[1]
    assert ids(m2._plist()) == ids(m2.parameters())

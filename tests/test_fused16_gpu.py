"""The exploratory split-fp16 plan ("fused16", csrc/tip_s16.hip: fp32 GEMM operands emulated as hi + lo fp16 on the f16 matrix cores,
fp32 accumulation) against the SAME bar as every other plan: golden vectors and oracle shapes ride in tests/test_hip_parity.py
(ALL_PLANS), the conditioning sweep in tests/test_conditioning_gpu.py; here the properties — batch independence, last row, T = 1..40
causality, keep-mask, determinism — and the error relative to the fp32-MFMA plan."""
import numpy as np
import pytest
import torch

from tip_amd import synth
from oracle import oracle
from test_host_cpu import make_model, load_synth

from tip_amd import lib as _tlib

# the exploratory plans are compiled into the measurement build only (csrc: make measure); tests/test_exploratory_build.py re-runs
# this file against it (TIP_LIB=measure)
pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not _tlib.MEASURE, reason="exploratory plans: measurement build only (TIP_LIB=measure)")]
TOL_TIGHT = 2e-5


def _gpu_model(seed):
    m = make_model(synth.PAPER)
    w = load_synth(m, synth.PAPER, seed)
    return m.cuda().eval(), w


def _fwd(m, xi, xs, last=False):
    n0 = m.hip_forward_count()
    with torch.no_grad():
        y = (m.forward_last if last else m)(torch.as_tensor(xi).cuda(), torch.as_tensor(xs).cuda())
    torch.cuda.synchronize()
    assert m.hip_forward_count() == n0 + 1
    return y.cpu().numpy()


@pytest.mark.parametrize("B,T", [(1, 40), (3, 17), (65, 40), (256, 40), (300, 33)])
def test_fused16_vs_oracle_and_properties(B, T):
    cfg = synth.PAPER
    m, w = _gpu_model(2)
    m.set_plan("fused16")
    x_imu, x_s = synth.make_inputs(cfg, B, T, seed=900 + B + T, nan_frac=0.02)
    y = _fwd(m, x_imu, x_s)
    assert np.isfinite(y).all()
    assert np.array_equal(y, _fwd(m, x_imu, x_s)), "run-to-run difference"
    n = min(B, 6)
    yo = oracle.forward(cfg, w, x_imu[:n], x_s[:n], dtype=np.float64)
    e16 = np.abs(y[:n] - yo).max()
    assert e16 < TOL_TIGHT, e16
    m.set_plan("fusedh")
    e32 = np.abs(_fwd(m, x_imu[:n], x_s[:n]) - yo).max()
    m.set_plan("fused16")
    assert e16 < 3.0 * e32 + 1e-6, (e16, e32)          # not a less accurate implementation than the fp32-MFMA plan
    assert np.array_equal(_fwd(m, x_imu, x_s, last=True), y[:, -1])
    if B > 8:
        sel = np.array([0, 1, B // 2, B - 1])
        assert np.array_equal(_fwd(m, x_imu[sel], x_s[sel]), y[sel]), "a stream's result depends on its batch neighbours"


def test_fused16_causality_T_1_to_40():
    cfg = synth.PAPER
    m, w = _gpu_model(0)
    m.set_plan("fused16")
    x_imu, x_s = synth.make_inputs(cfg, 2, 40, seed=9)
    yfull = oracle.forward(cfg, w, x_imu, x_s, dtype=np.float64)
    for T in range(1, 41):
        yl = _fwd(m, x_imu[:, :T], x_s[:, :T], last=True)
        assert np.abs(yl - yfull[:, T - 1]).max() < TOL_TIGHT, T


def test_fused16_keep_mask_golden(golden):
    from tip_amd import lib as tlib
    case = golden["paper_mask_s0_B2_T40"]
    m, _ = _gpu_model(0)
    h = m._ensure_handle()
    h.set_option(tlib.TIP_OPT_PACK_SPLIT16, tlib.TIP_PACK_SPLIT16_FUSED)    # the image must carry the split-fp16 section (C-ABI spelling)
    m.refresh_packed(torch.device("cuda:0"))
    xi, xs = torch.tensor(case["x_imu"]).cuda(), torch.tensor(case["x_s"]).cuda()
    mask = torch.tensor(case["mask"]).cuda()
    p = float(case["p"][0])
    y = torch.zeros(2, 40, 131, device="cuda")
    ws = torch.empty(h.workspace_bytes(2, 40), dtype=torch.uint8, device="cuda")
    h.set_option(tlib.TIP_OPT_PLAN, tlib.TIP_PLAN_FUSED16)
    h.forward(xi.data_ptr(), xs.data_ptr(), y.data_ptr(), 2, 40, tlib.TIP_FWD_KEEP_MASK, mask.data_ptr(), 1.0 / (1.0 - p), ws.data_ptr(),
              ws.numel(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.abs(y.cpu().numpy() - case["y64"]).max() < TOL_TIGHT
    h.set_option(tlib.TIP_OPT_PLAN, tlib.TIP_PLAN_AUTO)


def _model_s16_general(cfg, seed):
    """A model whose packed image carries the split-fp16 copies of the general plan's big linears (set_plan("general16") sets
    TIP_OPT_PACK_SPLIT16 bit 1 on the handle; the tests switch plans afterwards without losing the copies)."""
    m = make_model(cfg)
    w = load_synth(m, cfg, seed)
    m = m.cuda().eval()
    m.set_plan("general16")
    m.set_plan("auto")
    return m, w


@pytest.mark.parametrize("B,T", [(5, 80), (9, 40)])
def test_general16_scaled_widths_vs_oracle(B, T):
    """Exploratory "general16": the general plan with its big linears (d = 1024, ffn = 4096: panel-GEMM shapes, M >= 320 rows) on
    split-fp16 operands — same tolerance as the fp32 general plan, error not worse than 3x its error; host and device packers
    produce the same image including the split copies."""
    cfg = dict(synth.SCALED, tf_layers=2)
    m, w = _model_s16_general(cfg, 0)
    x_imu, x_s = synth.make_inputs(cfg, B, T, seed=14 + B)
    yo = oracle.forward(cfg, w, x_imu, x_s, dtype=np.float64)
    m.set_plan("general")
    e32 = np.abs(_fwd(m, x_imu, x_s) - yo).max()
    m.set_plan("general16")
    y = _fwd(m, x_imu, x_s)
    e16 = np.abs(y - yo).max()
    assert np.isfinite(y).all() and e16 < TOL_TIGHT, e16
    assert e16 < 3.0 * e32 + 1e-6, (e16, e32)
    assert np.array_equal(y, _fwd(m, x_imu, x_s))
    assert np.array_equal(_fwd(m, x_imu, x_s, last=True), y[:, -1])
    # the image: host packer == device packer, split copies included
    host = m.pack_host()
    dev = m.pack_device(torch.device("cuda:0")).cpu()
    assert host.numel() == dev.numel() and torch.equal(host, dev)
    m0 = make_model(cfg)
    assert m0._ensure_handle().packed_bytes() < m._ensure_handle().packed_bytes(), "the split copies must be opt-in"

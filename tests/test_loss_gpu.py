"""GPU: the fused training losses (csrc/tip_loss.hip through tip_amd.learning_utils) against the REAL reference functions'
output (tip_loss_golden.npz) and the oracle (oracle/loss_oracle.py)."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_loss_golden import CASES, N_SBPS, make_case    # noqa: E402
import tip_amd                                            # noqa: E402
from oracle import loss_oracle                            # noqa: E402
from test_loss_oracle import GOLD, case_inputs, close     # noqa: E402

pytestmark = pytest.mark.gpu
REL = 2e-5    # vs the fp32 reference; vs the fp64 oracle the kernels (fp64 reduction) are held to 2e-6 below


def run_fused(pred, gt, **kw):
    lu = tip_amd.learning_utils
    yp = torch.tensor(pred).cuda().requires_grad_(True)
    total, parts = lu.train_loss(yp, torch.tensor(gt).cuda(), N_SBPS, return_parts=True, **kw)
    total.backward()              # a NaN total still back-propagates, as in the reference
    return float(total.detach()), parts.cpu().numpy().astype(np.float64), yp.grad.cpu().numpy()


def test_fused_loss_matches_reference_golden():
    assert torch.cuda.is_available()
    z = np.load(GOLD)
    for tag in CASES:
        pred, gt = case_inputs(z, tag)
        total, parts, grad = run_fused(pred, gt)
        ref = z[tag + "/losses"]
        assert close([total, *parts], ref, REL), (tag, total, parts, ref)
        g = z[tag + "/grad"]
        if np.isnan(ref[0]):
            # d(NaN total)/dy: the reference's graph still yields the finite per-term gradients; so do we
            assert np.isfinite(grad).all()
        assert np.abs(grad - g).max() <= REL * max(1.0, np.abs(g).max()), tag


def test_fused_loss_matches_oracle_full_batch():
    pred, gt = make_case("full", 256, 40, 11)
    total, parts, grad = run_fused(pred, gt)
    o_total, o_parts, o_grad = loss_oracle.train_loss(pred, gt, N_SBPS)
    assert abs(total - o_total) <= 2e-6 * abs(o_total)
    assert np.abs(parts - o_parts).max() <= 2e-6 * np.abs(o_parts).max()
    assert np.abs(grad - o_grad).max() <= 2e-6 * np.abs(o_grad).max()
    # masked rows: exactly zero
    m = np.isnan(gt[:, :, 108]) | np.isnan(gt[:, :, 109])
    assert np.all(grad[m][:, 108:111] == 0.0)
    mc = np.isnan(gt[:, :, 111:]).any(axis=2)
    assert np.all(grad[mc][:, 111:] == 0.0)
    # deterministic
    total2, _, grad2 = run_fused(pred, gt)
    assert total2 == total and np.array_equal(grad, grad2)


def test_reference_style_loop_with_separate_functions():
    """The reference loop verbatim (train_model.py:177-189) on column-slice views, gradients accumulated by autograd."""
    lu = tip_amd.learning_utils
    z = np.load(GOLD)
    pred, gt = case_inputs(z, "b2t40")
    y_pred = torch.tensor(pred).cuda().requires_grad_(True)
    y = torch.tensor(gt).cuda()
    n_sbps = N_SBPS
    loss_j = lu.loss_jerk(y_pred[:, :, :-3 - (n_sbps * 4)])
    y_pred2 = y_pred.reshape(-1, y_pred.size()[-1])
    y2 = y.reshape(-1, y.size()[-1])
    loss_q = lu.loss_q_only_2axis(y2[:, :-(n_sbps * 4)], y_pred2[:, :-(n_sbps * 4)])
    loss_c = lu.loss_constr_multi(y2[:, -(n_sbps * 4):], y_pred2[:, -(n_sbps * 4):])
    loss = loss_c + loss_q
    loss += loss_j
    loss.backward()
    ref = z["b2t40/losses"]
    assert close([loss.item(), loss_q.item(), loss_c.item(), loss_j.item()], ref, REL)
    g = z["b2t40/grad"]
    assert np.abs(y_pred.grad.cpu().numpy() - g).max() <= REL * np.abs(g).max()
    # each function alone, against its own golden
    for name, fn in (("q", lambda p: lu.loss_q_only_2axis(y2[:, :-20], p.reshape(-1, 131)[:, :-20])),
                     ("c", lambda p: lu.loss_constr_multi(y2[:, -20:], p.reshape(-1, 131)[:, -20:])),
                     ("j", lambda p: lu.loss_jerk(p[:, :, :-23]))):
        pm, gm = case_inputs(z, "mix")
        y2 = torch.tensor(gm).cuda().reshape(-1, 131)
        p = torch.tensor(pm).cuda().requires_grad_(True)
        l = fn(p)
        l.backward()
        assert close([l.item()], z[f"mix_{name}/loss"], REL), name
        gr = z[f"mix_{name}/grad"]
        assert np.abs(p.grad.cpu().numpy() - gr).max() <= REL * max(1.0, np.abs(gr).max()), name


def test_upstream_gradient_scales_the_result():
    pred, gt = make_case("mix", 4, 9, 21)
    _, _, g1 = run_fused(pred, gt)
    yp = torch.tensor(pred).cuda().requires_grad_(True)
    (tip_amd.learning_utils.train_loss(yp, torch.tensor(gt).cuda(), N_SBPS) * 3.0).backward()
    assert np.abs(yp.grad.cpu().numpy() - 3.0 * g1).max() <= 1e-6 * np.abs(g1).max()
    # without the jerk term
    t, parts, g = run_fused(pred, gt, with_jerk=False)
    o = loss_oracle.train_loss(pred, gt, N_SBPS)
    assert parts[2] == 0.0 and abs(t - (o[1][0] + o[1][1])) <= 2e-6 * abs(t)


def test_loss_drives_the_hip_training_step():
    """y_pred = model(x_imu, x_s) in train mode; train_loss(...).backward() reaches every parameter through the HIP backward,
    and equals feeding the oracle's d loss / d y_pred into y_pred.backward()."""
    from tip_amd import synth
    cfg = synth.PAPER
    torch.manual_seed(0)
    model = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                                      dropout=0.0, in_dropout=0.0, past_state_dropout=0.0, with_acc_sum=True).cuda()
    model.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=3).items()})
    model.train()
    x_imu, x_s = synth.make_inputs(cfg, 8, 40, seed=77)
    _, gt = make_case("e2e", 8, 40, 5)
    xi, xs, y = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda(), torch.tensor(gt).cuda()
    torch.manual_seed(1)
    y_pred = model(xi, xs)
    loss = tip_amd.learning_utils.train_loss(y_pred, y, N_SBPS)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in model.named_parameters()}
    assert all(torch.isfinite(g).all() for g in grads.values())
    o_total, _, o_grad = loss_oracle.train_loss(y_pred.detach().cpu().numpy(), gt, N_SBPS)
    assert abs(loss.item() - o_total) <= 2e-6 * abs(o_total)
    model.zero_grad()
    torch.manual_seed(1)
    model(xi, xs).backward(torch.tensor(o_grad, dtype=torch.float32).cuda())
    for k, p in model.named_parameters():
        scale = max(1e-6, grads[k].abs().max().item())
        assert (p.grad - grads[k]).abs().max().item() <= 1e-4 * scale, k


def test_loud_failures_and_argument_checks():
    lu = tip_amd.learning_utils
    pred, gt = make_case("mix", 2, 5, 0)
    with pytest.raises(TypeError):
        lu.train_loss(torch.tensor(pred), torch.tensor(gt))                      # CPU tensors: no fallback
    with pytest.raises(TypeError):                                                   # precision mixes are refused, nothing is converted
        lu.train_loss(torch.tensor(pred).cuda().double(), torch.tensor(gt).cuda())
    with pytest.raises(TypeError):
        lu.train_loss(torch.tensor(pred).cuda().half(), torch.tensor(gt).cuda().half())
    with pytest.raises(AssertionError):
        lu.loss_q_only_2axis(torch.zeros(4, 110).cuda(), torch.zeros(4, 110).cuda())   # the reference asserts 18*6+3
    lib = tip_amd.lib.load()
    p, g = torch.tensor(pred).cuda(), torch.tensor(gt).cuda()
    stats = torch.zeros(16, device="cuda")
    ws = torch.zeros(64, dtype=torch.float64, device="cuda")
    args = lambda n_vel, wsb: (p.data_ptr(), 131, g.data_ptr(), 131, 2, 5, 108, n_vel, 5, 7, stats.data_ptr(), ws.data_ptr(), wsb, None)
    assert lib.tip_loss_forward(*args(2, 512)) == -1          # n_vel must be 0 or 3
    assert lib.tip_loss_forward(*args(3, 8)) == -4            # workspace too small
    nb = ctypes.c_size_t()
    assert lib.tip_loss_ws_bytes(2, 5, ctypes.byref(nb)) == 0 and nb.value == 64
    assert lib.tip_loss_forward(p.data_ptr(), 100, g.data_ptr(), 131, 2, 5, 108, 3, 5, 7, stats.data_ptr(), ws.data_ptr(), 512, None) == -1
    assert lib.tip_loss_forward(p.data_ptr(), 131, None, 0, 2, 5, 108, 3, 5, 7, stats.data_ptr(), ws.data_ptr(), 512, None) == -1


@pytest.mark.parametrize("tag", ["c17", "c24", "c64"])
def test_constraint_loss_with_more_than_16_constraints_per_row(tag):
    """loss_constr_multi accepts any (bs, 4*N) (learning_utils.py:13-35); the flag columns of N > 16 constraints span more
    than one sweep of the workgroup's threads over a 16-row tile: every BCE term and every gradient entry must be there.
    Golden = the REAL reference function (make_loss_golden.py WIDE_CASES)."""
    from make_loss_golden import WIDE_CASES, make_constr_case
    z = np.load(GOLD)
    gt, pred = make_constr_case(*WIDE_CASES[tag])
    rb = torch.tensor(pred).cuda().requires_grad_(True)
    loss = tip_amd.learning_utils.loss_constr_multi(torch.tensor(gt).cuda(), rb)
    loss.backward(inputs=[rb])
    assert close([float(loss)], z[tag + "/loss"], REL), (float(loss), z[tag + "/loss"])
    g, gr = rb.grad.cpu().numpy(), z[tag + "/grad"]
    assert np.isfinite(g).all() and np.all(g[7] == 0.0)
    assert np.abs(g - gr).max() <= REL * max(1.0, np.abs(gr).max())
    o_loss, o_grad = loss_oracle.loss_constr_multi(gt, pred)
    assert abs(float(loss) - o_loss) <= 2e-6 * abs(o_loss), (float(loss), o_loss)
    assert np.abs(g - o_grad).max() <= 2e-6 * np.abs(o_grad).max()


def test_fused_loss_fp64_matches_oracle():
    """train_model.py --double: predictions and targets in float64 -> tip_loss_forward_f64 / tip_loss_backward_f64; total, the
    three parts and the gradient against the fp64 oracle (sigmoid in fp64 too) to fp64 rounding; masked rows exactly zero."""
    lu = tip_amd.learning_utils
    for tag, (B, T, seed) in {"full": (64, 40, 3), "short": (5, 3, 4), "one": (1, 1, 5)}.items():
        pred, gt = make_case("full", B, T, seed)
        pred, gt = pred.astype(np.float64), gt.astype(np.float64)
        yp = torch.tensor(pred).cuda().requires_grad_(True)
        total, parts = lu.train_loss(yp, torch.tensor(gt).cuda(), N_SBPS, return_parts=True)
        assert total.dtype == torch.float64
        total.backward()
        o_total, o_parts, o_grad = loss_oracle.train_loss(pred, gt, N_SBPS, f32_sigmoid=False)
        g = yp.grad.cpu().numpy()
        if np.isnan(o_total):
            assert np.isnan(float(total))
            continue
        assert abs(float(total) - o_total) <= 1e-12 * abs(o_total), (tag, float(total), o_total)
        assert np.allclose(parts.cpu().numpy(), o_parts, rtol=1e-12, atol=0, equal_nan=True)
        assert np.abs(g - o_grad).max() <= 1e-12 * max(np.abs(o_grad).max(), 1e-300), tag
        # the three reference functions one by one on column slices, as train_model.py:177-185 calls them
        y2 = torch.tensor(pred).cuda().requires_grad_(True)
        yf, gf = y2.reshape(-1, y2.shape[-1]), torch.tensor(gt).cuda().reshape(-1, gt.shape[-1])
        lj = lu.loss_jerk(y2[:, :, :-3 - 4 * N_SBPS])
        lq = lu.loss_q_only_2axis(gf[:, :-4 * N_SBPS], yf[:, :-4 * N_SBPS])
        lc = lu.loss_constr_multi(gf[:, -4 * N_SBPS:], yf[:, -4 * N_SBPS:])
        (lc + lq + lj).backward()
        assert np.allclose([float(lq), float(lc), float(lj)], o_parts, rtol=1e-12, atol=0, equal_nan=True)
        assert np.abs(y2.grad.cpu().numpy() - o_grad).max() <= 1e-12 * max(np.abs(o_grad).max(), 1e-300)
    with pytest.raises(TypeError):                              # no silent conversion: fp32 prediction against fp64 targets
        lu.train_loss(torch.zeros(2, 4, 131, device="cuda"), torch.zeros(2, 4, 131, device="cuda", dtype=torch.float64), N_SBPS)

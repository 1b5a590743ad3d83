"""CPU: the training-loss oracle (oracle/loss_oracle.py) against losses and autograd gradients of the REAL reference
functions (tests/golden/make_loss_golden.py -> tip_loss_golden.npz; fp32 torch)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_loss_golden import CASES, N_SBPS, make_case   # noqa: E402  (synthetic input generator: data, not reference code)
from oracle import loss_oracle                           # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "tip_loss_golden.npz")
REL = 2e-5    # the golden side is fp32 torch (fp32 means over up to 8640 squares); the oracle is fp64


def case_inputs(z, tag):
    B, T, seed = CASES[tag]
    pred, gt = make_case(tag, B, T, seed)
    chk = np.array([pred.astype(np.float64).sum(), np.nansum(gt.astype(np.float64)), np.isnan(gt).sum()])
    assert np.allclose(chk, z[tag + "/insum"], rtol=0, atol=1e-9), "synthetic inputs drifted from the golden run"
    return pred, gt


def close(a, b, rel=REL):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    m = ~np.isnan(a)
    scale = max(1.0, float(np.abs(b[m]).max())) if m.any() else 1.0
    return (not m.any()) or float(np.abs(a[m] - b[m]).max()) <= rel * scale


def test_train_loss_matches_reference():
    z = np.load(GOLD)
    for tag in CASES:
        pred, gt = case_inputs(z, tag)
        total, parts, grad = loss_oracle.train_loss(pred, gt, N_SBPS)
        ref = z[tag + "/losses"]
        assert close([total, *parts], ref), (tag, total, parts, ref)
        g = z[tag + "/grad"]
        assert np.isfinite(grad).all()
        assert np.abs(grad - g).max() <= REL * max(1.0, np.abs(g).max()), tag


def test_nan_cases_are_nan_like_the_reference():
    z = np.load(GOLD)
    assert np.isnan(z["t3/losses"][[0, 3]]).all() and np.isfinite(z["t3/losses"][[1, 2]]).all()      # no jerk sample at T = 3
    assert np.isnan(z["allmask/losses"][[0, 1]]).all() and np.isfinite(z["allmask/losses"][[2, 3]]).all()
    # rows dropped by a mask get exactly zero gradient, in the reference and in the oracle
    pred, gt = case_inputs(z, "allmask")
    assert np.all(z["allmask/grad"][:, :, 108:111] == 0.0)
    assert np.all(loss_oracle.train_loss(pred, gt, N_SBPS)[2][:, :, 108:111] == 0.0)


def test_saturated_sigmoid_costs_100_and_has_no_gradient():
    z = np.load(GOLD)
    pred, gt = case_inputs(z, "sat")
    g = z["sat/grad"][:, :, 111::4]
    x = pred[:, :, 111::4]
    dead = (x >= 30.0) | (x <= -100.0)                        # sigmoid == 1.0f / 0.0f exactly; at -30 p = 9.4e-14 survives
    assert dead.any() and np.all(g[dead] == 0.0)              # torch: BCE backward (eps 1e-12) times p (1 - p) = 0
    assert np.all(loss_oracle.train_loss(pred, gt, N_SBPS)[2][:, :, 111::4][dead] == 0.0)
    # the fp64-sigmoid variant differs there, which is why the oracle restates the fp32 rounding
    assert abs(loss_oracle.train_loss(pred, gt, N_SBPS, f32_sigmoid=False)[1][1] - z["sat/losses"][2]) > 1.0


def test_separate_functions_match_reference():
    z = np.load(GOLD)
    pred, gt = case_inputs(z, "mix")
    p2, g2 = pred.reshape(-1, 131), gt.reshape(-1, 131)
    lq, gq = loss_oracle.loss_q_only_2axis(g2[:, :-20], p2[:, :-20])
    lc, gc = loss_oracle.loss_constr_multi(g2[:, -20:], p2[:, -20:])
    lj, gj = loss_oracle.loss_jerk(pred[:, :, :-23])
    for name, l, g, sl in (("q", lq, gq.reshape(3, 12, 111), np.s_[:, :, :111]), ("c", lc, gc.reshape(3, 12, 20), np.s_[:, :, 111:]),
                           ("j", lj, gj, np.s_[:, :, :108])):
        assert close([l], z[f"mix_{name}/loss"]), name
        full = np.zeros((3, 12, 131))
        full[sl] = g
        ref = z[f"mix_{name}/grad"]
        assert np.abs(full - ref).max() <= REL * max(1.0, np.abs(ref).max()), name


def test_constraint_loss_with_many_constraints_matches_reference():
    """loss_constr_multi for N = 17 / 24 / 64 constraints per row (the model emits 5): reference-generated golden."""
    from make_loss_golden import WIDE_CASES, make_constr_case
    z = np.load(GOLD)
    for tag, args in WIDE_CASES.items():
        gt, pred = make_constr_case(*args)
        chk = np.array([pred.astype(np.float64).sum(), np.nansum(gt.astype(np.float64)), np.isnan(gt).sum()])
        assert np.allclose(chk, z[tag + "/insum"], rtol=0, atol=1e-9)
        l, g = loss_oracle.loss_constr_multi(gt, pred)
        assert close([l], z[tag + "/loss"]), tag
        ref = z[tag + "/grad"]
        assert np.abs(g - ref).max() <= REL * max(1.0, np.abs(ref).max()), tag
        assert np.all(g[7] == 0.0) and np.all(ref[7] == 0.0)

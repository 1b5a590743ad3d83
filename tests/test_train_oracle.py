"""CPU: the training-step oracle (oracle/train_oracle.py) against digests of the REAL reference's gradients
(tests/golden/make_train_golden.py), and the module's torch-op composite against the same oracle."""
import os

import numpy as np
import pytest
import torch

import tip_amd
from tip_amd import synth
from oracle import train_oracle
from test_host_cpu import make_model, load_synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "tip_train_golden.npz")
CASES = {"train_s0_B2_T40": 0, "train_s1_B3_T17": 1, "train_s2_B12_T40": 2}   # tag -> weight / input seed


def case_inputs(z, tag):
    """(x_imu, x_s, cot) of a golden case: stored for the small ones, regenerated from the generator's seeds (and checked
    against its checksum) for the larger one."""
    if tag + "/x_imu" in z.files:
        return z[tag + "/x_imu"], z[tag + "/x_s"], z[tag + "/cot"]
    cfg = synth.PAPER
    seed = CASES[tag]
    B, T = int(tag.split("_B")[1].split("_")[0]), int(tag.split("_T")[1])
    x_imu, x_s = synth.make_inputs(cfg, B, T, seed=500 + seed)
    cot = synth.normal(900 + seed, "train/cot", B * T * cfg["size_s"]).reshape(B, T, cfg["size_s"]).astype(np.float32)
    chk = np.array([x_imu.astype(np.float64).sum(), np.nansum(x_s.astype(np.float64)), cot.astype(np.float64).sum()])
    assert np.allclose(chk, z[tag + "/insum"], rtol=0, atol=1e-6), "synthetic inputs drifted from the golden run"
    return x_imu, x_s, cot


def check_y(z, tag, y, tol):
    if tag + "/y" in z.files:
        assert np.abs(y - z[tag + "/y"]).max() < tol
    else:
        assert np.abs(y[-1] - z[tag + "/y_last_window"]).max() < tol


def digest_close(name, got, want, rtol=2e-4):
    """digest entries are sums over up to 262144 terms of fp32 reference gradients: compare relative to the tensor's
    L2 norm (sqrt of entry 1), which is what bounds their rounding."""
    scale = np.sqrt(max(want[1], 1e-30))
    n_dir = scale * 1.0
    assert abs(got[1] - want[1]) <= 4 * rtol * want[1] + 1e-12, (name, "sumsq", got[1], want[1])
    assert abs(got[2] - want[2]) <= rtol * n_dir * 8 + 1e-9, (name, "projection", got[2], want[2])
    assert np.all(np.abs(got[3:] - want[3:]) <= rtol * scale + 1e-7), (name, "entries", got[3:], want[3:])


@pytest.mark.parametrize("tag", list(CASES))
def test_oracle_matches_reference_gradients(tag):
    z = np.load(GOLD)
    cfg = synth.PAPER
    w = synth.make_weights(cfg, seed=CASES[tag])
    x_imu, x_s, cot = case_inputs(z, tag)
    y, grads = train_oracle.step(cfg, w, x_imu, x_s, cot)
    check_y(z, tag, y, 5e-6)
    names = list(w.keys())
    gn = np.sqrt(sum((g.astype(np.float64) ** 2).sum() for g in grads.values()))
    assert abs(gn - z[tag + "/gnorm"][0]) < 1e-4 * gn
    for i, n in enumerate(names):
        digest_close(n, train_oracle.digest(n, grads[n]), z[tag + "/digests"][i])


COND_GOLD = os.path.join(os.path.dirname(__file__), "golden", "tip_train_cond_golden.npz")
COND_GAINS = (2.0, 3.0)


@pytest.mark.parametrize("gain", COND_GAINS)
def test_oracle_matches_reference_gradients_outside_random_init(gain):
    """The training conditioning cases (tests/golden/make_train_golden.py --cond: weights x gain 2 / 3): the fp64 oracle against
    digests of the reference's fp64 gradients (tight) and of its fp32 gradients (to the reference's own fp32 noise)."""
    z = np.load(COND_GOLD)
    cfg = synth.PAPER
    w = synth.make_weights(cfg, seed=0, gain=gain)
    tag = f"traincond_g{gain:g}"
    y, grads = train_oracle.step(cfg, w, z["x_imu"], z["x_s"], z["cot"])
    assert np.abs(y - z[tag + "/y64"]).max() < 1e-10
    names = list(w.keys())
    gn = np.sqrt(sum((g.astype(np.float64) ** 2).sum() for g in grads.values()))
    assert abs(gn - z[tag + "/gnorm"][0]) < 1e-9 * gn
    noise = z[tag + "/ref_grad_noise"]
    assert noise.max() < 1e-4
    for i, n in enumerate(names):
        d = train_oracle.digest(n, grads[n])
        digest_close(n, d, z[tag + "/digests64"][i], rtol=1e-9)
        digest_close(n, d, z[tag + "/digests"][i], rtol=max(2e-4, 20 * float(noise[i])))


def test_dropout_hash_statistics_and_determinism():
    a = train_oracle.drop_scale(1234, 5, 200000, 0.1)
    b = train_oracle.drop_scale(1234, 5, 200000, 0.1)
    c = train_oracle.drop_scale(1234, 6, 200000, 0.1)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    keep = (a > 0).mean()
    assert abs(keep - 0.9) < 0.004
    assert np.allclose(a[a > 0], 1.0 / 0.9)
    assert np.all(train_oracle.drop_scale(1, 0, 10, 0.0) == 1.0)


def test_module_composite_matches_oracle_gradients():
    """The torch-op composite the module differentiates when the HIP training path does not apply (CPU tensors,
    unsupported configuration) computes the same gradients as the oracle."""
    cfg = synth.PAPER
    m = make_model(cfg)
    w = load_synth(m, cfg, 0)
    m = m.double().train()
    m.ENCODER_DROPOUT = 0.0
    x_imu, x_s = synth.make_inputs(cfg, 2, 9, seed=3)
    cot = np.random.RandomState(0).randn(2, 9, cfg["size_s"])
    y = m(torch.tensor(x_imu, dtype=torch.float64), torch.tensor(x_s, dtype=torch.float64))
    (y * torch.tensor(cot)).sum().backward()
    yo, go = train_oracle.step(cfg, w, x_imu, x_s, cot)
    assert np.abs(y.detach().numpy() - yo).max() < 1e-9
    for n, p in m.named_parameters():
        assert np.abs(p.grad.numpy() - go[n]).max() <= 1e-8 * (1 + np.abs(go[n]).max()), n


def test_train_oracle_without_rnn_and_other_widths_matches_forward_oracle():
    """The training oracle's forward (autograd graph) for the configurations the reference's constructor also builds
    (simple_transformer_with_state.py:43-46 with_rnn=False; --rnn_nhid other than 512, train_model.py:47-48) against the C / numpy
    forward oracle that the reference's goldens pin (tiny_nornn, tiny cases): same function, so the gradients it yields are those
    of the pinned forward."""
    import torch
    from oracle import oracle
    for cfg in (dict(synth.TINY, with_rnn=False), dict(synth.PAPER, with_rnn=False, tf_layers=1),
                dict(synth.PAPER, rnn_hid_size=192, tf_layers=1)):
        w = synth.make_weights(cfg, seed=4)
        x_imu, x_s = synth.make_inputs(cfg, 2, 9, seed=5)
        params = {k: torch.tensor(np.asarray(v), dtype=torch.float64) for k, v in w.items()}
        y = train_oracle.forward(cfg, params, x_imu, x_s).numpy()
        yo = oracle.forward(cfg, w, x_imu, x_s, dtype=np.float64)
        assert np.abs(y - yo).max() < 1e-12, cfg

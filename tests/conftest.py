import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden", "tip_forward_golden.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "handoff_fault: the test makes a hand-off wait give up on purpose (TIP_OPT_FAULT_INJECT)")


@pytest.fixture(scope="session")
def golden():
    z = np.load(GOLDEN)
    cases = {}
    for k in z.files:
        tag, name = k.split("/")
        cases.setdefault(tag, {})[name] = z[k]
    return cases


COND_GOLDEN = os.path.join(ROOT, "tests", "golden", "tip_cond_golden.npz")


@pytest.fixture(scope="session")
def cond_golden():
    """Conditioning sweep (tests/golden/make_golden.py --cond): {"x_imu", "x_s", "cases": {tag: {y32, y32_alt, y64, wsum,
    gain_ln, noise}}}; `noise` = the reference's own fp32 rounding noise max(|y32 - y64|, |y32_alt - y64|)."""
    z = np.load(COND_GOLDEN)
    cases = {}
    for k in z.files:
        if "/" not in k:
            continue
        tag, name = k.split("/")
        cases.setdefault(tag, {})[name] = z[k]
    for c in cases.values():
        c["noise"] = max(float(np.abs(c["y32"] - c["y64"]).max()), float(np.abs(c["y32_alt"] - c["y64"]).max()))
    # tags "g<gain>_ln<gamma>": paper configuration on the stored windows; "<config>__g.._ln..": another width, inputs regenerated
    paper = {t: c for t, c in cases.items() if "__" not in t}
    other = {t: c for t, c in cases.items() if "__" in t}
    return {"x_imu": z["x_imu"], "x_s": z["x_s"], "cases": paper, "other": other}


def cond_other_case(tag, c):
    """(cfg, weights, x_imu, x_s) of a non-paper conditioning case (tests/golden/make_golden.py --cond)."""
    import tip_amd
    s = tip_amd.synth
    name = tag.split("__")[0]
    cfg = {"scaled2": dict(s.SCALED, tf_layers=2), "tiny": s.TINY}[name]
    g, lg = (float(v) for v in c["gain_ln"])
    B, T = (int(v) for v in c["shape"])
    w = s.make_weights(cfg, seed=0, gain=g, ln_gamma=lg)
    x_imu, x_s = s.make_inputs(cfg, B, T, seed=4321)
    return cfg, w, x_imu, x_s


def cfg_for_tag(tag):
    import tip_amd
    s = tip_amd.synth
    if tag.startswith("paper"):
        return s.PAPER
    if tag.startswith("tiny_nornn"):
        return dict(s.TINY, with_rnn=False)
    if tag.startswith("tiny_noacc"):
        return dict(s.TINY, with_acc_sum=False)
    if tag.startswith("tiny"):
        return s.TINY
    raise KeyError(tag)


def seed_for_tag(tag):
    return int(tag.split("_s")[1].split("_")[0])


@pytest.fixture(autouse=True)
def _no_handoff_timeouts(request):
    """Around every GPU test: no inter-workgroup hand-off may time out (tip_spin_timeouts must not move) — except in the
    tests that inject the fault on purpose (marker `handoff_fault`)."""
    gpu = request.node.get_closest_marker("gpu") is not None
    if gpu:
        from tip_amd import lib as tlib
        before = tlib.spin_timeouts()
    yield
    if gpu and request.node.get_closest_marker("handoff_fault") is None:
        assert tlib.spin_timeouts() == before, "a cluster hand-off gave up: outputs of some launch were invalid"

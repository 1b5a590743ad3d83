"""Pins the numpy restatement of the runner's model-facing half (oracle/streaming_oracle.py) against a 70-frame trace
of the REAL RTRunnerMin (tests/golden/make_runner_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from conftest import ROOT
from oracle.streaming_oracle import StreamOracle

RUNNER_GOLDEN = os.path.join(ROOT, "tests", "golden", "tip_runner_golden.npz")


@pytest.fixture(scope="module")
def trace():
    z = np.load(RUNNER_GOLDEN)
    out = {}
    for k in z.files:
        tag, name = k.split("/")
        out.setdefault(tag, {})[name] = z[k]
    return out


def test_runner_protocol(trace):
    for tag, tr in trace.items():
        assert int(tr["n_calls"][0]) == 65 and tr["raw_imu"].shape == (70, 72)   # first 5 frames prime the smoother
        assert list(tr["call_T"][:40]) == list(range(1, 41)) and set(tr["call_T"][40:]) == {40}


def test_oracle_matches_reference_runner_teacher_forced(trace):
    """Feed the oracle the outputs the reference model produced (so the model itself is out of the loop) and compare
    every tensor the runner hands to the model and every history row it feeds back."""
    for tag, tr in trace.items():
        o = StreamOracle(tr["s_init"])
        k = 0
        for t in range(tr["raw_imu"].shape[0]):
            if not o.ingest(tr["raw_imu"][t]):
                assert np.array_equal(tr["qdq"][t], tr["s_init"])      # :125-128
                continue
            x_imu, x_s = o.build_inputs()
            assert x_imu.shape[0] == tr["call_T"][k]
            assert np.abs(x_imu[-1] - tr["x_imu_last_rows"][k]).max() < 1e-5
            assert np.abs(x_s[-1] - tr["x_s_last_rows"][k]).max() < 1e-5
            if f"x_imu_call{k}" in tr:
                assert np.abs(x_imu - tr[f"x_imu_call{k}"]).max() < 1e-5, (tag, k)
                assert np.abs(x_s - tr[f"x_s_call{k}"]).max() < 1e-5, (tag, k)
            s_rest, c_t = o.consume(tr["y_last_rows"][k])
            assert np.abs(np.array(o.hist[-1]) - tr["hist_last"][t]).max() < 1e-9, (tag, t)
            assert np.abs(s_rest - tr["qdq"][t][3:]).max() < 1e-9
            assert np.array_equal(c_t, tr["ct"][t])
            k += 1
        assert k == 65

"""CPU: the train-set combiner / window sampler oracle (oracle/data_oracle.py) against the output of the REAL reference
code (tests/golden/make_data_golden.py -> tip_data_golden.npz)."""
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_data_golden import motion_files, RATES, LENS   # noqa: E402  (synthetic motion generator: data, not reference code)
from oracle import data_oracle                             # noqa: E402

GOLD = os.path.join(os.path.dirname(__file__), "golden", "tip_data_golden.npz")


def combined_by_oracle(z):
    """Run the oracle over the same files the reference combined; returns (IMU, SUM, S, info)."""
    IMU, SUM, S, info = [], [], [], []
    start, kept = 0, 0
    dnames = list(LENS.keys())
    for dname, i, imu, s, c in motion_files():
        m = min(len(s), len(imu))
        if m <= data_oracle.ACC_SUM_WIN_LEN:          # preprocess_and_combine_syn_amass.py:68-70
            continue
        a, b, cc = data_oracle.combine_sequence(imu, s, c, z["biases"][kept], nan_root_vel="DIP" in dname)
        kept += 1
        IMU.append(a); SUM.append(b); S.append(cc)
        info.append([start, start + len(a), RATES[dnames.index(dname)]])
        start += len(a)
    return np.concatenate(IMU), np.concatenate(SUM), np.concatenate(S), np.array(info)


def test_combiner_matches_reference():
    z = np.load(GOLD)
    IMU, SUM, S, info = combined_by_oracle(z)
    assert np.array_equal(info, z["info"])
    assert IMU.dtype == np.float32 and IMU.shape == z["IMU"].shape
    assert np.abs(IMU - z["IMU"]).max() < 2e-6          # fp64 math, fp32 storage: at most an ulp apart
    assert np.abs(SUM - z["SUM"]).max() < 2e-6
    assert np.array_equal(np.isnan(S), np.isnan(z["S"]))   # augmented-DIP root velocity is NaN (:61-62)
    assert np.nanmax(np.abs(S - z["S"])) < 2e-6


def test_window_sampler_matches_reference():
    z = np.load(GOLD)
    random.seed(99)
    ends = data_oracle.sample_ends(z["info"], 40)
    assert len(ends) == int(z["n_windows"][0])
    for k, t in enumerate(ends):
        x_imu, x_s, y = data_oracle.window(z["IMU"], z["SUM"], z["S"], t, 40)
        sums = [np.nansum(a.astype(np.float64)) for a in (x_imu, x_s, y)]
        assert np.allclose(sums, z["win/sums"][k], rtol=0, atol=1e-9), k
        if k < 3:
            assert np.array_equal(x_imu, z["win/x_imu"][k])
            assert np.array_equal(np.nan_to_num(x_s, nan=9.0), np.nan_to_num(z["win/x_s"][k], nan=9.0))
            assert np.array_equal(np.nan_to_num(y, nan=9.0), np.nan_to_num(z["win/y"][k], nan=9.0))

"""GPU: the on-device train-set combiner and window gather (csrc/tip_data.hip through tip_amd.data) against the output of
the REAL reference code (tip_data_golden.npz) and the oracle."""
import os
import random
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
from make_data_golden import motion_files, RATES, LENS   # noqa: E402
import tip_amd                                             # noqa: E402
from oracle import data_oracle                             # noqa: E402
from test_data_oracle import GOLD                          # noqa: E402

pytestmark = pytest.mark.gpu
TOL = 2e-6    # fp64 math on both sides, float32 storage: results may differ in the last float32 bit


def _combined(z):
    files, rates, dip = [], [], []
    dn = list(LENS.keys())
    for dname, i, imu, s, c in motion_files():
        files.append({"imu": imu, "nimble_qdq": s, "constrs": c})
        rates.append(RATES[dn.index(dname)])
        dip.append("DIP" in dname)
    return tip_amd.data.combine_motions(files, rates, dip, biases=z["biases"])


def test_combiner_matches_reference_output():
    assert torch.cuda.is_available()
    z = np.load(GOLD)
    cmb = _combined(z)
    assert np.array_equal(cmb.info, z["info"])             # includes dropping the too-short file and the 1-frame mismatch
    IMU, SUM, S = (t.cpu().numpy() for t in (cmb.IMU, cmb.SUM, cmb.S))
    assert np.abs(IMU - z["IMU"]).max() < TOL
    assert np.abs(SUM - z["SUM"]).max() < TOL
    assert np.array_equal(np.isnan(S), np.isnan(z["S"]))
    assert np.nanmax(np.abs(S - z["S"])) < TOL


def test_window_dataset_matches_reference_sampling_and_slices():
    z = np.load(GOLD)
    IMU, SUM, S = (torch.tensor(z[k]).cuda() for k in ("IMU", "SUM", "S"))
    random.seed(99)
    ds = tip_amd.data.TrainSubDataset.from_arrays(40, z["info"], IMU, S, IMU_sum=SUM)
    assert len(ds) == int(z["n_windows"][0])
    x_imu, x_s, y = (t.cpu().numpy() for t in ds.batch(range(len(ds))))
    sums = np.stack([[np.nansum(a[k].astype(np.float64)) for a in (x_imu, x_s, y)] for k in range(len(ds))])
    assert np.allclose(sums, z["win/sums"], rtol=0, atol=1e-9)
    for k in range(3):   # a gather is a copy: bit-exact
        assert np.array_equal(x_imu[k], z["win/x_imu"][k])
        assert np.array_equal(np.nan_to_num(x_s[k], nan=9.0), np.nan_to_num(z["win/x_s"][k], nan=9.0))
        assert np.array_equal(np.nan_to_num(y[k], nan=9.0), np.nan_to_num(z["win/y"][k], nan=9.0))
    a, b, c = ds[5]                                            # the reference's item protocol
    assert np.array_equal(a.cpu().numpy(), x_imu[5]) and a.shape == (40, 90) and b.shape == (40, 131) and c.shape == (40, 131)
    # without acc-sum features (training_data_loader.py:60 with_acc_sum=False)
    random.seed(99)
    ds2 = tip_amd.data.TrainSubDataset.from_arrays(40, z["info"], IMU, S, with_acc_sum=False)
    xi2, _, _ = ds2.batch([0, 1, 2])
    assert xi2.shape == (3, 40, 72) and np.array_equal(xi2.cpu().numpy(), x_imu[:3, :, :72])


def test_large_sequence_against_oracle_and_training_handoff():
    """A long sequence (20 000 frames) against the oracle, then gathered batches go straight into the HIP training step."""
    from make_data_golden import synth_motion
    imu, s, c = synth_motion(20000, 77)
    bias = np.linspace(-0.1, 0.1, 18)
    cmb = tip_amd.data.combine_motions([{"imu": imu, "nimble_qdq": s, "constrs": c}], [60], biases=bias[None])
    a, b, cc = data_oracle.combine_sequence(imu, s, c, bias)
    assert np.abs(cmb.IMU.cpu().numpy() - a).max() < TOL
    assert np.abs(cmb.SUM.cpu().numpy() - b).max() < 2e-5      # sums of 40 terms around 1e1: a few float32 ulps
    assert np.abs(cmb.S.cpu().numpy() - cc).max() < TOL
    random.seed(1)
    ds = tip_amd.data.TrainSubDataset.from_arrays(40, cmb.info, cmb.IMU, cmb.S, IMU_sum=cmb.SUM)
    x_imu, x_s, y = ds.batch(range(min(64, len(ds))))
    from test_host_cpu import make_model, load_synth
    from tip_amd import synth
    m = make_model(synth.PAPER)
    load_synth(m, synth.PAPER, 0)
    m = m.cuda().train()
    yp = m(x_imu, x_s)
    loss = ((yp - y) ** 2).mean()
    loss.backward()
    assert torch.isfinite(loss) and all(torch.isfinite(p.grad).all() for p in m.parameters())
    assert type(yp.grad_fn).__name__.startswith("_HipTrainFunction")

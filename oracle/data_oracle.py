"""TEST INFRASTRUCTURE — numpy restatement of the reference's train-set combiner and window sampler
(SURVEY.md section 8 row f-3).  Only tests/ may import this; it is the checker for csrc/tip_data.hip.

  combine_sequence : /root/reference/preprocess_and_combine_syn_amass.py:73-101 for one motion file
      trim 4 frames at both ends (:73-75), 11-tap moving average of the 18 acceleration channels with edge
      replication (:81-83, scipy uniform_filter1d mode="nearest"), constant bias (:85, drawn by the caller), root-local
      IMU frame (:86, data_utils.py:190-219), running 40-frame sum of the local accelerations / 15 (:90-93),
      axis-angle -> first two rotation-matrix columns for the 18 joints + root velocity (:96, data_utils.py:182-187),
      SBP channels appended (:99-100, :127-128); everything cast to float32.
  window           : /root/reference/training_data_loader.py:53-58,72-86 — the (x_imu, x_s, y) triple of one sampled
      end frame t: IMU[t-T:t] | SUM[t-T:t], S[t-T:t], S[t-T+1:t+1].
  sample_ends      : training_data_loader.py:41-52 — which end frames one epoch draws (python `random`, seeded outside).

fairmotion's A2R is scipy's Rotation.from_rotvec().as_matrix() (fairmotion is not vendored: that dependency is
"parity unpinned"; the golden run substitutes scipy the same way).  Pinned by tests/golden/make_data_golden.py, which
runs the REAL store_imu_s_info and TrainSubDataset on synthetic motion files.
"""
from __future__ import annotations

import random

import numpy as np
from scipy.spatial.transform import Rotation

ACC_MOVING_AVE_LEN = 11     # constants.py:16
ACC_SUM_WIN_LEN = 40        # constants.py:17
ACC_SUM_DOWN_SCALE = 15.0   # constants.py:18
N_DOFS = 57                 # constants.py:24
TRIM = 4                    # preprocess_and_combine_syn_amass.py:73


def combine_sequence(imu: np.ndarray, s: np.ndarray, c: np.ndarray, bias: np.ndarray, nan_root_vel: bool = False):
    """imu [L,72], s [L,114] (nimble_qdq), c [L,20] fp64 -> (IMU [L',72], SUM [L',18], S [L',131]) float32, L' = m-8."""
    s = s.copy()
    if nan_root_vel:                                                    # :61-62 (augmented DIP files)
        s[:, N_DOFS:N_DOFS + 3] = np.nan
    m = min(len(s), len(imu))                                           # :67
    imu, s, c = imu[TRIM:m - TRIM].copy(), s[TRIM:m - TRIM], c[TRIM:m - TRIM]   # :73-75
    L = len(imu)
    acc = imu[:, 54:72]
    idx = np.clip(np.arange(L)[:, None] + np.arange(-5, 6)[None, :], 0, L - 1)  # mode="nearest"
    imu[:, 54:72] = acc[idx].mean(axis=1) + bias                       # :81-85
    root = imu[:, :9].reshape(L, 3, 3)
    inv = np.linalg.inv(root)
    other = imu[:, 9:54].reshape(L, 5, 3, 3)
    loc = np.concatenate([root.reshape(L, 9), np.einsum("nij,nsjk->nsik", inv, other).reshape(L, 45), imu[:, 54:57],
                          np.einsum("nij,nsj->nsi", inv, imu[:, 57:72].reshape(L, 5, 3)).reshape(L, 15)], axis=1)   # :86
    b = np.cumsum(loc[:, 54:72], axis=0)                                # :90-93
    b[ACC_SUM_WIN_LEN:] = b[ACC_SUM_WIN_LEN:] - b[:-ACC_SUM_WIN_LEN]
    batch_s = s[:, 3:N_DOFS + 3]                                        # :96: 54 axis-angles + 3 root velocity
    r = Rotation.from_rotvec(batch_s[:, :N_DOFS - 3].reshape(-1, 3)).as_matrix()[:, :, :2].reshape(L, -1)
    s_all = np.concatenate([r, batch_s[:, -3:], c], axis=1)             # :127-128
    return np.single(loc), np.single(b / ACC_SUM_DOWN_SCALE), np.single(s_all)


def sample_ends(info, seq_length: int):
    """training_data_loader.py:41-52.  info rows: [start, end, down_sample_rate]; uses python's `random` (seed outside)."""
    ends = []
    for start_t, end_t, rate in info:
        time_range = range(int(start_t) + seq_length, int(end_t) - 1)
        if len(time_range) == 0:
            continue
        k = int(np.maximum(round(len(time_range) / rate), 1))
        ends += random.sample(time_range, k=k)
    return ends


def window(IMU, SUM, S, t: int, T: int):
    """training_data_loader.py:55-58,72-86: (x_imu [T,72(+18)], x_s [T,131], y [T,131])."""
    x_imu = IMU[t - T:t]
    if SUM is not None:
        x_imu = np.concatenate([x_imu, SUM[t - T:t]], axis=1)
    s = S[t - T:t + 1]
    return x_imu, s[:-1], s[1:]

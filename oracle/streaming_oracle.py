"""TEST INFRASTRUCTURE — numpy/scipy restatement of the model-facing half of the reference streaming runner
(/root/reference/real_time_runner_minimal.py, RTRunnerMin), i.e. SURVEY.md section 8 rows a12/a13 = "next" row f-1.

Only tests/ (and bench legs that report a CPU baseline) may import this.  It is the checker for the on-device
streaming front/back-end in transformer-inertial-poser_amd/streaming.py.

What is restated, with the reference lines:
  * record_raw_imu              :59-76    5-frame priming, 11-tap acceleration mean, rotations delayed 5 frames
  * window build                :131-147  last <=40 smoothed frames, imu_rotate_to_local (data_utils.py:190-219),
                                          acc-sum feature over the window / 15 (constants.py:17-18), history slice
  * smooth_and_split_s_c        :87-112   6-tap 0.6^k output filter, SBP flag threshold, offsets / 5
  * pose assembly               :154-167  6D -> axis-angle (data_utils.py:164-179), root rotation from the IMU,
                                          averaging with the previous pose
  * record_state_aa_and_c       :78-85    axis-angle -> 6D history row (data_utils.py:182-187)
Not restated (stays on the CPU in the reference and here): PyBullet FK and the SBP root-translation correction
(:169-194) — they only move the root translation, which never reaches the model input
(simple_transformer_with_state.py:75 zeroes the root-velocity history; root xyz is not an input at all).

Rotation conversions: the reference calls fairmotion.ops.conversions (A2R / R2A), a thin wrapper over
scipy.spatial.transform.Rotation; fairmotion is not installed or vendored, so this file calls scipy directly.
Pinned against a 70-frame trace of the real RTRunnerMin run with that same substitution
(tests/golden/make_runner_golden.py -> tests/golden/tip_runner_golden.npz).
"""
from __future__ import annotations

import numpy as np
from scipy.spatial.transform import Rotation

DT = 1.0 / 60            # constants.py:7
IMU_N_SMOOTH = 5         # constants.py:15
ACC_WIN = 11             # constants.py:16
ACC_SUM_WIN = 40         # constants.py:17
ACC_SUM_SCALE = 15.0     # constants.py:18
N_DOFS = 57              # constants.py:24
N_SBP = 5
COEFF = 0.6 ** np.arange(6)[::-1]   # real_time_runner_minimal.py:57


def imu_rotate_to_local(batch_imu: np.ndarray) -> np.ndarray:
    """data_utils.py:190-219: the five non-root IMUs expressed in the root IMU's frame."""
    root_r = batch_imu[:, :9].reshape(-1, 3, 3)
    inv = np.linalg.inv(root_r)
    other_r = batch_imu[:, 9:54].reshape(-1, 5, 3, 3)
    other_r_local = np.einsum("nij,nsjk->nsik", inv, other_r)
    other_acc = batch_imu[:, 57:72].reshape(-1, 5, 3)
    other_acc_local = np.einsum("nij,nsj->nsi", inv, other_acc)
    return np.concatenate([root_r.reshape(-1, 9), other_r_local.reshape(-1, 45), batch_imu[:, 54:57],
                           other_acc_local.reshape(-1, 15)], axis=1)


def rot6d_to_aa(rm: np.ndarray) -> np.ndarray:
    """data_utils.py:164-179 (batch_rot_mat_2axis_to_aa): (Nj*6,) -> (Nj*3,).  Columns normalised with +1e-6,
    third column by cross product, NOT re-orthogonalised; R2A = scipy from_matrix().as_rotvec()."""
    m = rm.reshape(-1, 3, 2)
    a1 = m[:, :, 0] / (np.linalg.norm(m[:, :, 0], axis=1, keepdims=True) + 1e-6)
    a2 = m[:, :, 1] / (np.linalg.norm(m[:, :, 1], axis=1, keepdims=True) + 1e-6)
    a3 = np.cross(a1, a2)
    R = np.stack([a1, a2, a3], axis=2)
    return Rotation.from_matrix(R).as_rotvec().reshape(-1)


def aa_to_rot6d(aa: np.ndarray) -> np.ndarray:
    """data_utils.py:182-187 (batch_to_rot_mat_2axis): (Nj*3,) -> (Nj*6,), first two columns, (3x2) row-major."""
    R = Rotation.from_rotvec(aa.reshape(-1, 3)).as_matrix()
    return R[:, :, :2].reshape(-1)


class StreamOracle:
    """One stream.  step(raw_imu, model_last_row_fn) mirrors RTRunnerMin.step without the FK half."""

    def __init__(self, s_init: np.ndarray, max_len: int = 40):
        self.max_len = max_len
        self.s_init = s_init.astype(np.float64)
        self.raw, self.smoothed, self.acc_sum, self.outs = [], [], [], []
        self.last_s = None
        self.hist = [np.concatenate([aa_to_rot6d(self.s_init[3:N_DOFS]), self.s_init[N_DOFS:N_DOFS + 3],
                                     np.zeros(N_SBP * 4)])]          # :45,:78-85

    # -- front end ---------------------------------------------------------------------------------------
    def ingest(self, cur_imu: np.ndarray) -> bool:
        """:59-76.  Returns False while the smoother is still priming (the runner returns s_init, :125-128)."""
        if not self.raw:
            self.raw += [cur_imu.copy() for _ in range(IMU_N_SMOOTH)]
        self.raw.append(cur_imu.copy())
        if len(self.raw) >= ACC_WIN:
            win = np.array(self.raw[-ACC_WIN:])
            self.smoothed.append(np.concatenate([self.raw[-IMU_N_SMOOTH - 1][:54], win[:, 54:72].mean(axis=0)]))
        return len(self.smoothed) >= 1

    def build_inputs(self):
        """:131-147 -> (x_imu [T,90], x_s [T,131]) in float64 (the runner casts to float32 at :146-147)."""
        in_imu = imu_rotate_to_local(np.array(self.smoothed[-self.max_len:]))
        self.acc_sum.append(in_imu[-ACC_SUM_WIN:, 54:72].sum(axis=0))
        win = np.array(self.acc_sum[-self.max_len:]) / ACC_SUM_SCALE
        x_imu = np.concatenate([in_imu, win], axis=1)
        T = x_imu.shape[0]
        x_s = np.array(self.hist[-T:])
        self._root_R = in_imu[-1, :9].reshape(3, 3)
        return x_imu, x_s

    # -- back end ----------------------------------------------------------------------------------------
    def consume(self, y_last: np.ndarray):
        """:150-167,196: filter, decode, assemble the pose, feed the history.  Returns (s_t[3:], c_t)."""
        # The runner appends the float32 output row itself (:91).  While fewer than 6 rows exist it filters nothing and
        # works ON that row (:99,:103-110): numpy keeps float32 there, and the SBP threshold / scaling below lands
        # IN PLACE in the buffered row, which later filter windows then see.  Both quirks are part of the behaviour.
        self.outs.append(np.array(y_last, dtype=np.float32))
        if len(self.outs) >= len(COEFF):                               # :93-99
            s = (np.array(self.outs[-len(COEFF):]) * COEFF[:, None]).sum(axis=0) / COEFF.sum()
        else:
            s = self.outs[-1]
        st, c_t = s[:-N_SBP * 4], s[-N_SBP * 4:]                      # views, like the reference
        c_t[0::4] = (c_t[0::4] > 0.0) * 1.0                            # :107
        c_t[1::4] /= 5.0
        c_t[2::4] /= 5.0
        c_t[3::4] /= 5.0
        root_v = st[-3:]
        st_aa = rot6d_to_aa(st[:-3])                                   # :155
        s_t = np.zeros(2 * N_DOFS)
        s_t[N_DOFS:N_DOFS + 3] = root_v                                # :158
        s_t[6:N_DOFS] = st_aa[3:]                                      # :160
        s_t[3:6] = Rotation.from_matrix(self._root_R).as_rotvec()      # :161-162
        if self.last_s is not None:                                    # :165-166
            s_t[6:] = (s_t[6:] + self.last_s[6:]) / 2.0
        self.last_s = s_t.copy()
        self.hist.append(np.concatenate([aa_to_rot6d(s_t[3:N_DOFS]), s_t[N_DOFS:N_DOFS + 3], c_t]))   # :196, :78-85
        return s_t[3:], np.array(c_t, dtype=np.float64)

    def step(self, cur_imu, model_last_row):
        if not self.ingest(cur_imu):
            return None
        x_imu, x_s = self.build_inputs()
        y = model_last_row(x_imu.astype(np.float32), x_s.astype(np.float32))
        return self.consume(y)

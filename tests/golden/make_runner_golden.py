#!/usr/bin/env python3
"""Golden trace of the REFERENCE streaming runner (real_time_runner_minimal.py:RTRunnerMin) for the on-device
streaming front/back-end.

Runs only in the build container.  The reference runner is imported as-is; what is not installed here is stubbed at
the module boundary, exactly as SURVEY.md section 8c describes:
  * fairmotion.ops.conversions / quaternion -> scipy.spatial.transform.Rotation (xyzw quaternions; fairmotion itself
    wraps scipy, but its fork is not vendored or pinned, so the A2R/R2A branch conventions are "parity unpinned")
  * pybullet, bullet_agent.SimAgent          -> a kinematic character that returns fixed link transforms
    (FK only moves the root translation, which never reaches the model input: simple_transformer_with_state.py:75)
  * torch.Tensor.cuda                         -> identity (harness only; there is no GPU here)
The model is the reference TF_RNN_Past_State with the build's synthetic weights, dropout disabled.

Recorded per frame: raw IMU in, the tensors the runner hands to the model, the row it consumes, and the history
row it feeds back.  Only data is written.

Second fixture, tip_rotation_branches.npz: the reference's OWN two converters of the feedback path —
data_utils.batch_rot_mat_2axis_to_aa (:164-179, calls conversions.R2A) and data_utils.batch_to_rot_mat_2axis (:182-187,
conversions.A2R) — evaluated at the branch points of an axis-angle convention: angles 0, 1e-8, 1e-4, around the small-angle
switch (1e-3), pi/2, pi - 1e-2, pi - 1e-3, pi - 1e-6, pi, and beyond pi (a rotation vector longer than pi wraps), each about
an axis AND its negative (both hemispheres: where the sign of a near-pi rotation vector is decided).  The fixture records
which implementation of `fairmotion.ops.conversions` produced it (`conversions_impl`): today the scipy stand-in below, i.e.
"parity unpinned" for the fork's conventions.  With the author's fork installed, `--real-fairmotion` skips the stand-in for
`conversions` / `quaternion`; regenerating the two fixtures and re-running tests/test_streaming_oracle.py +
tests/test_streaming_gpu.py is then the whole pinning step.

usage: python tests/golden/make_runner_golden.py [--real-fairmotion]
"""
import os
import sys
import types

import numpy as np
import torch
from scipy.spatial.transform import Rotation

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"


# Which functions of the (absent) fairmotion fork are stood in, by what, and whether the accelerated path (SURVEY.md section 8
# rows a12 / a13) reaches them.  check_stand_ins_cover_the_reference() compares this table with what the reference's two files
# on the path actually call.
FAIRMOTION_STAND_INS = {
    # name                      (stand-in,                                             on the path?)
    "conversions.R2A": ("scipy Rotation.from_matrix(R).as_rotvec()", "yes: real_time_runner_minimal.py:161, data_utils.py:177 (6D -> axis-angle)"),
    "conversions.A2R": ("scipy Rotation.from_rotvec(a).as_matrix()", "yes: data_utils.py:185 (axis-angle -> 6D history row)"),
    "conversions.A2Q": ("scipy Rotation.from_rotvec(a).as_quat()   [xyzw]", "no: data_utils.py:274-279 (PyBullet state), :318 (metrics)"),
    "conversions.Q2A": ("scipy Rotation.from_quat(q).as_rotvec()", "no: data generation / metrics"),
    "conversions.Q2R": ("scipy Rotation.from_quat(q).as_matrix()", "no: SBP labels, root correction, IK"),
    "conversions.R2Q": ("scipy Rotation.from_matrix(R).as_quat()", "no: not called by the two files"),
    "quaternion.Q_mult": ("Hamilton product, xyzw", "no: data_utils.py:400 (root correction)"),
}
NOT_STOOD_IN = {"conversions.T2Qp": "data generation only (data_utils.py:132-146)", "quaternion.Q_diff": "metrics only (data_utils.py:318,347)"}


def check_stand_ins_cover_the_reference():
    """Every `conversions.X` / `quaternion.X` the reference's runner and data_utils call is either stood in or listed as off-path."""
    import re
    used = set()
    for f in ("real_time_runner_minimal.py", "data_utils.py"):
        for ln in open(os.path.join(REF, f)):
            code = ln.split("#", 1)[0]
            used |= {"conversions." + m for m in re.findall(r"conversions\.(\w+)", code)}
            used |= {"quaternion." + m for m in re.findall(r"quaternion\.(\w+)", code)}
            if re.search(r"\bQ_mult\(", code):
                used.add("quaternion.Q_mult")
    missing = used - set(FAIRMOTION_STAND_INS) - set(NOT_STOOD_IN)
    assert not missing, f"the reference calls {sorted(missing)}: add a stand-in (or list it as off-path)"
    on_path = {k for k, (_, where) in FAIRMOTION_STAND_INS.items() if where.startswith("yes")}
    assert on_path == {"conversions.R2A", "conversions.A2R"}, on_path
    return used


def rotation_branch_fixture(conv_impl: str):
    """The reference's 6D <-> axis-angle converters at the branch points of the convention (see the module docstring)."""
    import data_utils          # the reference's, with whatever `fairmotion.ops.conversions` is installed (stand-in or the fork)
    rng = np.random.RandomState(2026)
    angles = [0.0, 1e-8, 1e-4, 0.999e-3, 1.001e-3, 0.5, np.pi / 2, np.pi - 1e-2, np.pi - 1e-3, np.pi - 1e-6, np.pi]
    axes = rng.randn(6, 3)
    axes /= np.linalg.norm(axes, axis=1, keepdims=True)
    axes = np.concatenate([axes, np.eye(3)])                    # coordinate axes too (exact zeros in R)
    aa_in = np.array([s * ang * ax for ang in angles for ax in axes for s in (1.0, -1.0)])     # both hemispheres
    aa_long = np.array([s * ang * ax for ang in (np.pi + 1e-3, 1.5 * np.pi, 2 * np.pi - 1e-3) for ax in axes[:3] for s in (1.0, -1.0)])
    aa_all = np.concatenate([aa_in, aa_long])
    n = aa_all.shape[0]
    pad = (-n) % 18                                             # the converters work on whole poses of 18 joints
    aa_all = np.concatenate([aa_all, np.zeros((pad, 3))])
    poses = np.concatenate([aa_all.reshape(-1, 54), np.zeros((aa_all.shape[0] // 18, 3))], axis=1)                 # (b, 57): + the root xyz slot
    six = data_utils.batch_to_rot_mat_2axis(poses)[:, :108]                                                         # A2R
    back = data_utils.batch_rot_mat_2axis_to_aa(six)                                                                # R2A
    # and R2A on 6D rows that are NOT exactly orthonormal (what the network emits): scaled columns, a slightly skewed second axis
    six_noisy = six.reshape(-1, 3, 2) * np.array([1.7, 0.4]) + 0.02 * rng.randn(aa_all.shape[0], 3, 2)
    back_noisy = data_utils.batch_rot_mat_2axis_to_aa(six_noisy.reshape(-1, 108))
    out = {"aa_in": aa_all, "six_from_aa": six.reshape(-1, 6), "aa_from_six": back.reshape(-1, 3),
           "six_noisy": six_noisy.reshape(-1, 6), "aa_from_six_noisy": back_noisy.reshape(-1, 3), "n_real": np.array([n]),
           "conversions_impl": np.frombuffer(conv_impl.encode(), dtype=np.uint8).copy()}
    path = os.path.join(HERE, "tip_rotation_branches.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", n, "rotations; conversions:", conv_impl)


def install_stubs(real_fairmotion: bool = False):
    if real_fairmotion:                                          # only what is still missing: the simulator
        import fairmotion  # noqa: F401  (must be the author's fork, README.md:37-42)
        sys.modules["pybullet"] = types.ModuleType("pybullet")
        ba = types.ModuleType("bullet_agent")
        ba.SimAgent = type("SimAgent", (), {})
        sys.modules["bullet_agent"] = ba
        return
    conv = types.ModuleType("fairmotion.ops.conversions")
    conv.A2R = lambda a: Rotation.from_rotvec(np.asarray(a)).as_matrix()
    conv.R2A = lambda r: Rotation.from_matrix(np.asarray(r)).as_rotvec()
    conv.A2Q = lambda a: Rotation.from_rotvec(np.asarray(a)).as_quat()
    conv.Q2A = lambda q: Rotation.from_quat(np.asarray(q)).as_rotvec()
    conv.Q2R = lambda q: Rotation.from_quat(np.asarray(q)).as_matrix()
    conv.R2Q = lambda r: Rotation.from_matrix(np.asarray(r)).as_quat()
    quat = types.ModuleType("fairmotion.ops.quaternion")

    def q_mult(q1, q2):  # Hamilton product, xyzw
        x1, y1, z1, w1 = q1
        x2, y2, z2, w2 = q2
        return np.array([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                         w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2])

    quat.Q_mult = q_mult
    ops = types.ModuleType("fairmotion.ops")
    ops.conversions, ops.quaternion = conv, quat
    ops.math = types.ModuleType("fairmotion.ops.math")
    core = types.ModuleType("fairmotion.core")
    motion = types.ModuleType("fairmotion.core.motion")
    motion.Motion = type("Motion", (), {})
    core.motion = motion
    fm = types.ModuleType("fairmotion")
    fm.ops, fm.core = ops, core
    for name, mod in {"fairmotion": fm, "fairmotion.ops": ops, "fairmotion.ops.conversions": conv,
                      "fairmotion.ops.quaternion": quat, "fairmotion.ops.math": ops.math, "fairmotion.core": core,
                      "fairmotion.core.motion": motion}.items():
        sys.modules[name] = mod
    sys.modules["pybullet"] = types.ModuleType("pybullet")
    ba = types.ModuleType("bullet_agent")
    ba.SimAgent = type("SimAgent", (), {})
    sys.modules["bullet_agent"] = ba


class FakeChar:
    """Kinematic stand-in for bullet_agent.SimAgent: same accessors, fixed link transforms."""

    def __init__(self, info):
        self._info = info
        self._joint_indices = range(19)
        self.non_root_active_idx = [j for j in range(19) if j not in (info.lwrist, info.rwrist)]
        self._root_p = np.zeros(3)
        self._root_q = np.array([0.0, 0, 0, 1])

    def get_char_info(self):
        return self._info

    def set_root_pQvw(self, p, Q, v, w):
        self._root_p, self._root_q = np.array(p, dtype=float), np.array(Q, dtype=float)

    def set_joints_pv(self, idx, pos, vel):
        pass

    def get_root_pQ(self):
        return self._root_p, self._root_q

    def get_link_pQ(self, indices):
        return [self._root_p + 0.05 * (i + 1) for i in indices], [np.array([0.0, 0, 0, 1]) for _ in indices]

    def get_link_pQ_joint_frame(self, indices):
        return self.get_link_pQ(indices)


def smooth_imu_sequence(n_frames, seed):
    """Raw IMU frames (72,) = 6 global rotations (row-major 3x3) + 6 global accelerations, smooth in time."""
    rng = np.random.RandomState(seed)
    base = Rotation.random(6, random_state=seed)
    w = rng.randn(6, 3) * 0.6
    out = np.zeros((n_frames, 72))
    acc = rng.randn(6, 3)
    for t in range(n_frames):
        R = (Rotation.from_rotvec(w * (t / 60.0)) * base).as_matrix()
        acc = 0.9 * acc + 0.6 * rng.randn(6, 3)
        out[t, :54] = R.reshape(-1)
        out[t, 54:] = acc.reshape(-1)
    return out


def main():
    real = "--real-fairmotion" in sys.argv
    install_stubs(real)
    sys.path.insert(0, REF)
    print("fairmotion functions the reference's runner / data_utils call:", sorted(check_stand_ins_cover_the_reference()))
    impl = "fairmotion (installed package)" if real else "scipy stand-in (fairmotion fork absent: parity UNPINNED for A2R / R2A conventions)"
    import tip_amd  # noqa: F401
    from tip_amd import synth
    import amass_char_info
    from simple_transformer_with_state import TF_RNN_Past_State
    from real_time_runner_minimal import RTRunnerMin

    torch.Tensor.cuda = lambda self, *a, **k: self      # harness only
    cfg = synth.PAPER
    w = synth.make_weights(cfg, seed=0)
    model = TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                              dropout=0.0, in_dropout=0.0, past_state_dropout=0.0, with_acc_sum=True)
    model.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
    model.eval()

    calls = []
    orig_forward = model.forward

    def tapped(x_imu, x_s):
        y = orig_forward(x_imu, x_s)
        calls.append((x_imu.numpy().copy(), x_s.numpy().copy(), y.detach().numpy().copy()))
        return y

    model.forward = tapped

    n_frames = 70
    out = {}
    for sid, seed in enumerate((3, 11)):
        rng = np.random.RandomState(100 + seed)
        s_init = np.zeros(114)
        s_init[3:57] = rng.randn(54) * 0.3          # root + 17 joint axis-angles
        s_init[2] = 0.95
        runner = RTRunnerMin(FakeChar(amass_char_info), model, 40, s_init, with_acc_sum=True)
        raw = smooth_imu_sequence(n_frames, seed)
        calls.clear()
        qdq, ct, hist = [], [], []
        root = np.array([0.0, 0.0, 0.95])
        for t in range(n_frames):
            res = runner.step(raw[t], root)
            root = res["qdq"][:3]
            qdq.append(res["qdq"].copy())
            ct.append(res["ct"].copy())
            hist.append(np.array(runner.s_and_c_in_buffer[-1]).copy())
        tag = f"stream{sid}"
        out[tag + "/raw_imu"] = raw
        out[tag + "/s_init"] = s_init
        out[tag + "/qdq"] = np.array(qdq)
        out[tag + "/ct"] = np.array(ct)
        out[tag + "/hist_last"] = np.array(hist)               # history row appended after each frame (131,)
        out[tag + "/n_calls"] = np.array([len(calls)])
        out[tag + "/call_T"] = np.array([c[0].shape[1] for c in calls])
        # model inputs of a few calls in full, of every call's newest row, and the consumed output rows
        for k in (0, 1, 10, 39, 40, len(calls) - 1):
            out[f"{tag}/x_imu_call{k}"] = calls[k][0][0]
            out[f"{tag}/x_s_call{k}"] = calls[k][1][0]
        out[tag + "/x_imu_last_rows"] = np.array([c[0][0, -1] for c in calls])
        out[tag + "/x_s_last_rows"] = np.array([c[1][0, -1] for c in calls])
        out[tag + "/y_last_rows"] = np.array([c[2][0, -1] for c in calls])
        print(tag, "frames", n_frames, "model calls", len(calls), "T:", out[tag + "/call_T"][:8], "...")
    out["conversions_impl"] = np.frombuffer(impl.encode(), dtype=np.uint8).copy()
    path = os.path.join(HERE, "tip_runner_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")
    rotation_branch_fixture(impl)


if __name__ == "__main__":
    main()

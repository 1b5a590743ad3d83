"""Deterministic synthetic weights and IMU windows (no torch RNG, no numpy Generator).

Everything is derived from a counter-based splitmix64 hash so that this container (where
the golden vectors are captured from the reference) and the GPU box (where only this
repo exists) regenerate bit-identical tensors without shipping 14.7 MB of weights.

Value layouts follow the reference's tensor contract:
  * x_imu[..., 0:54]   six 3x3 rotations, row-major; root global, the other five in the root frame
                       (real_time_runner_minimal.py:132, data_utils.py:190-219)
  * x_imu[..., 54:72]  six accelerations (root global, five root-local)
  * x_imu[..., 72:90]  running sum of the local acc over <=40 frames, / 15
                       (real_time_runner_minimal.py:134-141, constants.py:17-18)
  * x_s[..., 0:108]    18 joints x first two columns of R, (3x2) row-major (data_utils.py:182-187)
  * x_s[..., 108:111]  root velocity history (zeroed inside forward, simple_transformer_with_state.py:75)
  * x_s[..., 111:131]  5 SBPs x (flag, xyz offset)
State-dict key order / shapes: simple_transformer_with_state.py:9-54 (56 tensors for the paper config).
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _stream_key(seed: int, stream: str) -> np.uint64:
    k = np.uint64(seed & 0xFFFFFFFFFFFFFFFF)
    for ch in stream.encode():
        k = _splitmix(np.asarray(k ^ np.uint64(ch), dtype=np.uint64))[()]
    return np.uint64(k)


def uniform01(seed: int, stream: str, n: int) -> np.ndarray:
    """n doubles in [0,1), a pure function of (seed, stream, index)."""
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        h = _splitmix(_splitmix(idx ^ _stream_key(seed, stream)))
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normal(seed: int, stream: str, n: int) -> np.ndarray:
    u1 = uniform01(seed, stream + "/a", n)
    u2 = uniform01(seed, stream + "/b", n)
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * math.pi * u2)


def _random_rotations(seed: int, stream: str, n: int) -> np.ndarray:
    """n rotation matrices uniform on SO(3) via normalised gaussian quaternions -> [n,3,3]."""
    q = normal(seed, stream, 4 * n).reshape(n, 4)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((n, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - z * w)
    R[:, 0, 2] = 2 * (x * z + y * w)
    R[:, 1, 0] = 2 * (x * y + z * w)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - x * w)
    R[:, 2, 0] = 2 * (x * z - y * w)
    R[:, 2, 1] = 2 * (y * z + x * w)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


# ----------------------------------------------------------------------------------------------
# model configuration + state-dict layout
# ----------------------------------------------------------------------------------------------

PAPER = dict(input_size_imu=72, size_s=131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256,
             n_heads=16, tf_layers=4, with_rnn=True, with_acc_sum=True)
# BASELINE.json configs[4]; H / R are not stated there, SURVEY.md section 8d assumes H=16, R=512.
SCALED = dict(input_size_imu=72, size_s=131, rnn_hid_size=512, tf_hid_size=4096, tf_in_dim=1024,
              n_heads=16, tf_layers=12, with_rnn=True, with_acc_sum=True)
# small config used by fast parity tests (exercises T=80, dh=32, non-paper widths)
TINY = dict(input_size_imu=72, size_s=131, rnn_hid_size=192, tf_hid_size=320, tf_in_dim=128,
            n_heads=4, tf_layers=2, with_rnn=True, with_acc_sum=True)


def state_dict_layout(cfg: dict) -> "OrderedDict[str, tuple]":
    """Key -> shape, in the reference's state_dict() order (simple_transformer_with_state.py:20-46)."""
    n_in = cfg["input_size_imu"] + cfg["size_s"] + (18 if cfg.get("with_acc_sum", False) else 0)
    D, F, R, S = cfg["tf_in_dim"], cfg["tf_hid_size"], cfg["rnn_hid_size"], cfg["size_s"]
    lay: "OrderedDict[str, tuple]" = OrderedDict()
    lay["in_linear.weight"] = (D, n_in)
    lay["in_linear.bias"] = (D,)
    for l in range(cfg["tf_layers"]):
        p = f"tf_encode.layers.{l}."
        lay[p + "self_attn.in_proj_weight"] = (3 * D, D)
        lay[p + "self_attn.in_proj_bias"] = (3 * D,)
        lay[p + "self_attn.out_proj.weight"] = (D, D)
        lay[p + "self_attn.out_proj.bias"] = (D,)
        lay[p + "linear1.weight"] = (F, D)
        lay[p + "linear1.bias"] = (F,)
        lay[p + "linear2.weight"] = (D, F)
        lay[p + "linear2.bias"] = (D,)
        lay[p + "norm1.weight"] = (D,)
        lay[p + "norm1.bias"] = (D,)
        lay[p + "norm2.weight"] = (D,)
        lay[p + "norm2.bias"] = (D,)
    if cfg.get("with_rnn", True):
        lay["rnn.weight_ih_l0"] = (R, D)
        lay["rnn.weight_hh_l0"] = (R, R)
        lay["rnn.bias_ih_l0"] = (R,)
        lay["rnn.bias_hh_l0"] = (R,)
        lay["linear.weight"] = (S, R)
    else:
        lay["linear.weight"] = (S, D)
    lay["linear.bias"] = (S,)
    return lay


def make_weights(cfg: dict, seed: int = 0, gain: float = 1.0, ln_gamma: float = 1.0) -> "OrderedDict[str, np.ndarray]":
    """fp32 weights: U(-g/sqrt(fan_in), g/sqrt(fan_in)) matrices and biases; LayerNorm gamma = ln_gamma (1 + 0.1 U(-1,1)),
    beta = 0.1 U(-1,1) so the affine part is exercised.  `gain` > 1 and `ln_gamma` > 1 leave the benign random-init regime
    (larger attention logits, saturating tanh recurrence): the conditioning sweep of tests/golden/make_golden.py."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in state_dict_layout(cfg).items():
        n = int(np.prod(shape))
        u = uniform01(seed, "w/" + name, n) * 2.0 - 1.0
        if ".norm" in name:
            v = ln_gamma * (1.0 + 0.1 * u) if name.endswith("weight") else 0.1 * u
        else:
            if len(shape) == 2:
                fan_in = shape[1]
            elif name.startswith("rnn."):
                fan_in = cfg["rnn_hid_size"]
            elif name == "in_linear.bias":
                fan_in = cfg["input_size_imu"] + cfg["size_s"] + (18 if cfg.get("with_acc_sum") else 0)
            elif name.endswith("linear1.bias") or name.endswith("in_proj_bias") or name.endswith("out_proj.bias"):
                fan_in = cfg["tf_in_dim"]
            elif name.endswith("linear2.bias"):
                fan_in = cfg["tf_hid_size"]
            else:  # linear.bias
                fan_in = cfg["rnn_hid_size"] if cfg.get("with_rnn", True) else cfg["tf_in_dim"]
            v = u * (gain / math.sqrt(fan_in))
        out[name] = v.astype(np.float32).reshape(shape)
    return out


def make_inputs(cfg: dict, B: int, T: int, seed: int = 1234, nan_frac: float = 0.05):
    """Synthetic IMU windows + past-state history -> (x_imu [B,T,n_imu(+18)], x_s [B,T,size_s]) fp32.

    Distributions follow SURVEY.md section 8d.  NaNs (DIP-style missing labels) are injected into the
    root-velocity columns 108:111 and a few pose columns so the scrub at
    simple_transformer_with_state.py:65 is exercised."""
    n_imu, S = cfg["input_size_imu"], cfg["size_s"]
    N = B * T
    n_sens = 6
    R = _random_rotations(seed, "imu/R", N * n_sens).reshape(N, n_sens, 3, 3)
    Rroot_T = np.transpose(R[:, 0], (0, 2, 1))
    Rloc = R.copy()
    Rloc[:, 1:] = np.einsum("nij,nsjk->nsik", Rroot_T, R[:, 1:])
    acc = np.clip(normal(seed, "imu/acc", N * n_sens * 3) * 2.0, -10.0, 10.0).reshape(N, n_sens, 3)
    acc_loc = acc.copy()
    acc_loc[:, 1:] = np.einsum("nij,nsj->nsi", Rroot_T, acc[:, 1:])
    x_imu = np.concatenate([Rloc.reshape(N, 54), acc_loc.reshape(N, 18)], axis=1)
    assert n_imu == 72, "synthetic generator is written for the 6-IMU (72-wide) layout"
    if cfg.get("with_acc_sum", False):
        a = acc_loc.reshape(B, T, 18)
        csum = np.cumsum(a, axis=1)
        if T > 40:  # window of <=40 frames (constants.py:17)
            csum[:, 40:] -= csum[:, :-40]
        x_imu = np.concatenate([x_imu, (csum / 15.0).reshape(N, 18)], axis=1)
    x_imu = x_imu.reshape(B, T, -1)

    n_j = 18
    Rj = _random_rotations(seed, "s/R", N * n_j)
    pose6 = Rj[:, :, :2].reshape(N, n_j * 6)            # (3x2) row-major, first two columns
    rootv = normal(seed, "s/v", N * 3).reshape(N, 3) * 0.5
    n_sbp = (S - 111) // 4
    flags = (uniform01(seed, "s/flag", N * n_sbp) < 0.4).astype(np.float64).reshape(N, n_sbp, 1)
    offs = (uniform01(seed, "s/off", N * n_sbp * 3) * 0.5 - 0.25).reshape(N, n_sbp, 3)
    sbp = np.concatenate([flags, offs], axis=2).reshape(N, n_sbp * 4)
    x_s = np.concatenate([pose6, rootv, sbp], axis=1)
    assert x_s.shape[1] == S
    if nan_frac > 0:
        m = uniform01(seed, "s/nan", N * 3).reshape(N, 3) < nan_frac
        x_s[:, 108:111][m] = np.nan
        m2 = uniform01(seed, "s/nan2", N * S).reshape(N, S) < nan_frac * 0.02
        x_s[m2] = np.nan
    return x_imu.astype(np.float32), x_s.reshape(B, T, S).astype(np.float32)


def make_keep_mask(cfg: dict, B: int, T: int, p: float, seed: int = 7) -> np.ndarray:
    """Bernoulli keep-mask for the past-state dropout (simple_transformer_with_state.py:77)."""
    S = cfg["size_s"]
    return (uniform01(seed, "mask", B * T * S) >= p).astype(np.float32).reshape(B, T, S)


def flops_per_window(cfg: dict, T: int) -> float:
    """Algorithmic FLOPs of one window forward."""
    In = cfg["input_size_imu"] + cfg["size_s"] + (18 if cfg.get("with_acc_sum") else 0)
    D, F, L, R, S = cfg["tf_in_dim"], cfg["tf_hid_size"], cfg["tf_layers"], cfg["rnn_hid_size"], cfg["size_s"]
    f = 2.0 * T * In * D + L * (2.0 * T * D * 3 * D + 4.0 * T * T * D + 2.0 * T * D * D + 4.0 * T * D * F)
    if cfg.get("with_rnn", True):
        f += 2.0 * T * D * R + 2.0 * T * R * R + 2.0 * T * R * S
    else:
        f += 2.0 * T * D * S
    return f

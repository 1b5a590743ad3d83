"""TEST INFRASTRUCTURE — ctypes front-end of the C oracle (oracle/tip_oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product path (transformer-inertial-poser_amd/) never does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libtip_oracle.so")
_lib = None


class _Cfg(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in (
        "n_imu_total", "n_state", "d_model", "n_heads", "d_ff", "n_layers", "d_rnn", "with_rnn",
        "rootv_begin", "rootv_end")]


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("tip_oracle.c", "tip_oracle_impl.h")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtip_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
        for suf in ("f32", "f64"):
            fn = getattr(_lib, "tip_oracle_forward_" + suf)
            fn.restype = ctypes.c_int
        _lib.tip_oracle_max_threads.restype = ctypes.c_int
    return _lib


def max_threads() -> int:
    return int(_load().tip_oracle_max_threads())


def _cfg_struct(cfg: dict) -> _Cfg:
    n_imu_total = cfg["input_size_imu"] + (18 if cfg.get("with_acc_sum", False) else 0)
    return _Cfg(n_imu_total, cfg["size_s"], cfg["tf_in_dim"], cfg["n_heads"], cfg["tf_hid_size"],
                cfg["tf_layers"], cfg["rnn_hid_size"], 1 if cfg.get("with_rnn", True) else 0, 18 * 6, 18 * 6 + 3)


def forward(cfg: dict, weights: Dict[str, np.ndarray], x_imu: np.ndarray, x_s: np.ndarray,
            keep_mask: Optional[np.ndarray] = None, keep_scale: float = 1.0, dtype=np.float32,
            nthreads: int = 0, taps: bool = False):
    """Run the oracle.  weights: state-dict-ordered mapping name -> array.  Returns y [B,T,S]
    (and a dict of taps when taps=True)."""
    lib = _load()
    dt = np.dtype(dtype)
    assert dt in (np.dtype(np.float32), np.dtype(np.float64))
    suf = "f32" if dt == np.float32 else "f64"
    cptr = ctypes.POINTER(ctypes.c_float if suf == "f32" else ctypes.c_double)
    creal = ctypes.c_float if suf == "f32" else ctypes.c_double
    c = _cfg_struct(cfg)
    ws = [np.ascontiguousarray(v, dtype=dt) for v in weights.values()]
    arr = (cptr * len(ws))(*[w.ctypes.data_as(cptr) for w in ws])
    xi = np.ascontiguousarray(x_imu, dtype=dt)
    xs = np.ascontiguousarray(x_s, dtype=dt)
    B, T = xi.shape[0], xi.shape[1]
    assert xi.shape[2] == c.n_imu_total and xs.shape == (B, T, c.n_state)
    y = np.empty((B, T, c.n_state), dtype=dt)
    km = None if keep_mask is None else np.ascontiguousarray(keep_mask, dtype=dt)
    D, L, R = c.d_model, c.n_layers, c.d_rnn
    t_in = np.empty((B, T, D), dtype=dt) if taps else None
    t_l = np.empty((L, B, T, D), dtype=dt) if taps else None
    t_r = np.empty((B, T, R), dtype=dt) if (taps and c.with_rnn) else None

    def p(a):
        return a.ctypes.data_as(cptr) if a is not None else None

    if nthreads <= 0:
        nthreads = max_threads()
    rc = getattr(lib, "tip_oracle_forward_" + suf)(
        ctypes.byref(c), arr, p(xi), p(xs), p(y), ctypes.c_int(B), ctypes.c_int(T), p(km), creal(keep_scale),
        p(t_in), p(t_l), p(t_r), ctypes.c_int(nthreads))
    if rc != 0:
        raise RuntimeError(f"tip_oracle_forward_{suf} failed rc={rc}")
    if taps:
        return y, {"in": t_in, "layers": t_l, "rnn": t_r}
    return y

"""ctypes binding of libtip_hip.so (include/tip_hip.h).  No fallback: if the library is missing or fails to
load, everything that needs it raises TipLibraryError."""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import List, Optional, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# TIP_LIB=measure (a PYTHON-side switch, for the scripts under tools/): load the measurement build, in which the launchers' TIP_*
# environment switches are alive (csrc/Makefile: `make measure`); the default library ignores the environment.
MEASURE = os.environ.get("TIP_LIB", "") == "measure"
LIB_PATH = os.path.join(CSRC, "libtip_hip_measure.so" if MEASURE else "libtip_hip.so")
if os.environ.get("TIP_LIB", "").endswith(".so"):   # (tools/: an A/B build of the library by path, e.g. gpurun_ab/libtip_x.so)
    LIB_PATH = os.path.abspath(os.environ["TIP_LIB"])

TIP_FWD_LAST_ROW_ONLY = 0x1
TIP_FWD_KEEP_MASK = 0x2
TIP_PLAN_AUTO, TIP_PLAN_GENERAL, TIP_PLAN_FUSED, TIP_PLAN_LATENCY, TIP_PLAN_FUSED2, TIP_PLAN_FUSEDH = 0, 1, 2, 3, 4, 6   # 5, 7, 8, 9: retired
TIP_PLAN_FUSED1S = 10   # one window on two co-resident workgroups (64 < B <= 128)
TIP_SAVED_QKV, TIP_SAVED_ATT, TIP_SAVED_X1, TIP_SAVED_HID, TIP_SAVED_XOUT, TIP_SAVED_HALL = range(6)
TIP_STREAM_FRAME_AUTO = -1   # tip_stream_ingest / tip_stream_consume: continue from the counter in the state buffer (HIP graphs)
TIP_OPT_PLAN, TIP_OPT_PROFILE, TIP_OPT_RNN_CLUSTER, TIP_OPT_FAULT_INJECT, TIP_OPT_FUSE_HEAD = 1, 2, 3, 4, 5
TIP_OPT_AUTO_DEMOTE, TIP_OPT_DEMOTED, TIP_OPT_F1S_PARTS = 7, 8, 9   # 6: retired
TIP_OPT_NO_FLOW, TIP_OPT_HANDOFF_KIND = 10, 11   # few-stream plan without its one-launch form; (read only) 0 none / 1 a wait gave up / 2 one-launch placement only
TIP_ABI_VERSION = 5
TIP_RNN_CLUSTER_ROWS4 = 0x44   # TIP_OPT_RNN_CLUSTER value: 4-window tiles on 4-workgroup clusters (AUTO's choice for rnn_hidden 512)
TIP_ERR_HANDOFF = -8
TIP_ERR_UNSUPPORTED_CONFIG = -2
TIP_LOSS_Q, TIP_LOSS_C, TIP_LOSS_J, TIP_LOSS_STATS = 1, 2, 4, 16

# every symbol include/tip_hip.h declares (tests check that the .so exports exactly these + the hooks of tip_hip_debug.h)
EXPORTS = (
    "tip_abi_version", "tip_create", "tip_destroy", "tip_strerror", "tip_last_hip_error", "tip_set_option",
    "tip_get_option", "tip_num_tensors", "tip_tensor_info", "tip_packed_bytes", "tip_pack_weights",
    "tip_pack_weights_device", "tip_attach_packed", "tip_workspace_bytes", "tip_max_batch", "tip_forward", "tip_forward_dropout", "tip_draw_keep_mask", "tip_forward_f64_bytes", "tip_forward_f64", "tip_forward_count", "tip_profile_read",
    "tip_spin_timeouts", "tip_check", "tip_stream_state_bytes", "tip_stream_reset", "tip_stream_window_len", "tip_stream_ingest", "tip_stream_consume",
    "tip_reuse_cache_bytes", "tip_reuse_reset", "tip_forward_reuse", "tip_stream_frame_counter_offset", "tip_stream_ingest_newest",
    "tip_train_bytes", "tip_train_saved_view", "tip_train_forward", "tip_train_backward", "tip_train_input_grads",
    "tip_train_bytes_f64", "tip_train_forward_f64", "tip_train_backward_f64",
    "tip_combine_frames", "tip_combine_scratch_bytes", "tip_combine_sequence", "tip_gather_windows",
    "tip_loss_ws_bytes", "tip_loss_forward", "tip_loss_backward", "tip_loss_forward_f64", "tip_loss_backward_f64",
)


class TipLibraryError(RuntimeError):
    pass


class TipStatusError(RuntimeError):
    def __init__(self, status: int, text: str):
        super().__init__(f"libtip_hip: {text} (status {status})")
        self.status = status


class TipHandoffError(TipStatusError):
    """An inter-workgroup hand-off wait of an EARLIER launch gave up (a co-tenant held CUs, or a partner workgroup never
    became resident): the outputs of that launch are NaN-poisoned.  Sticky until Handle.check(clear=True)."""


class TipConfig(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in (
        "input_size_imu", "size_s", "rnn_hid_size", "tf_hid_size", "tf_in_dim", "n_heads", "tf_layers",
        "with_rnn", "with_acc_sum", "t_max")]


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile csrc/*.hip for gfx950 into csrc/libtip_hip.so (hipcc cross-compiles without a GPU)."""
    if os.path.dirname(LIB_PATH) != CSRC:      # TIP_LIB=<path>.so: an A/B build somebody made by hand — nothing here knows its recipe
        return LIB_PATH
    cmd = ["make", "-C", CSRC, "-j8"] + (["-B"] if force else []) + [os.path.basename(LIB_PATH)]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise TipLibraryError("building libtip_hip.so failed:\n" + res.stdout[-4000:])
    return LIB_PATH


def build_measure(verbose: bool = False) -> str:
    """Compile the measurement build (csrc/libtip_hip_measure.so: TIP_* environment switches of the launchers alive)."""
    cmd = ["make", "-C", CSRC, "-j8", "libtip_hip_measure.so"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        raise TipLibraryError("building libtip_hip_measure.so failed:\n" + res.stdout[-4000:])
    if verbose:
        print(res.stdout)
    return os.path.join(CSRC, "libtip_hip_measure.so")


_lib = None


def load() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7; importing it first makes libtip_hip.so bind to that same HIP runtime
    # (two runtimes in one process do not share devices, streams or allocations).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise TipLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C transformer-inertial-poser_amd/csrc`).  There is no non-HIP fallback for the forward path.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise TipLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    vp, i32, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    lib.tip_abi_version.restype = i32
    lib.tip_create.argtypes = [ctypes.POINTER(TipConfig), ctypes.POINTER(vp)]
    lib.tip_destroy.argtypes = [vp]
    lib.tip_destroy.restype = None
    lib.tip_strerror.argtypes = [i32]
    lib.tip_strerror.restype = ctypes.c_char_p
    lib.tip_last_hip_error.argtypes = [vp]
    lib.tip_last_hip_error.restype = ctypes.c_char_p
    lib.tip_set_option.argtypes = [vp, i32, i32]
    lib.tip_get_option.argtypes = [vp, i32, ctypes.POINTER(i32)]
    lib.tip_num_tensors.argtypes = [vp]
    lib.tip_tensor_info.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(i32), ctypes.POINTER(i32)]
    lib.tip_packed_bytes.argtypes = [vp, ctypes.POINTER(sz)]
    lib.tip_pack_weights.argtypes = [vp, ctypes.POINTER(vp), i32, vp, sz]
    lib.tip_pack_weights_device.argtypes = [vp, ctypes.POINTER(vp), i32, vp, sz, vp]
    lib.tip_attach_packed.argtypes = [vp, vp, sz]
    lib.tip_workspace_bytes.argtypes = [vp, i32, i32, ctypes.POINTER(sz)]
    lib.tip_max_batch.argtypes = [vp, i32, i32, ctypes.POINTER(i32)]
    lib.tip_forward.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, ctypes.c_float, vp, sz, vp]
    lib.tip_forward_dropout.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, ctypes.c_float, ctypes.c_float, ctypes.c_uint64, ctypes.c_float,
                                        ctypes.c_uint64, vp, sz, vp]
    lib.tip_draw_keep_mask.argtypes = [ctypes.c_float, ctypes.c_uint64, vp, sz, vp]
    lib.tip_train_input_grads.argtypes = [vp, ctypes.POINTER(vp), i32, vp, vp, ctypes.c_float, vp, sz, vp, vp, i32, i32, vp]
    lib.tip_forward_f64_bytes.argtypes = [vp, i32, i32, ctypes.POINTER(sz)]
    lib.tip_forward_f64.argtypes = [vp, ctypes.POINTER(vp), i32, vp, vp, vp, i32, i32, i32, vp, ctypes.c_double, vp, sz, vp]
    lib.tip_forward_count.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    lib.tip_profile_read.argtypes = [vp, ctypes.POINTER(ctypes.c_char_p), ctypes.POINTER(ctypes.c_float),
                                     ctypes.POINTER(i32), i32]
    lib.tip_spin_timeouts.argtypes = [ctypes.POINTER(ctypes.c_uint)]
    lib.tip_check.argtypes = [vp, i32]
    lib.tip_stream_state_bytes.argtypes = [i32, ctypes.POINTER(sz)]
    lib.tip_stream_reset.argtypes = [vp, vp, i32, vp]
    lib.tip_stream_window_len.argtypes = [i32]
    lib.tip_stream_ingest.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    lib.tip_stream_consume.argtypes = [vp, vp, i32, i32, vp, vp, vp]
    lib.tip_stream_ingest_newest.argtypes = lib.tip_stream_ingest.argtypes
    lib.tip_reuse_cache_bytes.argtypes = [vp, i32, ctypes.POINTER(sz)]
    lib.tip_reuse_reset.argtypes = [vp, sz, vp]
    lib.tip_forward_reuse.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp, sz, i32, vp, vp, sz, vp]
    lib.tip_stream_frame_counter_offset.argtypes = [ctypes.POINTER(sz)]
    u64 = ctypes.c_ulonglong
    f32 = ctypes.c_float
    lib.tip_train_bytes.argtypes = [vp, i32, i32, ctypes.POINTER(sz), ctypes.POINTER(sz)]
    lib.tip_train_saved_view.argtypes = [vp, i32, i32, i32, i32, ctypes.POINTER(sz), ctypes.POINTER(sz)]
    lib.tip_train_forward.argtypes = [vp, ctypes.POINTER(vp), i32, vp, vp, vp, f32, f32, u64, vp, vp, sz, i32, i32, vp]
    lib.tip_train_backward.argtypes = [vp, ctypes.POINTER(vp), i32, vp, vp, sz, vp, sz, vp, sz, f32, u64, i32, i32, vp]
    f64t = ctypes.c_double
    lib.tip_train_bytes_f64.argtypes = [vp, i32, i32, ctypes.POINTER(sz), ctypes.POINTER(sz)]
    lib.tip_train_forward_f64.argtypes = [vp, ctypes.POINTER(vp), i32, vp, vp, vp, f64t, f32, u64, vp, vp, sz, i32, i32, vp]
    lib.tip_train_backward_f64.argtypes = [vp, ctypes.POINTER(vp), i32, vp, vp, sz, vp, sz, vp, sz, f32, u64, i32, i32, vp]
    lib.tip_combine_frames.argtypes = [i32, i32]
    lib.tip_combine_scratch_bytes.argtypes = [i32, i32, ctypes.POINTER(sz)]
    lib.tip_combine_sequence.argtypes = [vp, vp, vp, i32, i32, vp, i32, vp, vp, vp, vp, sz, vp]
    lib.tip_gather_windows.argtypes = [vp, vp, vp, ctypes.c_longlong, vp, i32, i32, vp, vp, vp, vp]
    ll = ctypes.c_longlong
    lib.tip_loss_ws_bytes.argtypes = [i32, i32, ctypes.POINTER(sz)]
    lib.tip_loss_forward.argtypes = [vp, ll, vp, ll, i32, i32, i32, i32, i32, i32, vp, vp, sz, vp]
    lib.tip_loss_backward.argtypes = [vp, ll, vp, ll, i32, i32, i32, i32, i32, i32, vp, vp, vp, ll, vp]
    lib.tip_loss_forward_f64.argtypes = lib.tip_loss_forward.argtypes
    lib.tip_loss_backward_f64.argtypes = lib.tip_loss_backward.argtypes
    for name in EXPORTS:
        if name not in ("tip_destroy", "tip_strerror", "tip_last_hip_error"):
            getattr(lib, name).restype = i32
    if lib.tip_abi_version() != TIP_ABI_VERSION:
        raise TipLibraryError(f"libtip_hip.so ABI version {lib.tip_abi_version()} != {TIP_ABI_VERSION}: rebuild it (make -C csrc)")
    _lib = lib
    return lib


class Handle:
    """Owns one tip_handle (one per GPU)."""

    def __init__(self, cfg: TipConfig):
        self.lib = load()
        self._h = ctypes.c_void_p()
        self.cfg = cfg
        self._check(self.lib.tip_create(ctypes.byref(cfg), ctypes.byref(self._h)))

    def _check(self, status: int) -> int:
        if status < 0:
            text = self.lib.tip_strerror(status).decode()
            if status == -5 and self._h:
                text += " — " + self.lib.tip_last_hip_error(self._h).decode()
            if status == TIP_ERR_HANDOFF:
                raise TipHandoffError(status, text)
            raise TipStatusError(status, text)
        return status

    def close(self):
        if getattr(self, "_h", None):
            self.lib.tip_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    # -- parameters ---------------------------------------------------------------------------------
    def tensor_table(self) -> List[Tuple[str, Tuple[int, ...]]]:
        out = []
        for i in range(self._check(self.lib.tip_num_tensors(self._h))):
            name = ctypes.c_char_p()
            r, c = ctypes.c_int(), ctypes.c_int()
            self._check(self.lib.tip_tensor_info(self._h, i, ctypes.byref(name), ctypes.byref(r), ctypes.byref(c)))
            out.append((name.value.decode(), (r.value, c.value) if c.value else (r.value,)))
        return out

    def packed_bytes(self) -> int:
        n = ctypes.c_size_t()
        self._check(self.lib.tip_packed_bytes(self._h, ctypes.byref(n)))
        return n.value

    def pack_weights(self, host_ptrs: List[int], out_ptr: int, out_bytes: int):
        arr = (ctypes.c_void_p * len(host_ptrs))(*host_ptrs)
        self._check(self.lib.tip_pack_weights(self._h, arr, len(host_ptrs), ctypes.c_void_p(out_ptr), out_bytes))

    def pack_weights_device(self, dev_ptrs: List[int], out_dev_ptr: int, out_bytes: int, stream: int):
        arr = (ctypes.c_void_p * len(dev_ptrs))(*dev_ptrs)
        self._check(self.lib.tip_pack_weights_device(self._h, arr, len(dev_ptrs), ctypes.c_void_p(out_dev_ptr), out_bytes,
                                                     stream))

    def attach_packed(self, dev_ptr: int, nbytes: int):
        self._check(self.lib.tip_attach_packed(self._h, ctypes.c_void_p(dev_ptr), nbytes))

    # -- forward ------------------------------------------------------------------------------------
    def workspace_bytes(self, B: int, T: int) -> int:
        n = ctypes.c_size_t()
        self._check(self.lib.tip_workspace_bytes(self._h, B, T, ctypes.byref(n)))
        return n.value

    def max_batch(self, T: int, fp64: bool = False) -> int:
        """Largest batch one forward / forward_f64 call serves at window length T (the host chunks beyond it)."""
        n = ctypes.c_int()
        self._check(self.lib.tip_max_batch(self._h, T, 1 if fp64 else 0, ctypes.byref(n)))
        return n.value

    def forward(self, x_imu: int, x_s: int, y: int, B: int, T: int, flags: int, keep_mask: Optional[int],
                keep_scale: float, workspace: int, workspace_bytes: int, stream: int):
        self._check(self.lib.tip_forward(self._h, x_imu, x_s, y, B, T, flags, keep_mask, keep_scale, workspace,
                                         workspace_bytes, stream))

    def reuse_cache_bytes(self, n_streams: int) -> int:
        n = ctypes.c_size_t()
        self._check(self.lib.tip_reuse_cache_bytes(self._h, n_streams, ctypes.byref(n)))
        return n.value

    def forward_reuse(self, x_imu: int, x_s: int, y: int, B: int, T: int, flags: int, cache: int, cache_bytes: int, frame_idx: int,
                      frame_ctr: Optional[int], workspace: int, workspace_bytes: int, stream: int):
        """tip_forward_reuse: exact streaming reuse — the newest row's in_linear / layer-0 QKV rows into the ring,
        full windows on the two-window encoder's ring-reading form."""
        self._check(self.lib.tip_forward_reuse(self._h, x_imu, x_s, y, B, T, flags, cache, cache_bytes, frame_idx, frame_ctr,
                                               workspace, workspace_bytes, stream))

    def train_input_grads(self, param_ptrs, x_s: int, keep_mask: Optional[int], keep_scale: float, scratch: int, scratch_bytes: int,
                          dx_imu: Optional[int], dx_s: Optional[int], B: int, T: int, stream: int):
        """tip_train_input_grads: right after train_backward on the same scratch."""
        arr = (ctypes.c_void_p * len(param_ptrs))(*param_ptrs)
        self._check(self.lib.tip_train_input_grads(self._h, arr, len(param_ptrs), x_s, keep_mask, keep_scale, scratch, scratch_bytes,
                                                   dx_imu, dx_s, B, T, stream))

    def forward_dropout(self, x_imu: int, x_s: int, y: int, B: int, T: int, flags: int, keep_mask: Optional[int], keep_scale: float,
                        p_state: float, state_seed: int, p_drop: float, seed: int, workspace: int, workspace_bytes: int, stream: int):
        """tip_forward_dropout: the few-stream forward with the training step's encoder dropout and no activation stash."""
        self._check(self.lib.tip_forward_dropout(self._h, x_imu, x_s, y, B, T, flags, keep_mask, keep_scale, p_state, state_seed, p_drop,
                                                 seed, workspace, workspace_bytes, stream))

    # -- training step (train_model.py:171-196) -------------------------------------------------------
    def train_bytes(self, B: int, T: int, fp64: bool = False) -> Tuple[int, int]:
        """(saved_bytes, scratch_bytes); raises TipStatusError(-2: unsupported config) when the HIP training path
        does not cover this configuration.  fp64: the step of a module built under --double (tip_train_*_f64)."""
        a, b = ctypes.c_size_t(), ctypes.c_size_t()
        fn = self.lib.tip_train_bytes_f64 if fp64 else self.lib.tip_train_bytes
        self._check(fn(self._h, B, T, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def train_saved_view(self, B: int, T: int, what: int, layer: int) -> Tuple[int, int]:
        """(float offset, float count) of one stashed activation inside the `saved` buffer of train_forward."""
        a, b = ctypes.c_size_t(), ctypes.c_size_t()
        self._check(self.lib.tip_train_saved_view(self._h, B, T, what, layer, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

    def forward_f64_bytes(self, B: int, T: int) -> int:
        n = ctypes.c_size_t()
        self._check(self.lib.tip_forward_f64_bytes(self._h, B, T, ctypes.byref(n)))
        return n.value

    def forward_f64(self, param_ptrs: List[int], x_imu: int, x_s: int, y: int, B: int, T: int, flags: int,
                    keep_mask: Optional[int], keep_scale: float, workspace: int, workspace_bytes: int, stream: int):
        """The module built under train_model.py's --double: fp64 parameters (raw device pointers, state-dict order) and windows."""
        arr = (ctypes.c_void_p * len(param_ptrs))(*param_ptrs)
        self._check(self.lib.tip_forward_f64(self._h, arr, len(param_ptrs), x_imu, x_s, y, B, T, flags, keep_mask, keep_scale,
                                             workspace, workspace_bytes, stream))

    def train_forward(self, param_ptrs: List[int], x_imu: int, x_s: int, keep_mask: Optional[int], keep_scale: float,
                      p_drop: float, seed: int, y: int, saved: int, saved_bytes: int, B: int, T: int, stream: int, fp64: bool = False):
        arr = (ctypes.c_void_p * len(param_ptrs))(*param_ptrs)
        fn = self.lib.tip_train_forward_f64 if fp64 else self.lib.tip_train_forward
        self._check(fn(self._h, arr, len(param_ptrs), x_imu, x_s, keep_mask, keep_scale, p_drop, seed, y, saved, saved_bytes, B, T, stream))

    def train_backward(self, param_ptrs: List[int], dy: int, saved: int, saved_bytes: int, scratch: int,
                       scratch_bytes: int, grads: int, grads_floats: int, p_drop: float, seed: int, B: int, T: int,
                       stream: int, fp64: bool = False):
        arr = (ctypes.c_void_p * len(param_ptrs))(*param_ptrs)
        fn = self.lib.tip_train_backward_f64 if fp64 else self.lib.tip_train_backward
        self._check(fn(self._h, arr, len(param_ptrs), dy, saved, saved_bytes, scratch, scratch_bytes, grads, grads_floats, p_drop, seed,
                       B, T, stream))

    def check(self, clear: bool = False):
        """Raise TipHandoffError if a hand-off wait of a completed launch gave up (no device sync: synchronise the stream first
        for a definitive answer about launches in flight).  clear=True also resets the sticky word."""
        self._check(self.lib.tip_check(self._h, 1 if clear else 0))

    def check_clear(self):
        """Reset the sticky hand-off word without raising."""
        self.lib.tip_check(self._h, 1)

    def forward_count(self) -> int:
        n = ctypes.c_uint64()
        self._check(self.lib.tip_forward_count(self._h, ctypes.byref(n)))
        return n.value

    def set_option(self, opt: int, value: int):
        self._check(self.lib.tip_set_option(self._h, opt, value))

    def get_option(self, opt: int) -> int:
        v = ctypes.c_int()
        self._check(self.lib.tip_get_option(self._h, opt, ctypes.byref(v)))
        return v.value

    def profile_read(self, cap: int = 128):
        names = (ctypes.c_char_p * cap)()
        ms = (ctypes.c_float * cap)()
        launches = (ctypes.c_int * cap)()
        n = self._check(self.lib.tip_profile_read(self._h, names, ms, launches, cap))
        return [(names[i].decode(), float(ms[i]), int(launches[i])) for i in range(n)]


def draw_keep_mask(p_state: float, state_seed: int, mask_ptr: int, n: int, stream: int):
    """tip_draw_keep_mask: the past-state keep decisions of (p_state, state_seed) as n floats (0 / 1) at device pointer mask_ptr."""
    st = load().tip_draw_keep_mask(p_state, state_seed, mask_ptr, n, stream)
    if st < 0:
        raise TipStatusError(st, load().tip_strerror(st).decode())


def spin_timeouts() -> int:
    """Hand-off waits that gave up since the library was loaded (synchronises the device); must be 0."""
    n = ctypes.c_uint()
    st = load().tip_spin_timeouts(ctypes.byref(n))
    if st < 0:
        raise TipStatusError(st, load().tip_strerror(st).decode())
    return int(n.value)

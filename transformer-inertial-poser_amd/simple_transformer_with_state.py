"""Drop-in for the reference module of the same name: `from simple_transformer_with_state import TF_RNN_Past_State`
resolves here through dropin/simple_transformer_with_state.py.

Mirrors the reference's module surface (/root/reference/simple_transformer_with_state.py):
  * constructor signature                                   (:9-17)
  * state_dict() keys / shapes / order — 56 tensors         (:22-46)
  * forward(x_imu [B,T,72(+18)], x_s [B,T,size_s]) -> [B,T,size_s], inputs untouched, NaNs in x_s scrubbed (:60-102)
  * nn.Module services the callers use: .cuda(), .eval()/.train(), .parameters(), load_state_dict, torch.save

Execution (ROCm tensors; fp32, or fp64 throughout for a module built under --double) — what `_dispatch` does:
  * .eval() (or .train() with encoder dropout 0) and no gradient wanted: `tip_forward`, the inference plans
    (csrc/libtip_hip.so, include/tip_hip.h).  No fallback: a missing library or a CPU tensor under no_grad raises.
    A module built under train_model.py's --double (:84-85: fp64 parameters, fp64 windows) runs `tip_forward_f64` — the same
    function in IEEE double on the fp64 matrix cores; a precision MIX (fp64 windows into an fp32 module or the reverse)
    raises, nothing is converted silently.
  * .train(): the reference's encoder layers carry torch's default dropout p=0.1 (nn.TransformerEncoderLayer default;
    the constructor's `dropout` argument only reaches nn.RNN, where it is a no-op for one layer), live in train mode
    WHETHER OR NOT autograd records.  Such a call — with gradients (train_model.py:132,175,192) or under
    torch.no_grad() (the reference's inference scripts never call .eval(): offline_testing_simple.py:98 is commented
    out, so they run with that dropout accidentally live) — goes to the HIP training step `tip_train_forward` /
    `tip_train_backward` (`_HipTrainFunction`; `tip_train_forward_f64` / `tip_train_backward_f64` for an fp64 module): dropout
    drawn from a counter-based hash, activations stashed, every row computed.  Call .eval() for deterministic, stash-free inference (StreamingEngine warns when it is handed a
    .train()-mode model).  Configurations the training kernels do not cover (fp32 widths outside tip_train_bytes' range — d_model
    not 256/512/1024, rnn_hid_size not a multiple of 64 up to 512 —, gradients w.r.t. the inputs; CPU tensors only when
    autograd records) run the torch-op composite with the same dropout, with a
    warning — never the dropout-free inference kernels, so the behaviour does not depend on the configuration.  A no_grad
    call on CPU tensors raises in either mode: inference values come from the HIP kernels or not at all.
  * .eval() with autograd on (the runners never enter no_grad): HIP forward wrapped in an autograd.Function whose
    backward, if ever called, recomputes the torch-op composite, so .backward() still works.
  The handle's workspace / backward scratch are per (device, stream) (a small LRU, `release_buffers()` drops them).  A module
  may be called from several streams, but its forwards do NOT overlap on the device: the default plans launch cooperating
  kernels that need all of their workgroups resident at once, so the library serialises forwards of different streams (one
  event wait per stream switch; csrc/tip_internal.h, CoopSerial).  A forward fills the GPU by itself.  If another PROCESS holds
  CUs and a cooperating launch loses a hand-off, that launch's rows are NaN and flagged; the next call demotes the handle to the
  plans without hand-offs (TIP_OPT_DEMOTED; `undemote()` restores) and runs — see `_forward_hip`.

Dropout semantics kept from the reference: `nn.Dropout(p)(x)` is constructed inside forward (:73,:77), i.e. it is
always in training mode, so past_state_dropout / in_dropout are live even under .eval().  The HIP path draws the
Bernoulli keep-mask with torch's device RNG and hands it to the kernel (TIP_FWD_KEEP_MASK).
"""
from __future__ import annotations

import math
import operator
import warnings
from typing import Optional

import torch
import torch.nn.functional as F
from torch import nn

from . import lib as _lib


import contextlib as _contextlib
_NO_CTX = _contextlib.nullcontext()
_req_grad = operator.attrgetter("requires_grad")
_version = operator.attrgetter("_version")
_data_ptr = torch.Tensor.data_ptr
_is = operator.is_
_getitem = operator.getitem


class _Leaf(nn.Module):
    """A namespace of explicitly registered parameters (keeps the reference's state-dict key names)."""

    def __init__(self, **shapes):
        super().__init__()
        for name, shape in shapes.items():
            self.register_parameter(name, nn.Parameter(torch.empty(shape)))


class _EncoderLayerParams(nn.Module):
    def __init__(self, d_model: int, d_ff: int):
        super().__init__()
        self.self_attn = _Leaf(in_proj_weight=(3 * d_model, d_model), in_proj_bias=(3 * d_model,))
        self.self_attn.out_proj = _Leaf(weight=(d_model, d_model), bias=(d_model,))
        self.linear1 = _Leaf(weight=(d_ff, d_model), bias=(d_ff,))
        self.linear2 = _Leaf(weight=(d_model, d_ff), bias=(d_model,))
        self.norm1 = _Leaf(weight=(d_model,), bias=(d_model,))
        self.norm2 = _Leaf(weight=(d_model,), bias=(d_model,))


def _uniform_(t: torch.Tensor, bound: float):
    with torch.no_grad():
        t.uniform_(-bound, bound)


class TF_RNN_Past_State(nn.Module):
    def __init__(self, input_size_imu, size_s, rnn_hid_size, tf_hid_size, tf_in_dim, n_heads, tf_layers,
                 dropout, in_dropout, past_state_dropout, with_rnn=True, with_acc_sum=False):
        super().__init__()
        self.input_size_imu = int(input_size_imu)
        self.size_s = int(size_s)
        self.rnn_hid_size = int(rnn_hid_size)
        self.tf_hid_size = int(tf_hid_size)
        self.tf_in_dim = int(tf_in_dim)
        self.n_heads = int(n_heads)
        self.tf_layers = int(tf_layers)
        self.dropout = float(dropout)
        self.in_dropout = float(in_dropout)
        self.past_state_dropout = float(past_state_dropout)
        self.with_rnn = bool(with_rnn)
        self.with_acc_sum = bool(with_acc_sum)
        if self.tf_in_dim % self.n_heads:
            raise AssertionError("embed_dim must be divisible by num_heads")

        n_in = self.input_size_imu + self.size_s + (18 if self.with_acc_sum else 0)
        D, Fh, R, S = self.tf_in_dim, self.tf_hid_size, self.rnn_hid_size, self.size_s
        if self.with_acc_sum:
            print("model with acc sum")
        self.in_linear = _Leaf(weight=(D, n_in), bias=(D,))
        self.tf_encode = nn.Module()
        self.tf_encode.layers = nn.ModuleList([_EncoderLayerParams(D, Fh) for _ in range(self.tf_layers)])
        if self.with_rnn:
            self.rnn = _Leaf(weight_ih_l0=(R, D), weight_hh_l0=(R, R), bias_ih_l0=(R,), bias_hh_l0=(R,))
            self.linear = _Leaf(weight=(S, R), bias=(S,))
        else:
            print("no RNN layer")
            self.rnn = None
            self.linear = _Leaf(weight=(S, D), bias=(S,))
        self.reset_parameters()
        print("number of parameters: %e", sum(p.numel() for p in self.parameters()))

        # HIP-side state (created lazily on the first accelerated forward)
        self._handle: Optional[_lib.Handle] = None
        self._packed_dev: Optional[torch.Tensor] = None
        self._packed_key = None
        self.flow_demotions = 0          # times _answer_handoff switched the one-launch few-stream form off (TIP_OPT_NO_FLOW)
        self._pack_epoch = 0             # bumped by every attach_packed: a reuse ring is only good for the image it was filled under
        self._ring_epoch = {}            # ring data_ptr -> _pack_epoch at its last reuse_reset()
        self._workspace = {}             # (device index, stream handle) -> uint8 tensor: calls on different streams never share one
        self._frozen = False
        self._warned_autograd = False
        self._warned_train_nograd = False
        self._train_scratch = {}         # same keying for the backward scratch
        self.use_hip_training = True     # .train() + autograd on the GPU -> tip_train_forward / tip_train_backward
        self.keep_train_stash = False    # debugging/tests: keep the last activation stash (see train_activation())
        self.last_train_stash = None
        self.demotions = 0               # times a lost hand-off switched the handle to the non-cooperating plans (_forward_hip)
        self.t_max = 80                  # sizing hint handed to the handle; any window length is served (general plan beyond T = 40)
        self._plist_cache = None         # list(self.parameters()): nn.Module.parameters() walks the module tree on every call (~0.6 us per
                                         # parameter and call site; the forward asks several times per frame), see _plist()
        self._pslots = None              # where every entry of _plist_cache lives: ([_parameters dict], [name]) + ([_modules dict], [name], [module])
        self._train_ok_cache = {}
        self._params_sig = None          # (dtype, all on the GPU) of the parameters ...
        self._sig_ptrs = None            # ... valid while their storage pointers are these
        self._ws_bytes_cache = {}
        self._chunk_cache = {}
        self._relaunches = 0
        self._eval_state = None          # few-stream .eval() call: (device, parameter storage pointers, sum of version counters) of the last validated call
        self._fast_state = None          # few-window .train()-mode call (the unedited runner's): what the last fully validated call saw
        self._backward_seen = False      # a .backward() has gone through this module's .train()-mode call: real training, not a runner

    # ------------------------------------------------------------------------------------------
    # initialisation: same distributions torch's nn.Linear / nn.MultiheadAttention / nn.LayerNorm / nn.RNN use
    # ------------------------------------------------------------------------------------------
    def reset_parameters(self):
        def linear_(leaf):
            fan_in = leaf.weight.shape[1]
            _uniform_(leaf.weight, 1.0 / math.sqrt(fan_in))
            _uniform_(leaf.bias, 1.0 / math.sqrt(fan_in))

        linear_(self.in_linear)
        for layer in self.tf_encode.layers:
            w = layer.self_attn.in_proj_weight
            _uniform_(w, math.sqrt(6.0 / (w.shape[0] + w.shape[1])))  # xavier_uniform_
            with torch.no_grad():
                layer.self_attn.in_proj_bias.zero_()
                layer.self_attn.out_proj.bias.zero_()
                layer.norm1.weight.fill_(1.0), layer.norm1.bias.zero_()
                layer.norm2.weight.fill_(1.0), layer.norm2.bias.zero_()
            _uniform_(layer.self_attn.out_proj.weight, 1.0 / math.sqrt(self.tf_in_dim))
            linear_(layer.linear1)
            linear_(layer.linear2)
        if self.rnn is not None:
            for p in self.rnn.parameters():
                _uniform_(p, 1.0 / math.sqrt(self.rnn_hid_size))
        linear_(self.linear)

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    ENCODER_DROPOUT = 0.1   # nn.TransformerEncoderLayer's default; the reference never overrides it (:26-28)

    def forward(self, x_imu, x_s):
        return self._dispatch(x_imu, x_s, last_row_only=False)

    def forward_last(self, x_imu, x_s, *, workspace=None, out=None):
        """Row T-1 of every window only ([B, size_s]) — what RTRunnerMin.step consumes
        (real_time_runner_minimal.py:150).  Extension over the reference API; same numerics as forward()[:, -1].
        workspace / out (inference kernels only): caller-owned uint8 workspace of at least workspace_bytes(B, T) and output
        tensor [B, size_s] — what a HIP-graph capture must own itself (StreamingEngine), since the module's per-stream
        workspaces are an LRU that may drop a buffer a live graph still points at."""
        if workspace is not None or out is not None:
            if self.training or torch.is_grad_enabled():
                raise RuntimeError("tip_amd: workspace= / out= serve the inference kernels (.eval() under torch.no_grad())")
            return self._forward_hip(x_imu, x_s, True, workspace=workspace, out=out)
        return self._dispatch(x_imu, x_s, last_row_only=True)

    def reuse_cache(self, n_streams: int) -> torch.Tensor:
        """A cleared ring for forward_last_reuse: 40 slots x 4 KiB per stream (in_linear row + layer-0 q | k | v row of each of the
        last 40 frames) behind a 256-byte header of frame tags."""
        h = self._ensure_handle()
        dev = self.in_linear.weight.device
        if dev.type != "cuda":
            raise RuntimeError("tip_amd: the reuse ring lives in HBM — move the module to the GPU first (.cuda())")
        cache = torch.empty(h.reuse_cache_bytes(int(n_streams)), dtype=torch.uint8, device=dev)
        self.reuse_reset(cache)
        return cache

    def reuse_reset(self, cache: torch.Tensor):
        """Forget every frame in the ring (the streams restart, or the parameters changed: the rows in it belong to the old weights)."""
        st = _lib.load().tip_reuse_reset(cache.data_ptr(), cache.numel(), torch.cuda.current_stream(cache.device).cuda_stream)
        if st < 0:
            raise _lib.TipStatusError(st, _lib.load().tip_strerror(st).decode())
        if len(self._ring_epoch) > 64:
            self._ring_epoch.clear()
        self._ring_epoch[cache.data_ptr()] = None      # filled under whatever image the next forward_last_reuse runs on

    def forward_last_reuse(self, x_imu, x_s, cache: torch.Tensor, frame_idx: int, *, frame_ctr_ptr=None, workspace=None, out=None):
        """forward_last for lock-stepped streams whose windows slide by one frame per call, with SURVEY.md 7-7's exact reuse
        (include/tip_hip.h: tip_forward_reuse): a frame's in_linear row and layer-0 Q / K / V rows are computed once, when the frame
        enters, and read from `cache` (reuse_cache(B)) in the 39 later windows it appears in.  frame_idx: consecutive across calls
        (rows 0 .. T-2 of this call's windows must be rows 1 .. T-1 of the previous call's); frame_ctr_ptr: device address of an int
        that holds it instead (HIP graphs).  Valid only where those rows are a function of the frame alone: .eval(), past_state_dropout
        = 0, in_dropout = 0 — anything else raises.  Bit-identical to forward_last under set_plan("fused2") (AUTO's choice at 1024
        streams)."""
        if self.training or torch.is_grad_enabled() or self.past_state_dropout > 0.0 or self.in_dropout > 0.0:
            raise RuntimeError("tip_amd: forward_last_reuse caches per-frame rows across windows, which is exact only without the "
                               "stochastic parts: .eval() under torch.no_grad(), past_state_dropout = 0, in_dropout = 0 "
                               "(the reference's fresh nn.Dropout at :77 is live even in .eval())")
        if not (x_imu.is_cuda and x_s.is_cuda) or x_imu.dtype != torch.float32 or x_s.dtype != torch.float32:
            raise RuntimeError("tip_amd: forward_last_reuse serves fp32 windows on the GPU")
        dev = x_imu.device
        if x_imu.dim() != 3 or x_s.dim() != 3 or x_imu.shape[:2] != x_s.shape[:2]:
            raise RuntimeError("expected x_imu [B,T,n_imu] and x_s [B,T,size_s]")
        B, T = int(x_imu.shape[0]), int(x_imu.shape[1])
        if x_imu.shape[2] != self.input_size_imu + (18 if self.with_acc_sum else 0) or x_s.shape[2] != self.size_s:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied: got feature widths {x_imu.shape[2]}+{x_s.shape[2]}")
        h = self._ensure_handle()
        with (torch.cuda.device(dev) if torch.cuda.current_device() != dev.index else _NO_CTX):
            if self._packed_dev is None or self._packed_dev.device != dev or (not self._frozen and self._packed_key != self._param_key(dev)):
                self.refresh_packed(dev)
            # The ring's rows are functions of the WEIGHTS too: it is good for the packed image it was filled under and no other —
            # whoever re-packed in between (this call, a plain forward after an optimiser step, attach_packed of a broadcast image)
            ep = self._ring_epoch.get(cache.data_ptr())
            if ep is None:
                self._ring_epoch[cache.data_ptr()] = self._pack_epoch
            elif ep != self._pack_epoch:
                self.reuse_reset(cache)
                raise RuntimeError("tip_amd: the parameters changed while the reuse ring held rows computed with the old ones — "
                                   "the ring was cleared; re-prime it with 40 consecutive frames (StreamingEngine.reset())")
            x_imu_c, x_s_c = x_imu.contiguous(), x_s.contiguous()
            y = out if out is not None else torch.empty((B, self.size_s), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            need = self._ws_bytes_cache.get((B, T))
            if need is None:
                need = self._ws_bytes_cache[(B, T)] = h.workspace_bytes(B, T)
            ws = workspace if workspace is not None else self._stream_buffer(self._workspace, dev, stream, need)
            h.forward_reuse(x_imu_c.data_ptr(), x_s_c.data_ptr(), y.data_ptr(), B, T, _lib.TIP_FWD_LAST_ROW_ONLY, cache.data_ptr(),
                            cache.numel(), int(frame_idx), frame_ctr_ptr, ws.data_ptr(), ws.numel(), stream)
        return y

    def chunk_batch(self, T: int, fp64: bool = False) -> int:
        """Windows per launch sequence when a batch exceeds what one tip_forward call serves (tip_max_batch): the library's
        limit, rounded down to whole rounds of 256 windows when it is that large (full waves of one-window workgroups)."""
        key = (int(T), bool(fp64))
        c = self._chunk_cache.get(key)
        if c is None:
            m = max(1, self._ensure_handle().max_batch(max(int(T), 1), fp64=fp64))
            c = self._chunk_cache[key] = m - m % 256 if m >= 512 else m
        return c

    def workspace_bytes(self, B: int, T: int) -> int:
        return self._ensure_handle().workspace_bytes(int(B), int(T))

    def _plist(self):
        """The module's parameters as a cached list (state-dict order), VALIDATED on every call: each entry must still be the object
        its owner's `_parameters` dict holds and every submodule the one its parent's `_modules` holds (one identity test each through
        C-level maps: ~2.5 us for the 56 + 33 slots).  Whatever swaps a Parameter or a submodule without going through this module's
        own `_apply` — torch.func.functional_call (writes `_parameters[name]` directly), load_state_dict(assign=True), re-assigning
        `model.rnn.weight_hh_l0`, replacing a submodule — is seen by the next call, which rebuilds the list and drops everything
        derived from it.  nn.Module.parameters() walks the module tree (~40 us): that is what the cache is for."""
        pl = self._plist_cache
        if pl is not None:
            (pd, pn), (md, mn, ml) = self._pslots
            try:
                if all(map(_is, map(_getitem, pd, pn), pl)) and all(map(_is, map(_getitem, md, mn), ml)):
                    return pl
            except KeyError:          # a parameter / submodule was deleted
                pass
        return self._rebuild_plist()

    def _rebuild_plist(self):
        pd, pn, pl, md, mn, ml = [], [], [], [], [], []

        def walk(mod):                # the traversal order of nn.Module.named_parameters(): own parameters, then the children in order
            for n, q in mod._parameters.items():
                if q is not None:
                    pd.append(mod._parameters), pn.append(n), pl.append(q)
            for n, c in mod._modules.items():
                if c is not None:
                    md.append(mod._modules), mn.append(n), ml.append(c)
                    walk(c)
        walk(self)
        if len({id(q) for q in pl}) != len(pl) or [id(q) for q in pl] != [id(q) for q in self.parameters()]:   # tied / unusual layouts: no slots
            pl = list(self.parameters())
            pd, pn, md, mn, ml = [self.__dict__], ["_never_valid"], [], [], []    # validation always fails -> rebuilt on every call
        self._plist_cache, self._pslots = pl, ((pd, pn), (md, mn, ml))
        self._params_sig = self._sig_ptrs = None
        self._train_ok_cache = {}
        self._fast_state = None
        self._eval_state = None
        self._packed_key = None       # the packed image is re-validated against the new list
        return pl

    def _apply(self, fn, *args, **kwargs):
        self._plist_cache = None
        self._params_sig = self._sig_ptrs = None
        self._train_ok_cache = {}
        self._fast_state = None
        self._eval_state = None
        return super()._apply(fn, *args, **kwargs)

    def _dispatch(self, x_imu, x_s, last_row_only: bool):
        drawn = None
        if self._fast_state is not None and self.training:      # the unedited runner's call, seen and validated before
            y = self._few_window_fast(x_imu, x_s)
            if y is not None:
                return y[:, -1] if last_row_only else y
            # (seeds the dropped attempt took from the generator: the re-run uses THEM, so a call draws once whichever way it is served)
            drawn, self._drawn_seeds = self._drawn_seeds, None
        plist = self._plist()
        needs_grad = torch.is_grad_enabled() and (x_imu.requires_grad or x_s.requires_grad or any(map(_req_grad, plist)))
        # .train() mode draws the encoder's dropout whether or not autograd records (nn.TransformerEncoderLayer p = 0.1): with
        # gradients wanted, or with dropout to apply, the call goes to the training kernels; .train() + no_grad + p = 0 is the
        # same function as .eval() and takes the inference kernels below
        if self.training and (needs_grad or self.ENCODER_DROPOUT > 0.0) and self._hip_train_ok(x_imu, x_s):
            # train_model.py:171-196 on the HIP path: forward with saved activations + live encoder dropout, HIP backward
            if not needs_grad and not self._warned_train_nograd:
                warnings.warn("tip_amd: .train()-mode call under torch.no_grad() — running the TRAINING forward (encoder dropout "
                              "p=0.1 live, every row computed, activation stash allocated), as the reference does when .eval() "
                              "is never called (offline_testing_simple.py:98); call .eval() for the deterministic inference kernels")
                self._warned_train_nograd = True
            xi = F.dropout(x_imu, self.in_dropout, training=True) if self.in_dropout > 0.0 else x_imu   # :73
            # :77 and the encoder's dropout: two seeds from torch's CPU generator (torch.manual_seed governs them; no device sync).
            # The past-state keep mask is a function of (p, seed) — the library's counter-based hash — drawn inside the first kernel
            # (few windows) or written out by tip_draw_keep_mask (_hash_keep_mask), never by three torch kernels per call.
            seeds = drawn or self._draw_seeds()
            mask = seeds[1] if 0.0 < self.past_state_dropout < 1.0 else self._draw_keep_mask(x_s)
            # A few windows (the unedited runner's call): the kernels are queued HERE, before autograd's bookkeeping for the 56
            # parameter inputs (~20 us of host time that then runs beside the GPU instead of in front of it); the Function below
            # picks the launched forward up instead of launching its own.
            self._pre_launched = self._few_window_train_launch(xi, x_s, mask, float(self.ENCODER_DROPOUT), seeds[0]) or False
            served_few = bool(self._pre_launched)
            try:
                y = _HipTrainFunction.apply(self, xi, x_s, mask, float(self.ENCODER_DROPOUT), seeds[0], *plist)
            finally:
                self._pre_launched = None
            if served_few and isinstance(mask, int) and self.in_dropout <= 0.0 and not self._frozen:
                # everything about this call was checked the slow way: the next ones of the same kind take _few_window_fast
                self._fast_state = (x_imu.device, list(map(_data_ptr, plist)), sum(map(_version, plist)))
            return y[:, -1] if last_row_only else y
        if (needs_grad and (self.training or not x_imu.is_cuda)) or (self.training and self.ENCODER_DROPOUT > 0.0 and x_imu.is_cuda):
            # not covered by the HIP training step: the torch-op composite, with the encoder dropout .train() implies
            if not self._warned_autograd:
                warnings.warn("tip_amd: .train()-mode call (or autograd on CPU) that the HIP training kernels do not cover "
                              "(unsupported widths, CPU tensors or gradients w.r.t. the inputs) — using the torch-op training "
                              "composite with encoder dropout p=0.1 live, as in the reference; call .eval() for the "
                              "inference kernels")
                self._warned_autograd = True
            y = self._forward_torch_ops(x_imu, x_s)
            return y[:, -1] if last_row_only else y
        if not needs_grad:
            return self._forward_hip(x_imu, x_s, last_row_only)
        # eval mode, autograd on (the runners never enter no_grad): the values come from the inference kernels, exactly as under
        # no_grad; a .backward() — if it ever comes — runs the HIP training step on the saved inputs with dropout off (activation
        # stash produced then), or differentiates the torch-op composite where the HIP step does not apply (input gradients,
        # unsupported widths)
        xi = F.dropout(x_imu, self.in_dropout, training=True) if self.in_dropout > 0.0 else x_imu   # :73
        mask = self._draw_keep_mask(x_s)
        if self._hip_train_ok(xi, x_s):
            return _HipForwardHipBackward.apply(self, last_row_only, xi, x_s, mask, *plist)
        return _HipForwardTorchBackward.apply(self, last_row_only, xi, x_s, mask, *plist)

    _seed_buf = None
    _drawn_seeds = None

    def _draw_seeds(self):
        """Two 62-bit seeds (encoder dropout, past-state keep mask) from torch's CPU generator: torch.manual_seed governs them, no
        device sync.  One in-place draw into a cached two-element tensor (1.1 us; torch.randint + tolist: 2.3)."""
        b = self._seed_buf
        if b is None:
            b = self._seed_buf = torch.empty(2, dtype=torch.int64)
        return b.random_(0, 2 ** 62).tolist()

    def _few_window_fast(self, x_imu, x_s):
        """The unedited runner's call (real_time_runner_minimal.py:149 on a module that never left .train(): B = 1, T <= 40, autograd
        recording, nobody differentiates) once a call of the same kind has been validated the slow way (_dispatch sets _fast_state).
        The kernels go on the GPU's queue FIRST, behind a handful of attribute tests; what the slow path checks in front of its launch
        — every parameter still the object the cache holds, storage pointers and version counters those the packed image was built
        from, requires_grad flags (~12 us of host time) — is checked HERE while the GPU works.  If any of it fails, the result is
        dropped, the state forgotten and the slow path runs the call again (its launch overwrites nothing the caller has seen).
        Returns y, or None when the call is not of that kind any more."""
        dev, ptrs, vsum = self._fast_state
        pk = self._packed_dev
        if not (x_imu.is_cuda and x_s.is_cuda and x_imu.dtype is torch.float32 and x_s.dtype is torch.float32 and x_imu.dim() == 3
                and x_s.dim() == 3 and x_imu.shape[0] <= self.LAZY_STASH_MAX_BATCH and x_imu.shape[:2] == x_s.shape[:2]
                and x_imu.device == dev and pk is not None and pk.device == dev and self.use_hip_training and not self.keep_train_stash
                and not self._backward_seen and not x_imu.requires_grad and not x_s.requires_grad and self.in_dropout <= 0.0
                and self._handle is not None and 0.0 < self.past_state_dropout < 1.0 and torch.cuda.current_device() == dev.index):
            self._fast_state = None
            return None
        seeds = self._draw_seeds()
        p_drop = float(self.ENCODER_DROPOUT)
        pre = self._few_window_train_launch(x_imu, x_s, seeds[1], p_drop, seeds[0], trusted=True)
        # -- beside the GPU from here on --
        plist = self._plist()
        if not pre or list(map(_data_ptr, plist)) != ptrs or sum(map(_version, plist)) != vsum or self._fast_state is None:
            self._fast_state = None
            self._drawn_seeds = seeds
            self._relaunches += 1 if pre else 0
            return None
        if not (torch.is_grad_enabled() and any(map(_req_grad, plist))):
            # (.train() under no_grad, or every parameter frozen: the same values; no graph to record)
            return pre[0]
        self._pre_launched = pre
        try:
            return _HipTrainFunction.apply(self, x_imu, x_s, seeds[1], p_drop, seeds[0], *plist)
        finally:
            self._pre_launched = None

    LAZY_STASH_MAX_BATCH = 32   # .train()-mode calls of up to this many windows run without an activation stash (see _HipTrainFunction)
    _pre_launched = None

    def _few_window_train_launch(self, x_imu, x_s, mask, p_drop, seed, trusted=False):
        """.train()-mode call of a few windows (the unedited runner's B = 1 call, real_time_runner_minimal.py:149, on a module that
        never left .train() mode): the same function on the few-stream kernels (tip_forward_dropout: same dropout decisions as
        tip_train_forward, no activation stash, ~0.2 ms instead of ~0.8 at one window per CU).  Returns (y, xi, xs, mask or its
        seed, scale) with the kernels queued, or None when the call is not served that way (batch, precision, configuration,
        keep_train_stash): _HipTrainFunction then runs tip_train_forward.  If .backward() is called after all, the stash is
        produced then."""
        B = int(x_imu.shape[0])
        # (once a .backward() has been seen on this module the caller is training, not streaming: a lazy forward would be paid twice —
        #  here and again, with the stash, inside backward — and the weight image re-packed after every optimizer step: ADVICE r05)
        if B > self.LAZY_STASH_MAX_BATCH or self.keep_train_stash or x_imu.dtype != torch.float32 or self._backward_seen:
            return None
        n_imu = self.input_size_imu + (18 if self.with_acc_sum else 0)
        if x_imu.shape[2] != n_imu or x_s.shape[2] != self.size_s:
            return None                                   # _HipTrainFunction.forward raises the reference's shape error
        h = self._ensure_handle()
        dev = x_imu.device
        T = int(x_imu.shape[1])
        # (the library launches on the CURRENT device: switch only when it is another one — the context manager costs ~3 us)
        ctx = torch.cuda.device(dev) if torch.cuda.current_device() != dev.index else None
        if ctx is not None:
            ctx.__enter__()
        try:
            xi, xs = x_imu.contiguous(), x_s.contiguous()
            pd = self.past_state_dropout
            mask_ptr, scale = None, (1.0 / (1.0 - pd) if pd < 1.0 else 0.0) if mask is not None else 1.0
            state_seed = mask if isinstance(mask, int) else None     # the keep mask as (p, seed): drawn by the library
            if state_seed is not None:
                mask = None
            elif mask is not None:
                mask = mask.to(torch.float32).contiguous()
                mask_ptr = mask.data_ptr()
            y = torch.empty((B, T, self.size_s), dtype=torch.float32, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            if not self._forward_dropout_hip(h, xi, xs, y, B, T, mask_ptr, scale, pd if state_seed is not None else 0.0,
                                             state_seed or 0, p_drop, seed, stream, trusted):
                return None
        finally:
            if ctx is not None:
                ctx.__exit__(None, None, None)
        return y, xi, xs, (mask if state_seed is None else state_seed), scale

    def _forward_dropout_hip(self, h, xi, xs, y, B, T, mask_ptr, scale, p_state, state_seed, p_drop, seed, stream, trusted=False) -> bool:
        """tip_forward_dropout into `y`; False when the library does not serve this call that way (configuration, window length,
        demoted handle): the caller then runs tip_train_forward.  trusted: the packed image was validated by the caller's protocol
        (_few_window_fast checks the parameters AFTER the launch and discards the result if they changed)."""
        dev = xi.device
        if not trusted and (self._packed_dev is None or self._packed_dev.device != dev or
                            (not self._frozen and self._packed_key != self._param_key(dev))):
            self.refresh_packed(dev)
        try:
            need = self._ws_bytes_cache.get((B, T))
            if need is None:
                need = self._ws_bytes_cache[(B, T)] = h.workspace_bytes(B, T)
            ws = self._stream_buffer(self._workspace, dev, stream, need)
            h.forward_dropout(xi.data_ptr(), xs.data_ptr(), y.data_ptr(), B, T, _lib.TIP_FWD_KEEP_MASK if mask_ptr else 0, mask_ptr,
                              scale, p_state, state_seed, p_drop, seed, ws.data_ptr(), ws.numel(), stream)
        except _lib.TipHandoffError:
            # An EARLIER launch of this handle lost an inter-workgroup hand-off (a co-tenant held CUs): same answer as _forward_hip's —
            # the first time, clear the word, demote the handle to the plans without cooperating kernels and let THIS call run there
            # (False: the caller takes tip_train_forward, whose recurrence then runs one workgroup per tile); with TIP_OPT_AUTO_DEMOTE
            # off, or on a handle that is demoted already, report it.
            how = self._answer_handoff(h)
            if how is None:
                raise
            self._fast_state = None
            if how == "chain":      # only the one-launch form is gone: the same entry point serves this call on the launch chain
                h.forward_dropout(xi.data_ptr(), xs.data_ptr(), y.data_ptr(), B, T, _lib.TIP_FWD_KEEP_MASK if mask_ptr else 0, mask_ptr,
                                  scale, p_state, state_seed, p_drop, seed, ws.data_ptr(), ws.numel(), stream)
                return True
            return False
        except _lib.TipStatusError as e:
            if e.status == _lib.TIP_ERR_UNSUPPORTED_CONFIG:
                return False
            raise
        return True

    def _hip_train_ok(self, x_imu, x_s) -> bool:
        """True when the HIP training step (libtip_hip tip_train_*, or tip_train_*_f64 for a module built under --double) covers
        this call: CUDA tensors of the parameters' precision (fp32 or fp64, no mix), gradients wanted for parameters only, and
        a configuration the kernels support (TIP_ERR_UNSUPPORTED_CONFIG otherwise)."""
        if not self.use_hip_training or not (x_imu.is_cuda and x_s.is_cuda):
            return False
        pdt = self.in_linear.weight.dtype
        if pdt not in (torch.float32, torch.float64):
            return False
        if x_imu.dtype != pdt or x_s.dtype != pdt:
            return False
        if (x_imu.requires_grad or x_s.requires_grad) and pdt != torch.float32:
            return False          # gradients w.r.t. the inputs: tip_train_input_grads (fp32 step only)
        if x_imu.dim() != 3 or x_s.dim() != 3 or x_imu.shape[:2] != x_s.shape[:2]:
            return False
        # (dtype, every parameter of that dtype on the GPU): valid while the parameters' storage pointers are the ones it was computed
        # for — a submodule's own .double() / .cuda() keeps the Parameter objects and swaps their storage
        plist = self._plist()
        ptrs = list(map(_data_ptr, plist))
        sig = self._params_sig
        if sig is None or ptrs != self._sig_ptrs:
            d0 = plist[0].dtype
            sig = self._params_sig = (d0, all([p.dtype == d0 and p.is_cuda for p in plist]))
            self._sig_ptrs = ptrs
        if not sig[1] or sig[0] != pdt or not self.in_linear.weight.is_cuda:
            return False
        key = (int(x_imu.shape[0]), int(x_imu.shape[1]), pdt)
        ok = self._train_ok_cache.get(key)
        if ok is None:
            try:
                self._ensure_handle().train_bytes(key[0], key[1], fp64=pdt == torch.float64)
                ok = True
            except _lib.TipStatusError:
                ok = False
            if len(self._train_ok_cache) > 256:
                self._train_ok_cache.clear()
            self._train_ok_cache[key] = ok
        return ok

    def train_activation(self, what: int, layer: int = 0) -> torch.Tensor:
        """One stashed activation of the last HIP training forward (needs keep_train_stash = True): `what` is one of
        lib.TIP_SAVED_*; returns a [B*T, width] float32 view."""
        saved, B, T = self.last_train_stash
        off, n = self._ensure_handle().train_saved_view(B, T, what, layer)
        return saved.view(torch.float32)[off:off + n].view(B * T, -1)

    def _hash_keep_mask(self, x_s, state_seed: int):
        """The past-state keep mask (:77) of (past_state_dropout, state_seed) as a tensor like x_s: tip_draw_keep_mask."""
        m = torch.empty(x_s.shape, dtype=torch.float32, device=x_s.device)
        with torch.cuda.device(x_s.device):
            _lib.draw_keep_mask(self.past_state_dropout, state_seed, m.data_ptr(), m.numel(), torch.cuda.current_stream(x_s.device).cuda_stream)
        return m if x_s.dtype == torch.float32 else m.to(x_s.dtype)

    def _draw_keep_mask(self, x_s):
        """Bernoulli keep-mask of the always-on past-state dropout (:77); None when p == 0."""
        if self.past_state_dropout <= 0.0:
            return None
        return (torch.rand_like(x_s) >= self.past_state_dropout).to(x_s.dtype)

    # -- HIP path -------------------------------------------------------------------------------
    def _tip_config(self) -> _lib.TipConfig:
        return _lib.TipConfig(self.input_size_imu, self.size_s, self.rnn_hid_size, self.tf_hid_size, self.tf_in_dim,
                              self.n_heads, self.tf_layers, 1 if self.with_rnn else 0, 1 if self.with_acc_sum else 0,
                              int(self.t_max))

    def _ensure_handle(self) -> _lib.Handle:
        if self._handle is None:
            self._handle = _lib.Handle(self._tip_config())
            names = [n for n, _ in self._handle.tensor_table()]
            if names != list(self.state_dict().keys()):
                raise RuntimeError("libtip_hip tensor table does not match the module's state_dict order")
        return self._handle

    def _param_key(self, device):
        plist = self._plist()
        return (str(device), list(map(_data_ptr, plist)), list(map(_version, plist)))

    def pack_host(self) -> torch.Tensor:
        """Build the packed weight image (uint8 CPU tensor) from the current parameters."""
        h = self._ensure_handle()
        host = [p.detach().to("cpu", torch.float32).contiguous() for p in self.state_dict().values()]
        out = torch.empty(h.packed_bytes(), dtype=torch.uint8)
        h.pack_weights([t.data_ptr() for t in host], out.data_ptr(), out.numel())
        return out

    def attach_packed(self, packed_dev: torch.Tensor):
        """Point the kernels at a packed image already resident on this GPU (e.g. received by RCCL broadcast)."""
        h = self._ensure_handle()
        assert packed_dev.is_cuda and packed_dev.dtype == torch.uint8 and packed_dev.is_contiguous()
        h.attach_packed(packed_dev.data_ptr(), packed_dev.numel())
        self._packed_dev = packed_dev
        self._packed_key = self._param_key(packed_dev.device)
        self._pack_epoch += 1

    def pack_device(self, device=None) -> torch.Tensor:
        """Build the packed weight image ON the GPU from the live parameters (tip_pack_weights_device): same bytes as
        pack_host(), no host round trip."""
        h = self._ensure_handle()
        device = torch.device(device) if device is not None else next(self.parameters()).device
        tensors = [p.detach().to(device, torch.float32).contiguous() for p in self.state_dict().values()]
        out = torch.empty(h.packed_bytes(), dtype=torch.uint8, device=device)
        with torch.cuda.device(device):
            h.pack_weights_device([t.data_ptr() for t in tensors], out.data_ptr(), out.numel(),
                                  torch.cuda.current_stream(device).cuda_stream)
        return out

    def refresh_packed(self, device=None):
        device = torch.device(device) if device is not None else next(self.parameters()).device
        if device.type == "cuda" and all([p.is_cuda for p in self._plist()]):
            self.attach_packed(self.pack_device(device))          # parameters already live on the GPU: pack there
        else:
            self.attach_packed(self.pack_host().to(device, non_blocking=False))

    def freeze_packed(self, frozen: bool = True):
        """Skip the per-call 'did the parameters change' check (streaming hot loop)."""
        self._frozen = bool(frozen)

    def set_plan(self, plan: str = "auto", rnn_cluster: int = 0, profile: int = 0):
        """Execution plan of the forward (measurement / deployment knob; AUTO is right for production).  rnn_cluster: 0 = auto
        (four-window tiles on 4-workgroup clusters, lib.TIP_RNN_CLUSTER_ROWS4, when rnn_hidden is 512), 1 / 2 / 4 / 8 / 16 =
        workgroups per 16-window tile (1: no inter-workgroup hand-off in the recurrence).  profile: 1 = per-stage timers
        (profile_read())."""
        h = self._ensure_handle()
        if plan in ("fused16", "general16", "fused2s"):
            raise RuntimeError(f"tip_amd: plan '{plan}' was retired in round 6 (exploratory split-fp16 emulation / superseded pair-split plan)")
        # "fused1s": one window on four workgroups while 4 B <= #CUs, on two otherwise; "fused1s2" / "fused1s4" pin the form
        h.set_option(_lib.TIP_OPT_F1S_PARTS, {"fused1s2": 2, "fused1s4": 4}.get(plan, 0))
        h.set_option(_lib.TIP_OPT_PLAN, {"auto": 0, "general": 1, "fused": 2, "latency": 3, "fused2": 4, "fusedh": 6, "fused1s": 10, "fused1s2": 10, "fused1s4": 10}[plan])
        h.set_option(_lib.TIP_OPT_RNN_CLUSTER, int(rnn_cluster))
        h.set_option(_lib.TIP_OPT_PROFILE, int(profile))

    def check_handoffs(self, synchronize: bool = True, clear: bool = False):
        """Raise lib.TipHandoffError if any launch so far lost an inter-workgroup hand-off (its outputs are NaN-poisoned, never
        finite-but-wrong).  The cooperating plans need the GPU to themselves: a co-tenant holding CUs can starve a partner
        workgroup.  Without this call the NEXT forward raises the same error (the flag is sticky until clear=True)."""
        if self._handle is None:
            return
        if synchronize and torch.cuda.is_available():
            torch.cuda.synchronize()
        try:
            self._handle.check(clear=False)
        finally:
            if clear:
                try:
                    self._handle.check(clear=True)
                except _lib.TipHandoffError:
                    pass

    def _answer_handoff(self, h):
        """A TipHandoffError is pending on `h` (an EARLIER launch lost an inter-workgroup hand-off: its outputs were NaN and flagged).
        What the host does about it, once per kind: if only the one-launch few-stream form lost its XCD placement (kernels of another
        stream were dispatched beside it: TIP_OPT_HANDOFF_KIND = 2), switch THAT form off — the launch chain needs co-residency only and
        costs 10 us per forward; anything else (or a second loss): the plans without any hand-off (TIP_OPT_DEMOTED).  Returns "chain" /
        "demoted" with the sticky word cleared, or None when the error is to be reported (TIP_OPT_AUTO_DEMOTE off, nothing left to
        give up)."""
        if not h.get_option(_lib.TIP_OPT_AUTO_DEMOTE):
            return None
        if h.get_option(_lib.TIP_OPT_HANDOFF_KIND) == 2 and not h.get_option(_lib.TIP_OPT_NO_FLOW):
            warnings.warn("tip_amd: an earlier few-stream forward found its workgroups spread over several XCDs (kernels of another stream "
                          "or process were dispatched beside it) — its outputs were NaN.  This model now runs the few-stream plan as a "
                          "launch chain (TIP_OPT_NO_FLOW: ~10 us per forward slower, no placement requirement); model.undemote() "
                          "restores the default")
            h.check_clear()
            h.set_option(_lib.TIP_OPT_NO_FLOW, 1)
            self.flow_demotions += 1
            return "chain"
        if h.get_option(_lib.TIP_OPT_DEMOTED):
            return None
        warnings.warn("tip_amd: an earlier forward lost an inter-workgroup hand-off (is another process or stream holding "
                      "CUs of this GPU?) — its outputs were NaN.  This model now runs the plans that need no co-resident "
                      "workgroups (TIP_OPT_DEMOTED: slower, safe under co-tenancy); model.undemote() restores the default")
        h.check_clear()
        h.set_option(_lib.TIP_OPT_DEMOTED, 1)
        self.demotions += 1
        return "demoted"

    def undemote(self):
        """Back to the default (cooperating) plans after a self-demotion (see _answer_handoff): the GPU is this process's again."""
        if self._handle is not None:
            self._handle.set_option(_lib.TIP_OPT_DEMOTED, 0)
            self._handle.set_option(_lib.TIP_OPT_NO_FLOW, 0)

    def is_demoted(self) -> bool:
        return bool(self._handle is not None and self._handle.get_option(_lib.TIP_OPT_DEMOTED))

    def profile_read(self):
        return self._ensure_handle().profile_read()

    def hip_forward_count(self) -> int:
        # (forwards as the caller counts them: a launch sequence that was queued on a trusted weight image and queued again after the
        #  parameters turned out to have changed — _forward_hip, _few_window_fast — is one forward)
        return self._handle.forward_count() - self._relaunches if self._handle is not None else 0

    MAX_STREAM_BUFFERS = 4   # scratch buffers kept per table (least recently used beyond that are dropped)

    @staticmethod
    def _stream_buffer(table: dict, dev, stream: int, nbytes: int) -> torch.Tensor:
        """The scratch buffer of (device, stream): grown on demand, never shared between streams.  The library serialises the
        forwards of different streams on the device (cooperating kernels: csrc/tip_internal.h, CoopSerial), but a buffer per
        stream keeps a forward queued on stream B from scribbling over the workspace of one still running on stream A only
        because of that ordering — it does not depend on it.  The table is a small LRU (short-lived streams — per-request
        streams, ExternalStream, a destroyed CU-mask stream whose handle value is later recycled — must not pin hundreds of MB
        for the life of the module); `release_buffers()` drops everything."""
        key = (dev.index, int(stream))
        buf = table.pop(key, None)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        table[key] = buf                                  # (re)inserted last = most recently used
        while len(table) > TF_RNN_Past_State.MAX_STREAM_BUFFERS:
            table.pop(next(iter(table)))
        return buf

    def release_buffers(self):
        """Drop every cached workspace / training scratch buffer (they are re-allocated on demand)."""
        self._workspace.clear()
        self._train_scratch.clear()

    def _forward_hip(self, x_imu, x_s, last_row_only: bool, keep_mask="draw", apply_in_dropout=True, workspace=None, out=None):
        if not (x_imu.is_cuda and x_s.is_cuda):
            raise RuntimeError("tip_amd.TF_RNN_Past_State: the inference forward runs on an MI355X through "
                               "libtip_hip.so only — move the module and its inputs to the GPU (.cuda()); "
                               "there is no CPU fallback")
        pdt = self.in_linear.weight.dtype
        if x_imu.dtype != x_s.dtype or x_imu.dtype != pdt or pdt not in (torch.float32, torch.float64):
            raise RuntimeError("tip_amd.TF_RNN_Past_State: the HIP path computes in fp32 (or, for a module built under "
                               "train_model.py's --double, in fp64) and never converts tensors silently; got inputs "
                               f"{x_imu.dtype}/{x_s.dtype} for {pdt} parameters")
        f64 = pdt == torch.float64          # train_model.py:84-85: the whole model in double -> tip_forward_f64
        dev = x_imu.device
        if x_imu.dim() != 3 or x_s.dim() != 3 or x_imu.shape[:2] != x_s.shape[:2]:
            raise RuntimeError("expected x_imu [B,T,n_imu] and x_s [B,T,size_s]")
        B, T = int(x_imu.shape[0]), int(x_imu.shape[1])
        n_imu = self.input_size_imu + (18 if self.with_acc_sum else 0)
        if x_imu.shape[2] != n_imu or x_s.shape[2] != self.size_s:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied: got feature widths "
                               f"{x_imu.shape[2]}+{x_s.shape[2]}, in_linear expects {n_imu}+{self.size_s}")
        h = self._ensure_handle()
        # The kernels address their activations through 32-bit buffer descriptors (tip_forward) / 32-bit element offsets and grid
        # limits (tip_forward_f64): the library says how many windows one call serves (tip_max_batch; TIP_ERR_UNSUPPORTED_CONFIG
        # beyond: B > 13 107 at T = 40 for the paper configuration).  Streams are independent (no op in :60-102 crosses batch
        # elements), so a larger batch is run in chunks — same numbers, one launch sequence per chunk.  (The past-state keep
        # mask and the input dropout are drawn per chunk, like any two calls.)
        max_b = self.chunk_batch(T, f64)
        if B > max_b:
            if workspace is not None or out is not None:
                raise RuntimeError("tip_amd: workspace= / out= serve a single launch sequence (batch within tip_max_batch)")
            km = None if isinstance(keep_mask, str) else keep_mask
            parts = []
            for lo in range(0, B, max_b):
                hi = min(B, lo + max_b)
                parts.append(self._forward_hip(x_imu[lo:hi], x_s[lo:hi], last_row_only,
                                               keep_mask if km is None else km[lo:hi], apply_in_dropout))
            return torch.cat(parts, dim=0)
        # (the library launches on the CURRENT device: switch only when it is another one — the context manager costs ~3 us)
        with (torch.cuda.device(dev) if torch.cuda.current_device() != dev.index else _NO_CTX):
            # A few streams (the runners' call, the latency benchmark): the kernels are queued FIRST and the ~8 us "did a parameter
            # change since the image was packed" walk runs beside the GPU; if it did, the image is re-packed and the forward queued
            # again behind the stale one (stream order: `y` is overwritten before anybody can read it).
            est = self._eval_state
            trusted = (not f64 and not self._frozen and B <= self.LAZY_STASH_MAX_BATCH and est is not None and est[0] == dev and
                       self._packed_dev is not None and self._packed_dev.device == dev)
            if not trusted and not f64 and (self._packed_dev is None or self._packed_dev.device != dev or
                                            (not self._frozen and self._packed_key != self._param_key(dev))):
                self.refresh_packed(dev)
            x_imu_c = x_imu.contiguous()
            x_s_c = x_s.contiguous()
            if apply_in_dropout and self.in_dropout > 0.0:  # :73 — a fresh nn.Dropout is always in training mode
                x_imu_c = F.dropout(x_imu_c, self.in_dropout, training=True)
            flags = 0
            mask_ptr, scale = None, 1.0
            mask = self._draw_keep_mask(x_s_c) if isinstance(keep_mask, str) else keep_mask   # :77
            if mask is not None:
                p = self.past_state_dropout
                mask = mask.to(pdt).contiguous()
                mask_ptr, scale = mask.data_ptr(), (1.0 / (1.0 - p) if p < 1.0 else 0.0)
                flags |= _lib.TIP_FWD_KEEP_MASK
            if last_row_only:
                flags |= _lib.TIP_FWD_LAST_ROW_ONLY
            shape = (B, self.size_s) if last_row_only else (B, T, self.size_s)
            if out is not None:
                if tuple(out.shape) != shape or out.dtype != pdt or out.device != dev or not out.is_contiguous():
                    raise RuntimeError(f"tip_amd: out= must be a contiguous {pdt} tensor of shape {shape} on {dev}")
                y = out
            else:
                y = torch.empty(shape, dtype=pdt, device=dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            if f64:
                params = [p.detach() for p in self.state_dict().values()]
                if any(p.dtype != torch.float64 or p.device != dev for p in params):
                    raise RuntimeError("tip_amd.TF_RNN_Past_State: fp64 forward needs every parameter in fp64 on the inputs' GPU")
                params = [p.contiguous() for p in params]
                ws = workspace if workspace is not None else self._stream_buffer(self._workspace, dev, stream, h.forward_f64_bytes(B, T))
                h.forward_f64([p.data_ptr() for p in params], x_imu_c.data_ptr(), x_s_c.data_ptr(), y.data_ptr(), B, T,
                              flags & _lib.TIP_FWD_LAST_ROW_ONLY, mask_ptr, scale, ws.data_ptr(), ws.numel(), stream)
                return y
            need = self._ws_bytes_cache.get((B, T))
            if need is None:
                need = self._ws_bytes_cache[(B, T)] = h.workspace_bytes(B, T)
            if workspace is not None:
                if workspace.dtype != torch.uint8 or workspace.device != dev or workspace.numel() < need or workspace.data_ptr() % 256:
                    raise RuntimeError(f"tip_amd: workspace= must be a 256-byte aligned uint8 tensor of >= {need} bytes on {dev}")
                ws = workspace
            else:
                ws = self._stream_buffer(self._workspace, dev, stream, need)
            try:
                h.forward(x_imu_c.data_ptr(), x_s_c.data_ptr(), y.data_ptr(), B, T, flags, mask_ptr, scale,
                          ws.data_ptr(), ws.numel(), stream)
            except _lib.TipHandoffError:
                # An EARLIER launch of this handle lost an inter-workgroup hand-off (a co-tenant held CUs): its outputs were
                # NaN-poisoned and flagged.  First time: demote the handle to the plans that need no co-residency (hybrid
                # encoder + single-workgroup recurrence tiles), clear the word and run THIS call — a co-tenant then costs
                # throughput, not every following frame.  TIP_OPT_AUTO_DEMOTE = 0 (or a second loss) reports the error.
                if self._answer_handoff(h) is None:
                    raise
                h.forward(x_imu_c.data_ptr(), x_s_c.data_ptr(), y.data_ptr(), B, T, flags, mask_ptr, scale,
                          ws.data_ptr(), ws.numel(), stream)
            if not self._frozen and B <= self.LAZY_STASH_MAX_BATCH:
                plist = self._plist()
                ptrs, vsum = list(map(_data_ptr, plist)), sum(map(_version, plist))
                if trusted and (ptrs != est[1] or vsum != est[2]):
                    self.refresh_packed(dev)                 # the parameters moved or changed under the trusted image: once more, behind it
                    self._relaunches += 1
                    h.forward(x_imu_c.data_ptr(), x_s_c.data_ptr(), y.data_ptr(), B, T, flags, mask_ptr, scale,
                              ws.data_ptr(), ws.numel(), stream)
                self._eval_state = (dev, ptrs, vsum)
        return y

    # -- torch-op composite (autograd / training) -----------------------------------------------
    def _forward_torch_ops(self, x_imu, x_s, keep_mask="draw", apply_in_dropout=True, relu_gates=None):
        """relu_gates (tests): per layer a [B,T,tf_hid_size] boolean tensor — the hidden units' gates are TAKEN from it instead of from
        the sign of the pre-activation (a unit within rounding of zero is open in one implementation and shut in another; one such
        flip moves a weight gradient by ~1e-3 relative: comparisons of gradients fix the gates)."""
        B, T = x_imu.shape[0], x_imu.shape[1]
        D, H = self.tf_in_dim, self.n_heads
        dh = D // H
        s = torch.where(x_s.isnan(), torch.zeros_like(x_s), x_s)          # :65 — NaN only: +/-inf stay, as in the reference
        keep = torch.ones(self.size_s, dtype=s.dtype, device=s.device)
        keep[18 * 6: 18 * 6 + 3] = 0.0                                  # :75
        s = s * keep
        xi = F.dropout(x_imu, self.in_dropout, training=True) if (apply_in_dropout and self.in_dropout > 0) else x_imu  # :73
        mask = self._draw_keep_mask(s) if isinstance(keep_mask, str) else keep_mask                     # :77
        if mask is not None:
            p = self.past_state_dropout
            s = s * mask * (1.0 / (1.0 - p) if p < 1.0 else 0.0)
        z = F.linear(torch.cat((xi, s), dim=2), self.in_linear.weight, self.in_linear.bias)
        z = z.reshape(B, T, H, dh).transpose(2, 3).reshape(B, T, D)      # :88-89 (batch-first view of the same shuffle)
        pdrop = self.ENCODER_DROPOUT if self.training else 0.0   # torch default inside nn.TransformerEncoderLayer
        for li, layer in enumerate(self.tf_encode.layers):
            qkv = F.linear(z, layer.self_attn.in_proj_weight, layer.self_attn.in_proj_bias)
            q, k, v = (t.reshape(B, T, H, dh).transpose(1, 2) for t in qkv.split(D, dim=2))
            a = F.scaled_dot_product_attention(q, k, v, dropout_p=pdrop, is_causal=True)
            a = a.transpose(1, 2).reshape(B, T, D)
            a = F.linear(a, layer.self_attn.out_proj.weight, layer.self_attn.out_proj.bias)
            z = F.layer_norm(z + F.dropout(a, pdrop, self.training), (D,), layer.norm1.weight, layer.norm1.bias, 1e-5)
            f = F.linear(z, layer.linear1.weight, layer.linear1.bias)
            f = F.relu(f) if relu_gates is None else f * relu_gates[li].to(f.dtype)
            f = F.linear(F.dropout(f, pdrop, self.training), layer.linear2.weight, layer.linear2.bias)
            z = F.layer_norm(z + F.dropout(f, pdrop, self.training), (D,), layer.norm2.weight, layer.norm2.bias, 1e-5)
        if self.rnn is not None:
            ih = F.linear(z, self.rnn.weight_ih_l0, self.rnn.bias_ih_l0 + self.rnn.bias_hh_l0)
            hcur = torch.zeros(B, self.rnn_hid_size, dtype=z.dtype, device=z.device)
            hs = []
            for t in range(T):
                hcur = torch.tanh(ih[:, t] + F.linear(hcur, self.rnn.weight_hh_l0))
                hs.append(hcur)
            z = torch.stack(hs, dim=1)
        return F.linear(z, self.linear.weight, self.linear.bias)


def _hip_input_grads(module, h, want_imu, want_s, bwd_inputs, pc, scratch, B, T, stream):
    """d x_imu, d x_s of the step that tip_train_backward has just differentiated on `scratch` (tip_train_input_grads)."""
    if not (want_imu or want_s):
        return None, None
    xs, mask, scale = bwd_inputs
    if isinstance(mask, int):
        mask = module._hash_keep_mask(xs, mask)
    n_imu = module.input_size_imu + (18 if module.with_acc_sum else 0)
    dxi = torch.empty((B, T, n_imu), dtype=torch.float32, device=xs.device) if want_imu else None
    dxs = torch.empty((B, T, module.size_s), dtype=torch.float32, device=xs.device) if want_s else None
    h.train_input_grads([p.data_ptr() for p in pc], xs.data_ptr(), mask.data_ptr() if mask is not None else None, scale,
                        scratch.data_ptr(), scratch.numel(), dxi.data_ptr() if want_imu else None, dxs.data_ptr() if want_s else None,
                        B, T, stream)
    return dxi, dxs


class _HipTrainFunction(torch.autograd.Function):
    """The model call of the reference training loop (train_model.py:175 forward, :192 backward) on the HIP kernels.
    forward: tip_train_forward (activations stashed in `saved`, encoder dropout from a counter-based hash of `seed`);
    backward: tip_train_backward -> gradients of the state-dict tensors, returned as views of one flat buffer."""

    @staticmethod
    def forward(ctx, module, x_imu, x_s, mask, p_drop, seed, *params):
        h = module._ensure_handle()
        dev = x_imu.device
        B, T = int(x_imu.shape[0]), int(x_imu.shape[1])
        n_imu = module.input_size_imu + (18 if module.with_acc_sum else 0)
        if x_imu.shape[2] != n_imu or x_s.shape[2] != module.size_s:
            raise RuntimeError(f"mat1 and mat2 shapes cannot be multiplied: got feature widths "
                               f"{x_imu.shape[2]}+{x_s.shape[2]}, in_linear expects {n_imu}+{module.size_s}")
        pdt = x_imu.dtype                                  # fp32, or fp64 for a module built under --double (_hip_train_ok: no mixes)
        f64 = pdt == torch.float64
        # a few windows: already on the GPU's queue (module._few_window_train_launch, called by _dispatch in front of this apply;
        # False = tried, not served that way)
        pre, module._pre_launched = module._pre_launched, None
        if pre is None:
            pre = module._few_window_train_launch(x_imu, x_s, mask, p_drop, seed)
        lazy = bool(pre)
        saved = None
        if lazy:
            y, xi, xs, lazy_mask, scale = pre
            state_seed = lazy_mask if isinstance(lazy_mask, int) else None
            mask = None if state_seed is not None else lazy_mask
        else:
            with torch.cuda.device(dev):
                xi, xs = x_imu.contiguous(), x_s.contiguous()
                pd = module.past_state_dropout
                mask_ptr, scale = None, (1.0 / (1.0 - pd) if pd < 1.0 else 0.0) if mask is not None else 1.0
                state_seed = mask if isinstance(mask, int) else None     # the keep mask as (p, seed): drawn by the library
                if state_seed is not None:
                    mask = module._hash_keep_mask(xs, state_seed)
                    mask_ptr = mask.data_ptr()
                elif mask is not None:
                    mask = mask.to(pdt).contiguous()
                    mask_ptr = mask.data_ptr()
                y = torch.empty((B, T, module.size_s), dtype=pdt, device=dev)
                stream = torch.cuda.current_stream(dev).cuda_stream
                saved_bytes, _ = h.train_bytes(B, T, fp64=f64)
                saved = torch.empty(saved_bytes, dtype=torch.uint8, device=dev)
                pc = [p.detach().contiguous() for p in params]
                h.train_forward([p.data_ptr() for p in pc], xi.data_ptr(), xs.data_ptr(), mask_ptr, scale, p_drop, seed,
                                y.data_ptr(), saved.data_ptr(), saved.numel(), B, T, stream, fp64=f64)
        ctx.module, ctx.dims, ctx.p_drop, ctx.seed, ctx.f64 = module, (B, T), p_drop, seed, f64
        ctx.saved_stash = saved
        # no stash yet: backward re-runs the forward on (xi, xs, keep mask or its seed) — the tensors go through save_for_backward, so
        # an in-place edit of the inputs between forward and backward is an autograd error, not a silently different gradient
        lazy_mask_t = mask if (lazy and state_seed is None and mask is not None) else None
        ctx.lazy = (state_seed, scale, lazy_mask_t is not None) if lazy else None
        # what gradients w.r.t. the inputs need (tip_train_input_grads): x_s (NaN positions), the keep mask (tensor, or its seed), its scale
        ctx.bwd_inputs = (xs, mask if mask is not None else state_seed, scale) if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) else None
        if module.keep_train_stash:
            module.last_train_stash = (saved, B, T)
        if lazy:
            ctx.save_for_backward(xi, xs, *([lazy_mask_t] if lazy_mask_t is not None else []), *params)
        else:
            ctx.save_for_backward(*params)
        return y

    @staticmethod
    def backward(ctx, gy):
        module, (B, T) = ctx.module, ctx.dims
        params = ctx.saved_tensors          # raises if a parameter (or a lazily kept input) was modified in place since the forward
        module._backward_seen = True        # this caller differentiates its .train()-mode calls: no more stash-free forwards (see _few_window_train_launch)
        module._fast_state = None
        lazy_in = None
        if ctx.lazy is not None:
            state_seed, lscale, has_mask_t = ctx.lazy
            n_in = 3 if has_mask_t else 2
            lazy_in = (params[0], params[1], params[2] if has_mask_t else state_seed, lscale)
            params = params[n_in:]
        h = module._ensure_handle()
        dev = gy.device
        with torch.cuda.device(dev):
            saved = ctx.saved_stash
            f64 = ctx.f64
            pdt = torch.float64 if f64 else torch.float32
            stream = torch.cuda.current_stream(dev).cuda_stream
            if saved is None and lazy_in is not None and not getattr(ctx, "lazy_done", False):
                # the forward ran without a stash (few windows, tip_forward_dropout): produce it now — tip_train_forward with the same
                # inputs, keep mask, dropout probability and seed evaluates the same function with the same keep decisions
                xi, xs, mask, scale = lazy_in
                if isinstance(mask, int):
                    mask = module._hash_keep_mask(xs, mask)
                saved_bytes, _ = h.train_bytes(B, T, fp64=f64)
                saved = torch.empty(saved_bytes, dtype=torch.uint8, device=dev)
                y2 = torch.empty((B, T, module.size_s), dtype=pdt, device=dev)
                pc0 = [p.detach().contiguous() for p in params]
                h.train_forward([p.data_ptr() for p in pc0], xi.data_ptr(), xs.data_ptr(), mask.data_ptr() if mask is not None else None,
                                scale, ctx.p_drop, ctx.seed, y2.data_ptr(), saved.data_ptr(), saved.numel(), B, T, stream, fp64=f64)
                ctx.lazy_done = True
            if saved is None:
                raise RuntimeError("tip_amd: backward through the HIP training step a second time — the activation stash "
                                   "is released after the first backward (retain_graph=True is not supported; run the "
                                   "forward again)")
            _, scratch_bytes = h.train_bytes(B, T, fp64=f64)
            scratch = module._stream_buffer(module._train_scratch, dev, stream, scratch_bytes)
            total = sum(p.numel() for p in params)
            flat = torch.empty(total, dtype=pdt, device=dev)
            pc = [p.detach().contiguous() for p in params]
            g = gy.to(pdt).contiguous()
            h.train_backward([p.data_ptr() for p in pc], g.data_ptr(), saved.data_ptr(), saved.numel(),
                             scratch.data_ptr(), scratch.numel(), flat.data_ptr(), total,
                             ctx.p_drop, ctx.seed, B, T, stream, fp64=f64)
            dxi, dxs = _hip_input_grads(module, h, ctx.needs_input_grad[1], ctx.needs_input_grad[2], ctx.bwd_inputs, pc, scratch, B, T, stream)
        ctx.saved_stash = None
        out, off = [], 0
        for i, p in enumerate(params):
            n = p.numel()
            out.append(flat[off:off + n].view(p.shape) if ctx.needs_input_grad[6 + i] else None)
            off += n
        return (None, dxi, dxs, None, None, None, *out)


class _HipForwardHipBackward(torch.autograd.Function):
    """.eval()-mode forward with autograd enabled, configurations the HIP training step covers: forward = the inference kernels
    (bit-identical to the no_grad call); backward = tip_train_forward (p_drop = 0: the same function, activations stashed) +
    tip_train_backward on the saved inputs and keep mask.  The extra forward is paid only when .backward() is really called."""

    @staticmethod
    def forward(ctx, module, last_row_only, x_imu, x_s, mask, *params):
        ctx.module, ctx.last = module, last_row_only
        ctx.has_mask = mask is not None
        # (inputs and keep mask through save_for_backward: an in-place edit between forward and backward is an autograd error)
        ctx.save_for_backward(x_imu.contiguous(), x_s.contiguous(), *([mask] if mask is not None else []), *params)
        with torch.no_grad():
            return module._forward_hip(x_imu, x_s, last_row_only, keep_mask=mask, apply_in_dropout=False)

    @staticmethod
    def backward(ctx, gy):
        module = ctx.module
        sv = ctx.saved_tensors
        xi, xs = sv[0], sv[1]
        mask = sv[2] if ctx.has_mask else None
        params = sv[3:] if ctx.has_mask else sv[2:]
        h = module._ensure_handle()
        dev = gy.device
        B, T = int(xi.shape[0]), int(xi.shape[1])
        pdt = xi.dtype
        f64 = pdt == torch.float64
        with torch.cuda.device(dev):
            stream = torch.cuda.current_stream(dev).cuda_stream
            mask_ptr, scale = None, 1.0
            if mask is not None:
                pd = module.past_state_dropout
                mask = mask.to(pdt).contiguous()
                mask_ptr, scale = mask.data_ptr(), (1.0 / (1.0 - pd) if pd < 1.0 else 0.0)
            saved_bytes, scratch_bytes = h.train_bytes(B, T, fp64=f64)
            saved = torch.empty(saved_bytes, dtype=torch.uint8, device=dev)
            y2 = torch.empty((B, T, module.size_s), dtype=pdt, device=dev)
            pc = [p.detach().contiguous() for p in params]
            ptrs = [p.data_ptr() for p in pc]
            h.train_forward(ptrs, xi.data_ptr(), xs.data_ptr(), mask_ptr, scale, 0.0, 0, y2.data_ptr(), saved.data_ptr(), saved.numel(),
                            B, T, stream, fp64=f64)
            if module.keep_train_stash:
                module.last_train_stash = (saved, B, T)
            if ctx.last:      # the cotangent of row T-1 only: zero everywhere else
                g = torch.zeros((B, T, module.size_s), dtype=pdt, device=dev)
                g[:, -1] = gy.to(pdt)
            else:
                g = gy.to(pdt).contiguous()
            scratch = module._stream_buffer(module._train_scratch, dev, stream, scratch_bytes)
            total = sum(p.numel() for p in params)
            flat = torch.empty(total, dtype=pdt, device=dev)
            h.train_backward(ptrs, g.data_ptr(), saved.data_ptr(), saved.numel(), scratch.data_ptr(), scratch.numel(), flat.data_ptr(),
                             total, 0.0, 0, B, T, stream, fp64=f64)
            dxi, dxs = _hip_input_grads(module, h, ctx.needs_input_grad[2], ctx.needs_input_grad[3], (xs, mask, scale), pc, scratch, B, T, stream)
        out, off = [], 0
        for i, p in enumerate(params):
            n = p.numel()
            out.append(flat[off:off + n].view(p.shape) if ctx.needs_input_grad[5 + i] else None)
            off += n
        return (None, None, dxi, dxs, None, *out)


class _HipForwardTorchBackward(torch.autograd.Function):
    """.eval()-mode forward on the HIP kernels with autograd still enabled (the reference's runners never enter
    no_grad): values come from libtip_hip.so; if backward is ever called, the torch-op composite is recomputed
    on the saved inputs (same keep-mask) and differentiated."""

    @staticmethod
    def forward(ctx, module, last_row_only, x_imu, x_s, mask, *params):
        ctx.module, ctx.last = module, last_row_only
        ctx.has_mask = mask is not None
        ctx.save_for_backward(x_imu, x_s, *( [mask] if mask is not None else [] ))
        with torch.no_grad():
            return module._forward_hip(x_imu, x_s, last_row_only, keep_mask=mask, apply_in_dropout=False)

    @staticmethod
    def backward(ctx, gy):
        saved = ctx.saved_tensors
        x_imu, x_s = saved[0], saved[1]
        mask = saved[2] if ctx.has_mask else None
        m = ctx.module
        with torch.enable_grad():
            xi = x_imu.detach().requires_grad_(ctx.needs_input_grad[2])
            xs = x_s.detach().requires_grad_(ctx.needs_input_grad[3])
            y = m._forward_torch_ops(xi, xs, keep_mask=mask, apply_in_dropout=False)
            if ctx.last:
                y = y[:, -1]
            params = list(m.parameters())
            wanted = [t for t in [xi, xs] + params if t.requires_grad]
            grads = torch.autograd.grad(y, wanted, gy, allow_unused=True)
        it = iter(grads)
        out = [None, None]
        for t in [xi, xs] + params:
            out.append(next(it) if t.requires_grad else None)
        out.insert(4, None)  # mask
        return tuple(out)

"""The reference's training losses on the HIP device.

Mirror of learning_utils.py:13-78 — same function names, argument order (`ra` = ground truth, `rb` = prediction), shape
asserts and masking rules — each evaluated by csrc/tip_loss.hip (one reduction pass forward, one gradient pass backward)
instead of a few dozen elementwise / boolean-index torch kernels with a host sync at every `x[mask]`.

`train_loss(y_pred, y, n_sbps)` is the three of them as train_model.py:177-189 combines them, in ONE pass over the rows:

    y_pred = model(x_imu, x_s + noise)
    loss = train_loss(y_pred, y, n_sbps=5)          # == loss_c + loss_q + loss_j of the reference loop
    loss.backward()

Device tensors only, fp32 or — under train_model.py's --double — fp64 (no mixes); there is no CPU path (TipLibraryError /
TypeError otherwise).
"""
from __future__ import annotations

import ctypes
import random

import numpy as np
import torch

from . import lib as _lib

Q, C, J = _lib.TIP_LOSS_Q, _lib.TIP_LOSS_C, _lib.TIP_LOSS_J


def _rows(t: torch.Tensor, what: str, dtype=None):
    """(tensor to keep alive, row stride in elements) of a [..., W] fp32 / fp64 device tensor whose rows are evenly spaced in
    memory (a column slice of a contiguous array qualifies); anything else is made contiguous first."""
    if not t.is_cuda or t.dtype not in (torch.float32, torch.float64) or (dtype is not None and t.dtype != dtype):
        raise TypeError(f"tip_amd.learning_utils: {what} must be a float32 (or, under --double, float64) tensor on the HIP device"
                        f"{'' if dtype is None else ' of the prediction precision ' + str(dtype)} "
                        f"(got {t.dtype} on {t.device}); the losses have no CPU path and convert nothing")
    W = t.shape[-1]
    ok = t.stride(-1) == 1 and t.stride(-2) >= W
    if ok and t.dim() == 3:
        ok = t.stride(0) == t.shape[1] * t.stride(1)
    if not ok or t.dim() not in (2, 3) or t.numel() == 0:
        t = t.contiguous()
        return t, max(W, 1)
    return t, t.stride(-2)


class _Loss(torch.autograd.Function):
    """stats = tip_loss_forward(...); backward = tip_loss_backward(...).  Returns (total, parts[3]); only total is
    differentiable (parts = loss_q, loss_c, loss_j for logging)."""

    @staticmethod
    def forward(ctx, pred, gt, B, T, n_pose, n_vel, n_sbp, terms):
        lib = _lib.load()
        dev = pred.device
        p, ldp = _rows(pred.detach(), "prediction")
        dt = p.dtype
        if gt is not None:
            g, ldg = _rows(gt.detach(), "ground truth", dt)
        else:
            g, ldg = None, 0
        stats = torch.empty(_lib.TIP_LOSS_STATS, dtype=dt, device=dev)
        nbytes = ctypes.c_size_t()
        rc = lib.tip_loss_ws_bytes(B, T, ctypes.byref(nbytes))
        if rc < 0:
            raise _lib.TipStatusError(rc, "tip_loss_ws_bytes")
        ws = torch.empty(max(nbytes.value // 8, 1), dtype=torch.float64, device=dev)
        with torch.cuda.device(dev):
            rc = (lib.tip_loss_forward_f64 if dt == torch.float64 else lib.tip_loss_forward)(p.data_ptr(), ldp, g.data_ptr() if g is not None else None, ldg, B, T, n_pose, n_vel,
                                      n_sbp, terms, stats.data_ptr(), ws.data_ptr(), nbytes.value,
                                      torch.cuda.current_stream(dev).cuda_stream)
        if rc < 0:
            raise _lib.TipStatusError(rc, "tip_loss_forward")
        ctx.save_for_backward(p, g, stats)
        ctx.args = (ldp, ldg, B, T, n_pose, n_vel, n_sbp, terms, tuple(pred.shape))
        total, parts = stats[0].clone(), stats[1:4].clone()
        ctx.mark_non_differentiable(parts)
        return total, parts

    @staticmethod
    def backward(ctx, g_total, _g_parts):
        p, g, stats = ctx.saved_tensors
        ldp, ldg, B, T, n_pose, n_vel, n_sbp, terms, shape = ctx.args
        dev = p.device
        W = n_pose + n_vel + 4 * n_sbp
        dpred = torch.empty(shape, dtype=p.dtype, device=dev)
        go = g_total.detach().to(device=dev, dtype=p.dtype).reshape(1).contiguous()
        with torch.cuda.device(dev):
            lib = _lib.load()
            rc = (lib.tip_loss_backward_f64 if p.dtype == torch.float64 else lib.tip_loss_backward)(p.data_ptr(), ldp, g.data_ptr() if g is not None else None, ldg, B, T, n_pose,
                                               n_vel, n_sbp, terms, stats.data_ptr(), go.data_ptr(), dpred.data_ptr(), W,
                                               torch.cuda.current_stream(dev).cuda_stream)
        if rc < 0:
            raise _lib.TipStatusError(rc, "tip_loss_backward")
        return dpred, None, None, None, None, None, None, None


def loss_constr_multi(ra: torch.Tensor, rb: torch.Tensor) -> torch.Tensor:
    """learning_utils.py:13-35.  ra, rb: (bs, 4*N); rb is the model prediction, ra is GT."""
    assert ra.size() == rb.size()
    assert (ra.size()[1] // 4) * 4 == ra.size()[1]
    return _Loss.apply(rb, ra, int(ra.size()[0]), 1, 0, 0, int(ra.size()[1]) // 4, C)[0]


def loss_jerk(rb: torch.Tensor) -> torch.Tensor:
    """learning_utils.py:38-47.  rb: (bs, t, 18*6), the model prediction."""
    assert rb.size()[-1] == 18 * 6
    return _Loss.apply(rb, None, int(rb.size()[0]), int(rb.size()[1]), 18 * 6, 0, 0, J)[0]


def loss_q_only_2axis(ra: torch.Tensor, rb: torch.Tensor) -> torch.Tensor:
    """learning_utils.py:50-78.  ra, rb: (bs, 18*6 + 3); rb is the model prediction, ra is GT."""
    assert ra.size() == rb.size()
    assert ra.size()[1] == 18 * 6 + 3
    return _Loss.apply(rb, ra, int(ra.size()[0]), 1, 18 * 6, 3, 0, Q)[0]


def train_loss(y_pred: torch.Tensor, y: torch.Tensor, n_sbps: int = 5, with_jerk: bool = True, return_parts: bool = False):
    """loss_c + loss_q (+ loss_j) exactly as the reference's training loop forms it (train_model.py:177-189) from
    y_pred, y: (bs, t, 18*6 + 3 + 4*n_sbps).  With return_parts also the detached [loss_q, loss_c, loss_j]."""
    assert y_pred.size() == y.size() and y_pred.dim() == 3
    n_pose = int(y_pred.size()[-1]) - 3 - 4 * n_sbps
    assert n_pose > 0
    total, parts = _Loss.apply(y_pred, y, int(y_pred.size()[0]), int(y_pred.size()[1]), n_pose, 3, n_sbps,
                               Q | C | (J if with_jerk else 0))
    return (total, parts) if return_parts else total


def set_seed(seed):
    """learning_utils.py:81-85: seeds python `random` (TrainSubDataset's window sampling), numpy (the combiner's bias draw)
    and torch's CPU and device generators (parameter init, dropout masks, train_model.py:172's input noise)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def top_k_logits(logits, k):
    """Rows of `logits` with everything below the row's k-th largest value set to -inf (the behaviour of learning_utils.py:88-92;
    unused by the training / inference scripts, kept so that `from learning_utils import *` is whole)."""
    kth = torch.topk(logits, k, dim=-1).values[..., -1:]
    return torch.where(logits < kth, torch.full_like(logits, float("-inf")), logits)

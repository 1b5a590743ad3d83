"""TEST INFRASTRUCTURE — CPU restatement of the model part of one reference training step
(/root/reference/train_model.py:171-196: y = model(x_imu, x_s) in train mode, loss.backward()), SURVEY.md section 8
rows a14 / f-2.  Only tests/ may import this.  It is the checker for tip_train_forward / tip_train_backward.

Forward = the same math as oracle/tip_oracle.c (simple_transformer_with_state.py:60-102) written with torch fp64 CPU
tensors so that torch.autograd supplies the gradients the reference's `loss.backward()` computes; the four dropout
sites of nn.TransformerEncoderLayer (torch transformer.py: attention probabilities, dropout1, dropout, dropout2) take
EXPLICIT keep masks, rebuilt here from the counter-based hash that include/tip_hip.h documents, so the HIP kernels can
be compared element for element with dropout live.

Pinned against the reference itself: tests/golden/make_train_golden.py runs the real TF_RNN_Past_State in train mode
(encoder dropout switched off on the instance) and records y and per-tensor gradient digests; test_train_oracle.py checks
this file against them.
"""
from __future__ import annotations

import numpy as np
import torch

GOLDEN = np.uint64(0x9E3779B97F4A7C15)
M1 = np.uint64(0xBF58476D1CE4E5B9)
M2 = np.uint64(0x94D049BB133111EB)


def drop_key(seed: int, site: int) -> np.uint32:
    """32-bit key of one dropout site: hi32 of the splitmix64 mix of seed + GOLDEN * (site + 1)."""
    with np.errstate(over="ignore"):
        z = np.uint64(seed) + GOLDEN * (np.uint64(site) + np.uint64(1))
        z = (z ^ (z >> np.uint64(30))) * M1
        z = (z ^ (z >> np.uint64(27))) * M2
        z = z ^ (z >> np.uint64(31))
    return np.uint32(z >> np.uint64(32))


def drop_scale(seed: int, site: int, n: int, p: float) -> np.ndarray:
    """keep/(1-p) factor of the n elements of dropout site `site` (include/tip_hip.h, tip_train_forward):
    keep(idx) = lowbias32((idx mod 2^32) * 0x9E3779B1 + key(seed, site)) >= floor(p * 2^32)."""
    if p <= 0.0:
        return np.ones(n, dtype=np.float32)
    with np.errstate(over="ignore"):
        z = np.arange(n, dtype=np.uint64).astype(np.uint32) * np.uint32(0x9E3779B1) + drop_key(seed, site)
        z ^= z >> np.uint32(16)
        z *= np.uint32(0x7FEB352D)
        z ^= z >> np.uint32(15)
        z *= np.uint32(0x846CA68B)
        z ^= z >> np.uint32(16)
    thresh = min(int(p * 4294967296.0), 4294967295)
    thresh = max(thresh, 1)
    keep = z.astype(np.uint64) >= np.uint64(thresh)
    return (keep.astype(np.float32) * np.float32(1.0 / (1.0 - np.float32(p)))).astype(np.float32)


def forward(cfg: dict, params: "dict[str, torch.Tensor]", x_imu, x_s, keep_mask=None, keep_scale=1.0, p_drop=0.0, seed=0,
            dtype=torch.float64, relu_gates=None):
    """y [B,T,size_s] with autograd attached to `params` (fp64 leaf tensors).  x_imu/x_s: numpy fp32.

    relu_gates: optional list of L boolean arrays [B,T,F].  ReLU makes the model piecewise linear; a hidden unit whose
    pre-activation is within fp32 rounding of zero may be open in one implementation and closed in another, and either
    choice is a valid (sub)gradient.  When the gates of the implementation under test are passed, the oracle
    differentiates the SAME linear piece, so the comparison is not polluted by such flips."""
    D, H, L = cfg["tf_in_dim"], cfg["n_heads"], cfg["tf_layers"]
    F_, R = cfg["tf_hid_size"], cfg["rnn_hid_size"]
    dh = D // H
    xi = torch.tensor(np.asarray(x_imu), dtype=dtype)
    s = torch.tensor(np.nan_to_num(np.asarray(x_s), nan=0.0), dtype=dtype)           # :65
    B, T = xi.shape[0], xi.shape[1]
    M = B * T
    s[..., 18 * 6:18 * 6 + 3] = 0.0                                                    # :75
    if keep_mask is not None:
        s = s * torch.tensor(np.asarray(keep_mask), dtype=dtype) * keep_scale         # :77
    z = torch.nn.functional.linear(torch.cat((xi, s), dim=2), params["in_linear.weight"], params["in_linear.bias"])  # :79
    z = z.reshape(B, T, H, dh).transpose(2, 3).reshape(B, T, D)                       # :88-89
    causal = torch.triu(torch.full((T, T), float("-inf"), dtype=dtype), diagonal=1)   # :56-58

    def site(l, k, shape):
        n = int(np.prod(shape))
        return torch.tensor(drop_scale(seed, 4 * l + k, n, p_drop).reshape(shape), dtype=dtype)

    for l in range(L):
        p = f"tf_encode.layers.{l}."
        qkv = torch.nn.functional.linear(z, params[p + "self_attn.in_proj_weight"], params[p + "self_attn.in_proj_bias"])
        q, k, v = (t.reshape(B, T, H, dh).transpose(1, 2) for t in qkv.split(D, dim=2))
        sc = (q * (1.0 / np.sqrt(dh))) @ k.transpose(-1, -2) + causal
        pr = torch.softmax(sc, dim=-1) * site(l, 0, (B, H, T, T))
        a = (pr @ v).transpose(1, 2).reshape(B, T, D)
        a = torch.nn.functional.linear(a, params[p + "self_attn.out_proj.weight"], params[p + "self_attn.out_proj.bias"])
        z = torch.nn.functional.layer_norm(z + a * site(l, 1, (B, T, D)), (D,), params[p + "norm1.weight"],
                                           params[p + "norm1.bias"], 1e-5)
        f = torch.nn.functional.linear(z, params[p + "linear1.weight"], params[p + "linear1.bias"])
        f = torch.relu(f) if relu_gates is None else f * torch.tensor(np.asarray(relu_gates[l]).reshape(B, T, F_), dtype=dtype)
        f = torch.nn.functional.linear(f * site(l, 2, (B, T, F_)), params[p + "linear2.weight"], params[p + "linear2.bias"])
        z = torch.nn.functional.layer_norm(z + f * site(l, 3, (B, T, D)), (D,), params[p + "norm2.weight"],
                                           params[p + "norm2.bias"], 1e-5)
    if not cfg.get("with_rnn", True):                                                  # :43-46: no RNN, the projection reads the encoder
        return torch.nn.functional.linear(z, params["linear.weight"], params["linear.bias"])
    ih = torch.nn.functional.linear(z, params["rnn.weight_ih_l0"], params["rnn.bias_ih_l0"] + params["rnn.bias_hh_l0"])
    h = torch.zeros(B, R, dtype=dtype)
    hs = []
    for t in range(T):                                                                # :98-99
        h = torch.tanh(ih[:, t] + torch.nn.functional.linear(h, params["rnn.weight_hh_l0"]))
        hs.append(h)
    return torch.nn.functional.linear(torch.stack(hs, dim=1), params["linear.weight"], params["linear.bias"])   # :102


def step(cfg, weights: "dict[str, np.ndarray]", x_imu, x_s, cot, keep_mask=None, keep_scale=1.0, p_drop=0.0, seed=0,
         relu_gates=None):
    """(y, grads) of loss = sum(y * cot): what `loss.backward()` leaves in .grad for that loss."""
    params = {k: torch.tensor(np.asarray(v), dtype=torch.float64, requires_grad=True) for k, v in weights.items()}
    y = forward(cfg, params, x_imu, x_s, keep_mask, keep_scale, p_drop, seed, relu_gates=relu_gates)
    (y * torch.tensor(np.asarray(cot), dtype=torch.float64)).sum().backward()
    return y.detach().numpy(), {k: v.grad.numpy() for k, v in params.items()}


def digest(name: str, g: np.ndarray) -> np.ndarray:
    """Small fixed-size summary of one gradient tensor: [sum, sum of squares, projection on a fixed pseudo-random
    direction, first 5 entries]."""
    g = np.asarray(g, dtype=np.float64).reshape(-1)
    rng = np.random.RandomState(abs(hash_name(name)) % (2 ** 31))
    r = rng.standard_normal(g.size)
    return np.concatenate([[g.sum(), (g * g).sum(), (g * r).sum()], g[:5]])


def hash_name(name: str) -> int:
    v = 2166136261
    for ch in name.encode():
        v = ((v ^ ch) * 16777619) & 0xFFFFFFFF
    return v

#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REFERENCE module.

Runs only in the build container (needs /root/reference; the GPU box never has it).  It imports
/root/reference/simple_transformer_with_state.py as-is, loads the build's deterministic synthetic
weights (transformer-inertial-poser_amd/synth.py) through load_state_dict, and records
inputs + outputs (+ intermediate taps via forward hooks).  Only data is written: no reference
source travels.

Dropout: the reference applies a FRESH nn.Dropout(past_state_dropout) on every call
(simple_transformer_with_state.py:77), stochastic even in .eval().  Goldens use
past_state_dropout=0.0, dropout=0.0, .eval().  One extra case pins the keep-mask semantics
(x * mask / (1 - p)) by substituting a deterministic mask for torch's Bernoulli draw.

A second file, tip_cond_golden.npz (`--cond`), holds the CONDITIONING SWEEP: the paper configuration with the synthetic
weights scaled out of the random-init regime (`synth.make_weights(gain=g, ln_gamma=s)`: larger attention logits, a
saturating tanh recurrence, LayerNorm gamma up to x3), fp32 and fp64 reference outputs of the same two windows, plus a
second fp32 run of the reference in a different batch shape / thread count (`y32_alt`) so that the reference's OWN fp32
rounding noise |y32 - y64| is known per case: the HIP path is held to a small multiple of it (tests/test_hip_parity.py).

usage: python tests/golden/make_golden.py            # tip_forward_golden.npz
       python tests/golden/make_golden.py --cond     # tip_cond_golden.npz
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import tip_amd  # noqa: E402
from tip_amd import synth  # noqa: E402
from simple_transformer_with_state import TF_RNN_Past_State  # noqa: E402  (the reference)


def build_ref(cfg, weights, dtype, p_state=0.0):
    torch.set_default_dtype(dtype)   # reference's `hidden`/`mask` use the default dtype (:58,:98)
    m = TF_RNN_Past_State(cfg["input_size_imu"], cfg["size_s"], rnn_hid_size=cfg["rnn_hid_size"],
                          tf_hid_size=cfg["tf_hid_size"], tf_in_dim=cfg["tf_in_dim"], n_heads=cfg["n_heads"],
                          tf_layers=cfg["tf_layers"], dropout=0.0, in_dropout=0.0, past_state_dropout=p_state,
                          with_rnn=cfg.get("with_rnn", True), with_acc_sum=cfg.get("with_acc_sum", False))
    sd = {k: torch.tensor(v).to(dtype) for k, v in weights.items()}
    assert list(sd.keys()) == list(m.state_dict().keys()), "state-dict order/keys differ from the reference"
    m.load_state_dict(sd)
    m.eval()
    return m


def run_ref(m, x_imu, x_s, dtype, want_taps):
    taps = {}
    hooks = []
    if want_taps:
        hooks.append(m.in_linear.register_forward_hook(lambda mod, i, o: taps.__setitem__("in_linear_raw", o.detach())))
        for l, layer in enumerate(m.tf_encode.layers):
            hooks.append(layer.register_forward_hook(
                lambda mod, i, o, l=l: taps.__setitem__(f"layer{l}", o.detach().transpose(0, 1))))
        if m.rnn is not None:
            hooks.append(m.rnn.register_forward_hook(lambda mod, i, o: taps.__setitem__("rnn", o[0].detach())))
    with torch.no_grad():
        y = m(torch.tensor(x_imu).to(dtype), torch.tensor(x_s).to(dtype))
    for h in hooks:
        h.remove()
    return y.numpy(), {k: v.numpy() for k, v in taps.items()}


def weights_checksum(weights):
    s = 0.0
    s2 = 0.0
    for v in weights.values():
        v64 = v.astype(np.float64)
        s += float(v64.sum())
        s2 += float((v64 * v64).sum())
    return np.array([s, s2])


def main():
    out = {}
    cases = []
    for seed in (0, 1):
        for (B, T) in ((1, 1), (1, 7), (2, 40), (3, 39)):
            cases.append(("paper", synth.PAPER, seed, B, T))
    cases.append(("tiny", synth.TINY, 0, 2, 80))
    cases.append(("tiny", synth.TINY, 1, 5, 13))
    norn = dict(synth.TINY, with_rnn=False)
    cases.append(("tiny_nornn", norn, 0, 2, 9))
    noacc = dict(synth.TINY, with_acc_sum=False)
    cases.append(("tiny_noacc", noacc, 0, 2, 11))

    wcache = {}
    for (name, cfg, seed, B, T) in cases:
        key = (name, seed)
        if key not in wcache:
            wcache[key] = synth.make_weights(cfg, seed=seed)
        w = wcache[key]
        x_imu, x_s = synth.make_inputs(cfg, B, T, seed=1234 + seed)
        m32 = build_ref(cfg, w, torch.float32)
        want_taps = (B, T) in ((1, 7), (2, 40)) and seed == 0 or name != "paper"
        y32, taps = run_ref(m32, x_imu, x_s, torch.float32, want_taps)
        # inputs must not be mutated by forward (reference clones, :63-64)
        m64 = build_ref(cfg, w, torch.float64)
        y64, _ = run_ref(m64, x_imu.astype(np.float64), x_s.astype(np.float64), torch.float64, False)
        tag = f"{name}_s{seed}_B{B}_T{T}"
        out[tag + "/x_imu"] = x_imu
        out[tag + "/x_s"] = x_s
        out[tag + "/y32"] = y32.astype(np.float32)
        out[tag + "/y64"] = y64
        out[tag + "/wsum"] = weights_checksum(w)
        for k, v in taps.items():
            out[tag + "/tap_" + k] = v.astype(np.float32)
        print(f"{tag}: |y32-y64|max = {np.abs(y32 - y64).max():.3e}  |y|max = {np.abs(y64).max():.3f}")

    # keep-mask semantics (p = 0.8): substitute the Bernoulli draw with a deterministic mask.
    cfg, seed, B, T, p = synth.PAPER, 0, 2, 40, 0.8
    w = wcache[("paper", 0)]
    x_imu, x_s = synth.make_inputs(cfg, B, T, seed=1234)
    mask = synth.make_keep_mask(cfg, B, T, p, seed=7)

    class _FixedDropout(torch.nn.Module):
        def __init__(self, p_):
            super().__init__()
            self.p_ = p_

        def forward(self, x):
            if self.p_ == 0.0:
                return x
            return x * torch.tensor(mask).to(x.dtype) / (1.0 - self.p_)

    # statistical check that torch's own dropout uses the same scaling: E[dropout(x)] ~= x, kept values == x/(1-p)
    torch.set_default_dtype(torch.float32)
    probe = torch.ones(1000)
    d = torch.nn.Dropout(p)(probe)
    kept = d[d != 0]
    assert torch.allclose(kept, torch.full_like(kept, 1.0 / (1.0 - p)))
    orig = torch.nn.Dropout
    try:
        torch.nn.Dropout = _FixedDropout
        m32 = build_ref(cfg, w, torch.float32, p_state=p)
        y32, _ = run_ref(m32, x_imu, x_s, torch.float32, False)
        m64 = build_ref(cfg, w, torch.float64, p_state=p)
        y64, _ = run_ref(m64, x_imu.astype(np.float64), x_s.astype(np.float64), torch.float64, False)
    finally:
        torch.nn.Dropout = orig
        torch.set_default_dtype(torch.float32)
    tag = "paper_mask_s0_B2_T40"
    out[tag + "/x_imu"], out[tag + "/x_s"], out[tag + "/mask"] = x_imu, x_s, mask
    out[tag + "/p"] = np.array([p])
    out[tag + "/y32"], out[tag + "/y64"] = y32.astype(np.float32), y64
    out[tag + "/wsum"] = weights_checksum(w)
    print(f"{tag}: |y32-y64|max = {np.abs(y32 - y64).max():.3e}")

    path = os.path.join(HERE, "tip_forward_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


COND_GAINS = (1.0, 1.5, 2.0, 3.0, 4.0)
COND_LN = (1.0, 3.0)


def _cond_case(out, tag, cfg, w, x_imu, x_s, B):
    m32 = build_ref(cfg, w, torch.float32)
    y32, _ = run_ref(m32, x_imu, x_s, torch.float32, False)
    # the same windows one at a time on one thread: other GEMM blocking, other rounding, same arithmetic
    nthr = torch.get_num_threads()
    torch.set_num_threads(1)
    y32_alt = np.concatenate([run_ref(m32, x_imu[b:b + 1], x_s[b:b + 1], torch.float32, False)[0] for b in range(B)], axis=0)
    torch.set_num_threads(nthr)
    m64 = build_ref(cfg, w, torch.float64)
    y64, _ = run_ref(m64, x_imu.astype(np.float64), x_s.astype(np.float64), torch.float64, False)
    out[tag + "/y32"] = y32.astype(np.float32)
    out[tag + "/y32_alt"] = y32_alt.astype(np.float32)
    out[tag + "/y64"] = y64
    out[tag + "/wsum"] = weights_checksum(w)
    n1, n2 = np.abs(y32 - y64).max(), np.abs(y32_alt - y64).max()
    print(f"{tag}: reference fp32 noise {n1:.3e} / {n2:.3e} (alt shape)   |y|max = {np.abs(y64).max():.3f}")


def cond_main():
    """Conditioning sweep (reference: /root/reference/simple_transformer_with_state.py:60-102 run as-is)."""
    cfg, seed, B, T = synth.PAPER, 0, 2, 40
    x_imu, x_s = synth.make_inputs(cfg, B, T, seed=1234)
    out = {"x_imu": x_imu, "x_s": x_s}
    for g in COND_GAINS:
        for lg in COND_LN:
            if g == 1.0 and lg == 1.0:
                continue                       # the random-init regime is tip_forward_golden.npz
            w = synth.make_weights(cfg, seed=seed, gain=g, ln_gamma=lg)
            tag = f"g{g:g}_ln{lg:g}"
            _cond_case(out, tag, cfg, w, x_imu, x_s, B)
            out[tag + "/gain_ln"] = np.array([g, lg])
    # other widths (the general plan's GEMM / attention kernels: d_head 64 and 32, rnn 192, T = 80): inputs are regenerated by the
    # tests from the seeds, tag = <config>__g<gain>_ln<gamma>
    for name, c2, B2, T2, g, lg in (("scaled2", dict(synth.SCALED, tf_layers=2), 2, 80, 2.0, 1.0),
                                    ("scaled2", dict(synth.SCALED, tf_layers=2), 2, 80, 3.0, 3.0),
                                    ("tiny", synth.TINY, 3, 33, 3.0, 1.0), ("tiny", synth.TINY, 3, 33, 4.0, 3.0)):
        xi2, xs2 = synth.make_inputs(c2, B2, T2, seed=4321)
        w = synth.make_weights(c2, seed=seed, gain=g, ln_gamma=lg)
        tag = f"{name}__g{g:g}_ln{lg:g}"
        _cond_case(out, tag, c2, w, xi2, xs2, B2)
        out[tag + "/gain_ln"] = np.array([g, lg])
        out[tag + "/shape"] = np.array([B2, T2])
    torch.set_default_dtype(torch.float32)
    path = os.path.join(HERE, "tip_cond_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if "--cond" in sys.argv[1:]:
        cond_main()
    else:
        main()

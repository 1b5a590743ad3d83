#!/usr/bin/env python3
"""Golden digests of the REFERENCE model's training-step gradients.

Runs only in the build container: imports /root/reference/simple_transformer_with_state.py, loads the build's
synthetic weights, puts the module in train() mode as train_model.py:160 does, and — because torch's dropout RNG cannot
be reproduced outside torch — switches the encoder's four dropout sites off ON THE INSTANCE (layer.dropout*.p = 0,
self_attn.dropout = 0).  Then y = model(x_imu, x_s); (y * cot).sum().backward() (train_model.py:175,192 with a linear
stand-in loss so that dL/dy = cot is known exactly).

Written: inputs, cot, y, and for each of the 56 state-dict tensors an 8-number digest of its gradient
(oracle/train_oracle.py:digest).  Data only.

usage: python tests/golden/make_train_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")


def main():
    import tip_amd  # noqa: F401
    from tip_amd import synth
    from oracle import train_oracle
    from simple_transformer_with_state import TF_RNN_Past_State

    cfg = synth.PAPER
    out = {}
    for tag, (B, T, seed) in {"train_s0_B2_T40": (2, 40, 0), "train_s1_B3_T17": (3, 17, 1)}.items():
        w = synth.make_weights(cfg, seed=seed)
        torch.manual_seed(0)
        model = TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                                  dropout=0.0, in_dropout=0.0, past_state_dropout=0.0, with_acc_sum=True)
        model.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
        model.train()
        for layer in model.tf_encode.layers:
            layer.dropout.p = 0.0
            layer.dropout1.p = 0.0
            layer.dropout2.p = 0.0
            layer.self_attn.dropout = 0.0
        x_imu, x_s = synth.make_inputs(cfg, B, T, seed=500 + seed)
        cot = synth.normal(900 + seed, "train/cot", B * T * cfg["size_s"]).reshape(B, T, cfg["size_s"]).astype(np.float32)
        y = model(torch.tensor(x_imu), torch.tensor(x_s))
        (y * torch.tensor(cot)).sum().backward()
        out[tag + "/x_imu"], out[tag + "/x_s"], out[tag + "/cot"] = x_imu, x_s, cot
        out[tag + "/y"] = y.detach().numpy()
        names = list(model.state_dict().keys())
        grads = dict(model.named_parameters())
        out[tag + "/digests"] = np.stack([train_oracle.digest(n, grads[n].grad.numpy()) for n in names])
        out[tag + "/gnorm"] = np.array([float(np.sqrt(sum((grads[n].grad.double() ** 2).sum() for n in names)))])
        print(tag, "y", y.shape, "grad norm", out[tag + "/gnorm"][0])
    path = os.path.join(HERE, "tip_train_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

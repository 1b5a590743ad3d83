"""TEST INFRASTRUCTURE / CPU BASELINE — the model built from STOCK torch.nn modules (PyTorch's nn.Linear,
nn.TransformerEncoder(Layer), nn.RNN: the dispatch the reference takes on a CPU,
/root/reference/simple_transformer_with_state.py:22-42), with the data flow of its forward (:60-102) restated by the build.
Only tests/ and bench.py's cpu_baseline leg may import this; the product path never does.

Attribute names equal the reference's, so a state dict in the reference's layout (synth.make_weights) loads directly.
Pinned against the golden vectors captured from the reference in tests/test_host_cpu.py.
"""
from __future__ import annotations

import torch
from torch import nn


class StockTIP(nn.Module):
    def __init__(self, cfg: dict, past_state_keep=None):
        super().__init__()
        n_imu = cfg["input_size_imu"] + (18 if cfg.get("with_acc_sum", False) else 0)
        D, S = cfg["tf_in_dim"], cfg["size_s"]
        self.heads = cfg["n_heads"]
        self.in_linear = nn.Linear(n_imu + S, D)
        # post-norm, ReLU, sequence-first: torch's defaults, as the reference leaves them; dropout 0 so eval == train
        layer = nn.TransformerEncoderLayer(d_model=D, nhead=self.heads, dim_feedforward=cfg["tf_hid_size"], dropout=0.0)
        self.tf_encode = nn.TransformerEncoder(layer, num_layers=cfg["tf_layers"], enable_nested_tensor=False)
        self.with_rnn = bool(cfg.get("with_rnn", True))
        if self.with_rnn:
            self.rnn = nn.RNN(input_size=D, hidden_size=cfg["rnn_hid_size"], num_layers=1, nonlinearity="tanh",
                              batch_first=True)
            self.linear = nn.Linear(cfg["rnn_hid_size"], S)
        else:
            self.linear = nn.Linear(D, S)

    def forward(self, x_imu: torch.Tensor, x_s: torch.Tensor, keep_mask=None, keep_scale: float = 1.0) -> torch.Tensor:
        B, T = x_imu.shape[0], x_imu.shape[1]
        state = torch.where(torch.isnan(x_s), torch.zeros_like(x_s), x_s)        # NaN history entries count as zero
        state = torch.cat((state[..., :108], torch.zeros_like(state[..., 108:111]), state[..., 111:]), dim=-1)
        if keep_mask is not None:                                               # explicit past-state dropout draw
            state = state * keep_mask * keep_scale
        z = self.in_linear(torch.cat((x_imu, state), dim=-1)).transpose(0, 1)   # [T, B, D]
        D = z.shape[-1]
        z = z.reshape(T, B, self.heads, D // self.heads).transpose(2, 3).reshape(T, B, D)   # head-interleave shuffle
        causal = torch.full((T, T), float("-inf"), dtype=z.dtype, device=z.device).triu(1)
        z = self.tf_encode(z, causal).transpose(0, 1)                           # back to [B, T, D]
        if self.with_rnn:
            h0 = torch.zeros(1, B, self.rnn.hidden_size, dtype=z.dtype, device=z.device)
            z, _ = self.rnn(z, h0)
        return self.linear(z)


def build(cfg: dict, weights: dict) -> StockTIP:
    m = StockTIP(cfg)
    m.load_state_dict({k: torch.as_tensor(v) for k, v in weights.items()})
    return m.eval()

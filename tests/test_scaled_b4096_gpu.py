"""GPU: BASELINE.json configs[4] at ITS batch — scaled model (12 layers, d=1024, ffn=4096, 16 heads of 64, T=80), B = 4096
windows — on one GPU (reference function: simple_transformer_with_state.py:60-102).  The 8-GPU run of the config gives every
rank 512 windows; here the same 4096 distinct windows go through one GPU twice:

  * as the 8 x 512 shards the ranks would run, one call each;
  * as ONE module call, which the host splits into chunks of at most tip_max_batch windows (32-bit buffer offsets; 1536 at
    T = 80 for these widths);

and the two must agree bit for bit (a window's result does not depend on its batch neighbours or on where a shard boundary
falls).  Every one of the 4096 windows is then compared with tip_forward_f64 — the same function in IEEE double on the same
fp32-valued weights and windows — and that fp64 path is anchored to the CPU fp64 oracle on two windows of different shards."""
import numpy as np
import pytest
import torch

from tip_amd import synth
from oracle import oracle
from test_host_cpu import make_model, load_synth

pytestmark = pytest.mark.gpu
TOL32 = 1e-4          # north_star's bound on fp32 outputs
TOL64 = 1e-10


def test_config5_b4096_one_gpu_shards_chunks_and_fp64():
    cfg = synth.SCALED
    B, T, SH = 4096, 80, 512
    m32 = make_model(cfg)
    w = load_synth(m32, cfg, 0)
    m32 = m32.cuda().eval()
    assert m32.chunk_batch(T) == 1536 and m32._ensure_handle().max_batch(T) == 1638
    x_imu, x_s = synth.make_inputs(cfg, B, T, seed=4096, nan_frac=0.002)
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    with torch.no_grad():
        n0 = m32.hip_forward_count()
        y_full = m32(xi, xs)                                             # one call: 1536 + 1536 + 1024
        torch.cuda.synchronize()
        assert m32.hip_forward_count() == n0 + 3, "B = 4096 did not run as three chunked launch sequences"
        y_sh = torch.cat([m32(xi[k:k + SH], xs[k:k + SH]) for k in range(0, B, SH)])   # what the 8 ranks would compute
        torch.cuda.synchronize()
        assert m32.hip_forward_count() == n0 + 3 + B // SH
        assert y_full.shape == (B, T, cfg["size_s"]) and bool(torch.isfinite(y_full).all())
        assert torch.equal(y_full, y_sh), "shard outputs differ from the one-call (chunked) outputs"
        yl = m32.forward_last(xi, xs)                                    # the streaming form, chunked the same way
        assert torch.equal(yl, y_full[:, -1])
    # every window against the on-device fp64 forward (512 windows per call: 3.4 GB of fp64 workspace)
    m64 = make_model(cfg)
    load_synth(m64, cfg, 0)
    m64 = m64.double().cuda().eval()
    worst, y64_keep = 0.0, {}
    with torch.no_grad():
        for k in range(0, B, SH):
            y64 = m64(xi[k:k + SH].double(), xs[k:k + SH].double())
            worst = max(worst, float((y_full[k:k + SH].double() - y64).abs().max()))
            if k in (0, B - SH):
                y64_keep[k] = y64[[0, SH - 1]].cpu().numpy()
    assert worst < TOL32, worst
    # ... whose own anchor is the CPU fp64 oracle: first window of shard 0, last window of shard 7
    sel = np.array([0, B - 1])
    yo = oracle.forward(cfg, w, x_imu[sel], x_s[sel], dtype=np.float64)
    got = np.stack([y64_keep[0][0], y64_keep[B - SH][1]])
    assert np.abs(got - yo).max() < TOL64, np.abs(got - yo).max()
    assert np.abs(y_full[sel.tolist()].cpu().numpy() - yo).max() < TOL32

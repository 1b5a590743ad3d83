#!/usr/bin/env python3
"""Measurement: the recurrence + head at B=256 (and other B) per RNN cluster size, from the library's stage timers, plus a
bit-exact comparison of every variant with cluster 1.  usage: python tools/rnn_variants.py [B ...]"""
import contextlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
os.environ.setdefault("TIP_LIB", "measure")   # the launchers' TIP_* switches exist in the measurement build only (csrc: make measure)
import tip_amd
from tip_amd import synth
cfg = synth.PAPER
with contextlib.redirect_stdout(sys.stderr):
    m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                                  dropout=0.0, in_dropout=0.0, past_state_dropout=0.0, with_acc_sum=True)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
m = m.cuda().eval()
Bs = [int(a) for a in sys.argv[1:]] or [256]
for B in Bs:
    x_imu, x_s = synth.make_inputs(cfg, min(B, 256), 40)
    xi = torch.tensor(np.tile(x_imu, ((B + 255) // 256, 1, 1))[:B]).cuda()
    xs = torch.tensor(np.tile(x_s, ((B + 255) // 256, 1, 1))[:B]).cuda()
    ref = None
    for cl in (1, 4, 8, 16, 32, 0):
        try:
            m.set_plan("auto", rnn_cluster=cl, profile=0)
            with torch.no_grad():
                y = m(xi, xs)
                torch.cuda.synchronize()
                if ref is None:
                    ref = y.clone()
                same = bool(torch.equal(y, ref))
                for _ in range(10):
                    m(xi, xs)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    m(xi, xs)
                e1.record(); e1.synchronize()
                step = e0.elapsed_time(e1) / 50
                m.set_plan("auto", rnn_cluster=cl, profile=1)
                for _ in range(20):
                    m(xi, xs)
                torch.cuda.synchronize()
                st = {n: ms / k for n, ms, k in m.profile_read()}
            print(f"B={B} cluster={cl:2d}: step {step:.4f} ms | rnn {st.get('rnn_recurrence', 0)*1e3:7.1f} us  head {st.get('out_linear', 0)*1e3:6.1f} us  "
                  f"encoder {st.get('fused_encoder', 0)*1e3:7.1f} us | bit-identical to cluster 1: {same}", flush=True)
        except Exception as e:
            print(f"B={B} cluster={cl}: {type(e).__name__}: {e}", flush=True)
    m.check_handoffs()

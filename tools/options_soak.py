#!/usr/bin/env python3
"""Race screen for the streaming engine's HIP-graph mode — a long closed loop, graph engine vs launch-by-launch engine, every frame
compared bit for bit; no hand-off time-out allowed anywhere.  (Until round 4 also TIP_OPT_FUSE_HEAD, since removed.)
usage: python tools/options_soak.py [unused] [frames = 3000]"""
import contextlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import tip_amd
from tip_amd import synth, lib as tlib
cfg = synth.PAPER
with contextlib.redirect_stdout(sys.stderr):
    m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                                  dropout=0.0, in_dropout=0.0, past_state_dropout=0.0, with_acc_sum=True)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
m = m.cuda().eval()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
t0 = tlib.spin_timeouts()
h = m._ensure_handle()
from scipy.spatial.transform import Rotation
for n in (1, 8, 64):
    rng = np.random.RandomState(n)
    s_init = (rng.randn(n, 114) * 0.2).astype(np.float32)
    a = tip_amd.streaming.StreamingEngine(m, s_init)
    b = tip_amd.streaming.StreamingEngine(m, s_init, use_graph=True)
    pool = []
    for f in range(16):
        fr = np.zeros((n, 72), dtype=np.float32)
        fr[:, :54] = Rotation.random(n * 6, random_state=1000 * n + f).as_matrix().reshape(n, 54)
        fr[:, 54:] = rng.randn(n, 18)
        pool.append(torch.tensor(fr).cuda())
    bad = 0
    for f in range(frames):
        oa, ob = a.step(pool[f % 16]), b.step(pool[f % 16])
        if oa is None:
            continue
        if not (torch.equal(oa["s_rest"], ob["s_rest"]) and torch.equal(oa["c_t"], ob["c_t"]) and torch.equal(oa["y_last"], ob["y_last"])):
            bad += 1
    torch.cuda.synchronize()
    m.check_handoffs()
    print(f"graph mode n={n:3d}: {frames} frames, {bad} differing from the launch-by-launch engine, finite {bool(torch.isfinite(ob['y_last']).all())}", flush=True)
print("spin time-outs:", tlib.spin_timeouts() - t0)

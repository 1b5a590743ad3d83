"""Race screen of the window-split encoder (plan fused1s) and of AUTO's two-part forwards: thousands of forwards over alternating batch
sizes, every output compared bit for bit with the first one of its batch size; hand-off time-outs must stay at zero."""
import contextlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tip_amd
from tip_amd import synth, lib as tlib
cfg = synth.PAPER
with contextlib.redirect_stdout(sys.stderr):
    m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                                  dropout=0.0, in_dropout=0.0, past_state_dropout=0.0, with_acc_sum=True)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
m = m.cuda().eval()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
sizes = [33, 49, 64, 100, 127, 128, 257, 300, 320, 356, 384, 1, 7, 32, 48]
data, ref = {}, {}
for B in sizes:
    x_imu, x_s = synth.make_inputs(cfg, B, 40, seed=B)
    data[B] = (torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda())
t0 = tlib.spin_timeouts()
bad = 0
start = time.time()
with torch.no_grad():
    for it in range(N):
        B = sizes[it % len(sizes)]
        last = (it // len(sizes)) % 2 == 1
        xi, xs = data[B]
        y = (m.forward_last if last else m)(xi, xs)
        key = (B, last)
        if key not in ref:
            torch.cuda.synchronize()
            assert bool(torch.isfinite(y).all()), key
            ref[key] = y.clone()
        elif not torch.equal(y, ref[key]):
            bad += 1
torch.cuda.synchronize()
m.check_handoffs()
print(f"{N} forwards over batch sizes {sizes} (both output forms): {bad} differing, hand-off time-outs {tlib.spin_timeouts() - t0}, {time.time() - start:.1f} s")
assert bad == 0 and tlib.spin_timeouts() == t0

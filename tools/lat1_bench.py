"""B = 1 .. 8 forward latency: the latency launch chain vs the persistent kernel (plan latency1); run on the GPU box.
TIP_LAT1_SPREAD=1 in the environment selects the all-XCD worker placement."""
import contextlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tip_amd
from tip_amd import synth
cfg = synth.PAPER
with contextlib.redirect_stdout(sys.stderr):
    m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                                  dropout=0.0, in_dropout=0.0, past_state_dropout=0.0, with_acc_sum=True)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
m = m.cuda().eval()
m.freeze_packed(True)
for B in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    x_imu, x_s = synth.make_inputs(cfg, B, 40)
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    for plan in ("latency", "latency1"):
        m.set_plan(plan)
        with torch.no_grad():
            for _ in range(50):
                m.forward_last(xi, xs)
            torch.cuda.synchronize()
            ts = []
            for _ in range(300):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); m.forward_last(xi, xs); e1.record(); e1.synchronize()
                ts.append(e0.elapsed_time(e1) * 1e3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(500):
                m.forward_last(xi, xs)
            e1.record(); e1.synchronize()
        ts = np.array(ts)
        print(f"B={B} {plan:9s}: p50 {np.median(ts):7.1f} us  p95 {np.percentile(ts, 95):7.1f}  back-to-back {e0.elapsed_time(e1) * 2:7.1f} us/forward", flush=True)

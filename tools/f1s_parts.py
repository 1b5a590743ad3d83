"""The window-split encoder with one window on 2 or 4 workgroups ("fused1s2" / "fused1s4": TIP_OPT_F1S_PARTS) against the
   latency plan, 1 <= B <= 64 (best of 3 x 100 forwards; encoder time from the in-library stage timers)."""
import contextlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import tip_amd
from tip_amd import synth
from oracle import oracle
cfg = synth.PAPER
with contextlib.redirect_stdout(sys.stderr):
    m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4,
                                  dropout=0.0, in_dropout=0.0, past_state_dropout=0.0, with_acc_sum=True)
w = synth.make_weights(cfg, seed=0)
m.load_state_dict({k: torch.tensor(v) for k, v in w.items()})
m = m.cuda().eval()
for B in (1, 8, 16, 24, 32, 33, 36, 40, 48, 56, 64):
    x_imu, x_s = synth.make_inputs(cfg, B, 40, seed=B)
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    res = {}
    for plan in ("latency", "fused1s2", "fused1s4"):
        m.set_plan(plan, profile=0)
        with torch.no_grad():
            for _ in range(10):
                y = m(xi, xs)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(100):
                    m(xi, xs)
                e1.record(); e1.synchronize()
                best = min(best, e0.elapsed_time(e1) * 10)
            m.set_plan(plan, profile=1)
            for _ in range(20):
                m(xi, xs)
            torch.cuda.synchronize()
            st = {n: ms / k for n, ms, k in m.profile_read()}
        res[plan] = (y.cpu().numpy(), best, st)
    sel = np.arange(min(B, 3))
    yo = oracle.forward(cfg, w, x_imu[sel], x_s[sel], dtype=np.float64)
    line = f"B={B:3d}"
    for plan, (y, step, st) in res.items():
        line += f"  {plan}: {step:6.1f} us (enc {st.get('fused_encoder', 0) * 1e3:5.1f}) err {np.abs(y[sel] - yo).max():.1e} finite {bool(np.isfinite(y).all())}"
    print(line, flush=True)
m.check_handoffs()
print("hand-offs clean")

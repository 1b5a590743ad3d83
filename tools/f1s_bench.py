"""The window-split encoder (plan fused1s) against the one-window hybrid kernel (fusedh) at B = 1 .. 128: step time, encoder and
recurrence stage times, error against the fp64 oracle, and the distance between the two plans (summation order only)."""
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
for B in (1, 3, 65, 100, 128):
    x_imu, x_s = synth.make_inputs(cfg, B, 40, seed=B)
    xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
    res = {}
    for plan in ("fusedh", "fused1s"):
        m.set_plan(plan, profile=0)
        with torch.no_grad():
            for _ in range(10):
                y = m(xi, xs)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(100):
                m(xi, xs)
            e1.record(); e1.synchronize()
            step = e0.elapsed_time(e1) * 10
            m.set_plan(plan, profile=1)
            for _ in range(20):
                m(xi, xs)
            torch.cuda.synchronize()
            st = {n: ms / k for n, ms, k in m.profile_read()}
        res[plan] = (y.cpu().numpy(), step, st)
    sel = np.arange(min(B, 3))
    yo = oracle.forward(cfg, w, x_imu[sel], x_s[sel], dtype=np.float64)
    for plan, (y, step, st) in res.items():
        print(f"B={B:4d} {plan:8s}: step {step:7.1f} us  encoder {st.get('fused_encoder', 0) * 1e3:7.1f}  rnn {st.get('rnn_recurrence', 0) * 1e3:6.1f}  err vs f64 {np.abs(y[sel] - yo).max():.2e}  finite {np.isfinite(y).all()}", flush=True)
    print(f"        fused1s vs fusedh max diff {np.abs(res['fused1s'][0] - res['fusedh'][0]).max():.2e}")
m.check_handoffs()

"""Every explicit plan and AUTO at the given batch sizes: step time and the encoder / recurrence / projection stage times
(in-library HIP events).  usage: python tools/plan_bench.py 256 300 512 1024"""
import contextlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
for B in [int(a) for a in sys.argv[1:]] or [256]:
    x_imu, x_s = synth.make_inputs(cfg, min(B, 256), 40)
    xi = torch.tensor(np.tile(x_imu, ((B + 255) // 256, 1, 1))[:B]).cuda()
    xs = torch.tensor(np.tile(x_s, ((B + 255) // 256, 1, 1))[:B]).cuda()
    for plan in ("fused", "fusedh", "fused1s", "fused2", "latency", "auto"):
        try:
            m.set_plan(plan, profile=0)
            with torch.no_grad():
                for _ in range(10):
                    m(xi, xs)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(100):
                    m(xi, xs)
                e1.record(); e1.synchronize()
                step = e0.elapsed_time(e1) / 100
                m.set_plan(plan, profile=1)
                for _ in range(20):
                    m(xi, xs)
                torch.cuda.synchronize()
                st = {n: ms / k for n, ms, k in m.profile_read()}
            print(f"B={B} {plan:8s}: step {step:.4f} ms | encoder {st.get('fused_encoder', 0)*1e3:7.1f} us  rnn {st.get('rnn_recurrence', 0)*1e3:6.1f}  head {st.get('out_linear', 0)*1e3:5.1f}", flush=True)
        except Exception as e:
            print(f"B={B} {plan}: {type(e).__name__}: {str(e)[:80]}", flush=True)

"""Where a persistent latency forward (plan latency1) spends its time: worker 0's s_memtime stamps at every stage end and barrier
exit.  Run on the GPU box with TIP_LAT1_TRACE=1."""
import contextlib, ctypes, os, sys
os.environ["TIP_LAT1_TRACE"] = "1"
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
m.set_plan("latency1")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
x_imu, x_s = synth.make_inputs(cfg, B, 40)
xi, xs = torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()
with torch.no_grad():
    for _ in range(20):
        m.forward_last(xi, xs)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
assert tlib.load().tip_debug_read_lat1_trace(buf, 128) == 0
n = int(buf[0])
t = np.array([buf[1 + i] for i in range(n)], dtype=np.float64)
GHZ = 0.1   # s_memtime counts at 100 MHz on this part (tools/probes); printed in us
names = ["in"] + [f"L{l}.{s}" for l in range(4) for s in ("qkv_attn", "out", "ffn1", "ffn2")] + ["ih", "rnn", "head"]
print(f"B={B}: {n} stamps, total {(t[-1] - t[0]) / GHZ / 1e3:.1f} us (if 100 MHz)")
# stamps: start, then per barrier (stage done, barrier passed), final
i = 1
for k, name in enumerate(names):
    if i + 1 >= n: break
    if name == "head":
        print(f"{name:12s} compute {(t[i] - t[i-1]) / GHZ / 1e3:7.2f}")
        break
    print(f"{name:12s} compute {(t[i] - t[i-1]) / GHZ / 1e3:7.2f}  barrier {(t[i+1] - t[i]) / GHZ / 1e3:7.2f}")
    i += 2

"""AUTO plan over a batch sweep: step time, frames/s, fraction of fp32-MFMA peak (run on the GPU box).
TIP_AUTO_SPLIT=0 TIP_PLAN_BASE=1 reproduces the round-3 selection (no remainder split, no window-split plan) for comparison."""
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
base = os.environ.get("TIP_PLAN_BASE") == "1"
fpw = synth.flops_per_window(cfg, 40)
for B in [int(a) for a in sys.argv[1:]] or [1, 8, 32, 33, 40, 48, 49, 64, 65, 100, 128, 129, 200, 256, 257, 272, 300, 320, 356, 384, 512, 556, 1000, 1024, 2048]:
    x_imu, x_s = synth.make_inputs(cfg, min(B, 256), 40)
    xi = torch.tensor(np.tile(x_imu, ((B + 255) // 256, 1, 1))[:B]).cuda()
    xs = torch.tensor(np.tile(x_s, ((B + 255) // 256, 1, 1))[:B]).cuda()
    if base:   # round-3 AUTO: latency plan up to 64 windows, then the cheaper of the one- / two-window kernels
        ncu = 256
        m.set_plan("latency" if B <= 64 else ("fused2" if ((B + 1) // 2 + ncu - 1) // ncu * 1049 < (B + ncu - 1) // ncu * 527 else "fusedh"))
    else:
        m.set_plan("auto")
    with torch.no_grad():
        for _ in range(10):
            m(xi, xs)
        torch.cuda.synchronize()
        n = 200 if B <= 512 else 50
        ms = 1e9
        for _ in range(3):                       # best of three loops: a loop of 30-100 ms can catch a clock dip or a neighbour's burst
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                m(xi, xs)
            e1.record(); e1.synchronize()
            ms = min(ms, e0.elapsed_time(e1) / n)
    print(f"B={B:5d}: {ms * 1e3:8.1f} us/step  {B / ms:9.1f} k frames/s  {B / (ms * 1e-3) * fpw / 157.3e12:6.3f} of fp32-MFMA peak", flush=True)
m.check_handoffs()

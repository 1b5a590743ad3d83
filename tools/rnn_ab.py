"""Measurement: recurrence stage time, whole step and an output digest per batch — run once per variant (environment switch),
the digests must be equal.  usage: [TIP_RNN_W4=1] python tools/rnn_ab.py [B ...]"""
import contextlib, hashlib, os, sys
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
for B in [int(a) for a in sys.argv[1:]] or [100, 256, 512, 1024, 2048]:
    x_imu, x_s = synth.make_inputs(cfg, min(B, 256), 40)
    xi = torch.tensor(np.tile(x_imu, ((B + 255) // 256, 1, 1))[:B]).cuda()
    xs = torch.tensor(np.tile(x_s, ((B + 255) // 256, 1, 1))[:B]).cuda()
    with torch.no_grad():
        m.set_plan("auto", profile=0)
        for _ in range(20):
            y = m(xi, xs)
        torch.cuda.synchronize()
        m.set_plan("auto", profile=1)
        for _ in range(40):
            y = m(xi, xs)
        torch.cuda.synchronize()
        st = {n: ms / k for n, ms, k in m.profile_read()}
        m.set_plan("auto", profile=0)
        for _ in range(10):
            m(xi, xs)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            m(xi, xs)
        e1.record(); e1.synchronize()
    dig = hashlib.sha1(y.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"B={B:5d}: rnn {st['rnn_recurrence'] * 1e3:7.1f} us  step {e0.elapsed_time(e1) * 5:7.1f} us  digest {dig}  finite {bool(torch.isfinite(y).all())}", flush=True)
m.check_handoffs()

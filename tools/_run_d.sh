python - <<'PY'
import sys, json, contextlib, time, cProfile, pstats, io
sys.path.insert(0, '.')
import torch, numpy as np, warnings
import tip_amd
from tip_amd import synth
cfg = synth.PAPER
with contextlib.redirect_stdout(sys.stderr):
    m = tip_amd.TF_RNN_Past_State(72, 131, rnn_hid_size=512, tf_hid_size=1024, tf_in_dim=256, n_heads=16, tf_layers=4, dropout=0.0, in_dropout=0.0, past_state_dropout=0.8, with_acc_sum=True)
m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
m = m.cuda()
x_imu, x_s = synth.make_inputs(cfg, 1, 40, seed=1234)
h_i, h_s = torch.tensor(x_imu), torch.nan_to_num(torch.tensor(x_s))
warnings.simplefilter("ignore")
def frame():
    y = m(h_i.cuda(), h_s.cuda()).cpu()
    return y.squeeze(0)[-1, :].detach().numpy()
for _ in range(50): frame()
ts=[]
for _ in range(300):
    t0=time.perf_counter(); frame(); ts.append(time.perf_counter()-t0)
print("p50 host call ms", np.median(ts)*1e3)
# host time only (no sync)
d_i, d_s = h_i.cuda(), h_s.cuda()
torch.cuda.synchronize()
ts=[]
for _ in range(300):
    torch.cuda.synchronize(); t0=time.perf_counter(); y = m(d_i, d_s); ts.append(time.perf_counter()-t0)
torch.cuda.synchronize()
print("p50 host time of the call alone (no sync) ms", np.median(ts)*1e3)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): frame()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
PY

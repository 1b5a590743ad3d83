"""Race screen of the exact-reuse engine (tip_forward_reuse: ring writer, double-buffered planes of layer 0, newest-row ingest):
every output of every frame must be the same bits as the engine that recomputes every window, over thousands of frames with fresh
random IMU frames, at stream counts that exercise one window per workgroup tail, several pairs per workgroup and the HIP-graph mode;
a side stream keeps a GEMM running so that workgroups do not start in lock step.

    gpurun -- 'python tools/reuse_soak.py [frames]'
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tip_amd  # noqa: E402
from tip_amd import synth  # noqa: E402


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    cfg = synth.PAPER
    m = tip_amd.TF_RNN_Past_State(cfg["input_size_imu"], cfg["size_s"], rnn_hid_size=cfg["rnn_hid_size"], tf_hid_size=cfg["tf_hid_size"],
                                  tf_in_dim=cfg["tf_in_dim"], n_heads=cfg["n_heads"], tf_layers=cfg["tf_layers"], dropout=0.0,
                                  in_dropout=0.0, past_state_dropout=0.0, with_rnn=True, with_acc_sum=True)
    m.load_state_dict({k: torch.tensor(v) for k, v in synth.make_weights(cfg, seed=0).items()})
    m = m.cuda().eval()
    lib = tip_amd.lib.load()
    side = torch.cuda.Stream()
    a = torch.randn(2048, 2048, device="cuda")
    res = []
    for n, graph, load in ((1024, False, False), (515, False, True), (1024, True, False), (7, True, True)):
        g = torch.Generator(device="cuda").manual_seed(n)
        rot = torch.linalg.qr(torch.randn(n, 6, 3, 3, generator=g, device="cuda"))[0].reshape(n, 54)   # orthonormal 3x3 blocks
        s_init = (torch.randn(n, 114, generator=g, device="cuda") * 0.2).cpu().numpy()
        ref = tip_amd.streaming.StreamingEngine(m, s_init)
        eng = tip_amd.streaming.StreamingEngine(m, s_init, use_graph=graph, reuse=True)
        bad = 0
        t0 = time.time()
        for f in range(frames):
            fr = torch.cat([rot, torch.randn(n, 18, generator=g, device="cuda")], dim=1)
            if load:
                with torch.cuda.stream(side):
                    torch.mm(a, a)
            m.set_plan("fused2" if lib.tip_stream_window_len(f) == 40 else "auto")
            x, y = ref.step(fr), eng.step(fr)
            if x is None:
                continue
            for k in ("y_last", "s_rest", "c_t"):
                if not torch.equal(x[k], y[k]):
                    bad += 1
        torch.cuda.synchronize()
        m.set_plan("auto")
        m.check_handoffs()
        res.append({"streams": n, "hip_graph": graph, "side_stream_load": load, "frames": frames, "differing_outputs": bad,
                    "seconds": round(time.time() - t0, 1)})
        print(json.dumps(res[-1]), flush=True)
    print("reuse soak:", "CLEAN" if all(r["differing_outputs"] == 0 for r in res) else "DIFFERENCES")


if __name__ == "__main__":
    main()

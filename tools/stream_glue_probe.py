"""What the streaming front / back end kernels cost on their own: tip_stream_ingest / ingest_newest / consume launched back to back
(their state and outputs stay warm in the caches) against the same kernels inside the closed loop, where a 2-ms encoder runs between
two calls and 0.4 GB of its traffic has pushed the streams' state out of every cache.

    gpurun -- 'python tools/stream_glue_probe.py [n ...]'
"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tip_amd import lib as tlib  # noqa: E402


def main():
    lib = tlib.load()
    ns = [int(a) for a in sys.argv[1:]] or [1, 256, 1024, 4096]
    st = torch.cuda.current_stream().cuda_stream
    for n in ns:
        nb = ctypes.c_size_t()
        lib.tip_stream_state_bytes(n, ctypes.byref(nb))
        state = torch.empty(nb.value, dtype=torch.uint8, device="cuda")
        s_init = torch.randn(n, 114, device="cuda") * 0.2
        lib.tip_stream_reset(state.data_ptr(), s_init.data_ptr(), n, st)
        x_imu = torch.empty(n, 40, 90, device="cuda")
        x_s = torch.empty(n, 40, 131, device="cuda")
        s_rest, c_t = torch.empty(n, 111, device="cuda"), torch.empty(n, 20, device="cuda")
        raw = torch.randn(n, 72, device="cuda")
        raw[:, :54] = torch.linalg.qr(torch.randn(n, 6, 3, 3, device="cuda"))[0].reshape(n, 54)
        y = torch.randn(n, 131, device="cuda") * 0.3
        if os.environ.get("PROBE_SAME_ROWS"):       # every stream the same prediction row: the rotation math takes the same branches everywhere
            y = y[:1].expand(n, 131).contiguous()
            s_init = s_init[:1].expand(n, 114).contiguous()
            raw = raw[:1].expand(n, 72).contiguous()
            lib.tip_stream_reset(state.data_ptr(), s_init.data_ptr(), n, st)
        f = 0
        for f in range(60):                    # prime: windows full
            lib.tip_stream_ingest(state.data_ptr(), raw.data_ptr(), n, f, x_imu.data_ptr(), x_s.data_ptr(), st)
            if f >= 5:
                lib.tip_stream_consume(state.data_ptr(), y.data_ptr(), n, f - 5, s_rest.data_ptr(), c_t.data_ptr(), st)
        res = {"streams": n}
        big = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")     # 256 MB: written between calls = caches flushed
        for flush in (False, True):
            for tag, fn in (("ingest", lambda f: lib.tip_stream_ingest(state.data_ptr(), raw.data_ptr(), n, f, x_imu.data_ptr(), x_s.data_ptr(), st)),
                            ("ingest_newest", lambda f: lib.tip_stream_ingest_newest(state.data_ptr(), raw.data_ptr(), n, f, x_imu.data_ptr(), x_s.data_ptr(), st)),
                            ("consume", lambda f: lib.tip_stream_consume(state.data_ptr(), y.data_ptr(), n, f - 5, s_rest.data_ptr(), c_t.data_ptr(), st))):
                ts = []
                for i in range(30):
                    f += 1
                    if flush:
                        big.fill_(i & 1)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    fn(f)
                    e1.record()
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1e3)
                res[f"{tag}_{'cold' if flush else 'warm'}_us"] = round(float(np.median(ts)), 1)
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()

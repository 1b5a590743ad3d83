"""GPU: the N>1 launch path end to end with the real backend — `python -m torch.distributed.run --nproc-per-node 1 bench.py`
initialises RCCL ("nccl"), broadcasts the packed image, all-gathers the per-rank times and the image checksums, and prints the
one JSON line the driver parses.  (More ranks need more GPUs than the test box has; the rank-count-independent control flow is
what runs here, the world_size-2 data flow is covered on CPU by tests/test_dist_cpu.py.)"""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("config", ["paper256", "streams1024"])
def test_bench_under_torchrun_with_rccl(config):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2",
           "--config", config, "--no-extra", "--no-cpu-baseline"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["world_size"] == 1 and d["backend"].startswith("nccl")
    assert d["packed_image_identical_on_all_ranks"] is True
    assert d["config"]["name"] == config and d["value"] > 1000 and d["higher_is_better"] is True
    assert d["per_rank_ms_per_step"]["min"] <= d["per_rank_ms_per_step"]["max"]
    assert d["roofline"] and 0.05 < d["roofline"]["frac"] < 1.0

"""GPU: the N>1 launch path.  (i) end to end with the real backend — `python -m torch.distributed.run --nproc-per-node 1 bench.py`
initialises RCCL ("nccl"), broadcasts the packed image, all-gathers the per-rank times and the image checksums, and prints the
one JSON line the driver parses.  (More ranks need more GPUs than the test box has; the rank-count-independent control flow is
what runs here.)  (ii) world_size 2 for real on the one GPU the box has: both ranks on cuda:0 over gloo, on a plan without cooperating
kernels (fusedh + rnn_cluster 1) — bench.py itself (TIP_BENCH_SHARE_GPU / TIP_BENCH_BACKEND: its all-gathers, the probe-window
output digests, the scaling table), and a worker pair whose shard outputs must concatenate to a single-process run bit for bit."""
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
    assert d["packed_image_identical_on_all_ranks"] is True and d["ranks_output_identical"] is True
    assert d["config"]["name"] == config and d["value"] > 1000 and d["higher_is_better"] is True
    assert d["per_rank_ms_per_step"]["min"] <= d["per_rank_ms_per_step"]["max"]
    assert d["roofline"] and 0.05 < d["roofline"]["frac"] < 1.0


def _torchrun(nproc, script_args, env_extra, timeout=900):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    return subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, env=env, cwd=ROOT)


def test_bench_world_size_2_on_one_gpu():
    """bench.py as the driver launches it for N = 2, except that both ranks share cuda:0 and talk over gloo: every rank-count-
    dependent line of it runs with world_size 2 — broadcast of the packed image, image checksums, probe-window output digests
    (`ranks_output_identical`), max-over-ranks timing, the scaling table."""
    res = _torchrun(2, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--plan", "fusedh",
                        "--rnn-cluster", "1", "--no-cpu-baseline", "--prewarm-s", "0.05"],
                    {"TIP_BENCH_SHARE_GPU": "1", "TIP_BENCH_BACKEND": "gloo"})
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and d["backend"] == "gloo"
    assert d["packed_image_identical_on_all_ranks"] is True and d["ranks_output_identical"] is True
    assert d["config"]["global_batch"] == 512 and len(d["per_rank_ms_per_step"]["all"]) == 2
    tab = d["extra"]["scaling_table"]
    assert tab["batch1_per_gpu"]["streams_total"] == 2 and tab["streams8192_total"]["streams_per_gpu"] == 4096
    assert tab["streams8192_total"]["frames_per_s"] > 1000
    assert tab["batch256_total"]["streams_per_gpu"] == 128 and tab["batch256_total"]["output"] == "full"
    assert tab["scaled_b4096_total"]["streams_per_gpu"] == 2048 and tab["scaled_b4096_total"]["weight_broadcast_ms"] > 0


def test_two_ranks_one_gpu_shards_equal_single_process(tmp_path):
    """Data flow at world_size 2 with the REAL engine: rank 1 never sees the parameters (only the broadcast image); the two
    shard outputs concatenate to this process's single-rank run of the whole batch, bit for bit, full and last-row output."""
    import numpy as np
    import torch
    from tip_amd import synth
    from test_host_cpu import make_model, load_synth
    B, T = 37, 40
    res = _torchrun(2, [os.path.join(ROOT, "tests", "dist_gpu_worker.py"), str(tmp_path), str(B), str(T)], {})
    assert res.returncode == 0, res.stderr[-3000:]
    ranges = [tuple(np.load(tmp_path / f"r{r}.npy")) for r in range(2)]
    assert ranges == [(0, 19), (19, 37)]
    assert int(np.load(tmp_path / "img0.npy")) == int(np.load(tmp_path / "img1.npy"))
    cfg = synth.PAPER
    m = make_model(cfg)
    load_synth(m, cfg, 0)
    m = m.cuda().eval()
    m.set_plan("fusedh", rnn_cluster=1)
    x_imu, x_s = synth.make_inputs(cfg, B, T, seed=31)
    with torch.no_grad():
        y = m(torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()).cpu().numpy()
        yl = m.forward_last(torch.tensor(x_imu).cuda(), torch.tensor(x_s).cuda()).cpu().numpy()
    assert np.array_equal(np.concatenate([np.load(tmp_path / f"y{r}.npy") for r in range(2)]), y)
    assert np.array_equal(np.concatenate([np.load(tmp_path / f"yl{r}.npy") for r in range(2)]), yl)


def test_plain_bench_command_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher (the driver's invocation): bench.py re-runs itself under torch.distributed.run
    with two ranks; here both share cuda:0 over gloo.  The line must say n_gpus == world_size == 2."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TIP_BENCH_SHARE_GPU="1", TIP_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--plan", "fusedh",
           "--rnn-cluster", "1", "--no-cpu-baseline", "--no-extra", "--prewarm-s", "0.05"]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["world_size"] == 2 and len(d["per_rank_ms_per_step"]["all"]) == 2
    assert d["config"]["global_batch"] == 512 and d["ranks_output_identical"] is True


def test_plain_bench_command_gpus_8_carries_the_whole_north_star_table():
    """The driver's 8-GPU invocation, `python bench.py --gpus 8` (no launcher, default steps), rehearsed with eight ranks on the one GPU
    this box has (TIP_BENCH_SHARE_GPU=1, gloo, a plan without cooperating kernels): the line says n_gpus == world_size == 8 and its
    scaling table has every row of north_star's table — batch 1 per GPU, 256 windows in total (strong), 8192 streams in total (BASELINE
    configs[3]) and the scaled model at 4096 windows in total (configs[4], with its own 609-MB broadcast) — finite, and every rank's
    outputs for the probe windows identical; the whole run inside the wall-time bound (5 minutes; eight ranks SHARING one GPU are
    slower than eight GPUs)."""
    import math
    import time
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", TIP_BENCH_SHARE_GPU="1", TIP_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--plan", "fusedh", "--rnn-cluster", "1"]
    t0 = time.time()
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env, cwd=ROOT)
    wall = time.time() - t0
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["world_size"] == 8 and len(d["per_rank_ms_per_step"]["all"]) == 8
    assert d["config"]["global_batch"] == 2048 and d["scaling"] == "weak"
    assert d["ranks_output_identical"] is True and d["packed_image_identical_on_all_ranks"] is True
    tab = d["extra"]["scaling_table"]
    want = {"batch1_per_gpu": (1, 8), "batch256_total": (32, 256), "streams8192_total": (1024, 8192),
            "streams8192_closed_loop_total": (1024, 8192), "scaled_b4096_total": (512, 4096)}
    assert set(tab) == set(want)
    for k, (per, tot) in want.items():
        row = tab[k]
        assert "error" not in row, (k, row)
        assert row["streams_per_gpu"] == per and row["streams_total"] == tot
        for f in ("ms_per_step", "frames_per_s", "whole_forward_frac_of_fp32_mfma_peak"):
            assert math.isfinite(row[f]) and row[f] > 0, (k, f, row[f])
    assert tab["scaled_b4096_total"]["weight_broadcast_ms"] > 0 and tab["scaled_b4096_total"]["packed_image_mb"] > 600
    assert tab["streams8192_closed_loop_total"]["reuse_engaged"] is True and tab["streams8192_closed_loop_total"]["recompute_ms_per_step"] > 0
    assert len(d["host_pinning"]["numa_node_per_rank"]) == 8
    assert d["tail_summary"]["table_8gpu"]["scaled_b4096_total"][0] > 0
    assert wall < 300 and d["wall_s"] < 300, (wall, d["wall_s"])


def test_plain_bench_command_refuses_more_gpus_than_the_box_has():
    """`python bench.py --gpus 8` on a box with fewer GPUs fails loudly: no N = 1 line, non-zero exit."""
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("an 8-GPU node: the command is legitimate here")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TIP_BENCH_SHARE_GPU"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode != 0
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")], res.stdout
    assert "refusing" in res.stderr


def test_world_size_mismatch_is_an_error():
    """One rank under the launcher but --gpus 2: a hard error, not an N = 1 line."""
    res = _torchrun(1, [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-extra",
                        "--no-cpu-baseline"], {})
    assert res.returncode != 0
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")], res.stdout

"""world_size-2 gloo tests (CPU) of the N>1 path: one weight broadcast, batch sharding, no data-path collective."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    import tip_amd
    from tip_amd import synth, dist as tdist
    from oracle import oracle
    from test_host_cpu import make_model, load_synth
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = synth.TINY
    m = make_model(cfg)                      # every rank starts from its own random init ...
    if rank == 0:
        w = load_synth(m, cfg, 0)            # ... only rank 0 holds the real weights
    packed = tdist.broadcast_packed(m, src=0, device="cpu")
    ref = make_model(cfg)
    load_synth(ref, cfg, 0)
    assert torch.equal(packed, ref.pack_host()), "broadcast image differs from rank 0's packed weights"
    # shard the streams, run the shard (oracle stands in for the GPU engine in this CPU test), gather on rank 0
    B, T = 7, 9
    x_imu, x_s = synth.make_inputs(cfg, B, T, seed=3)
    lo, hi = tdist.shard_range(B, rank, world)
    w0 = synth.make_weights(cfg, seed=0)
    y_shard = oracle.forward(cfg, w0, x_imu[lo:hi], x_s[lo:hi], dtype=np.float32, nthreads=1)
    np.save(os.path.join(out_dir, f"y{rank}.npy"), y_shard)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), np.array([lo, hi]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_broadcast_and_sharding(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    import tip_amd
    from tip_amd import synth
    from oracle import oracle
    cfg = synth.TINY
    x_imu, x_s = synth.make_inputs(cfg, 7, 9, seed=3)
    y = oracle.forward(cfg, synth.make_weights(cfg, seed=0), x_imu, x_s, dtype=np.float32, nthreads=1)
    parts = [np.load(tmp_path / f"y{r}.npy") for r in range(world)]
    ranges = [tuple(np.load(tmp_path / f"r{r}.npy")) for r in range(world)]
    assert ranges == [(0, 4), (4, 7)]
    assert np.array_equal(np.concatenate(parts), y)     # shards concatenate bit-for-bit: no cross-stream op


def test_shard_range_partitions():
    from tip_amd import dist as tdist
    for n in (0, 1, 7, 8, 256, 8192, 8195):
        for world in (1, 2, 3, 4, 8):
            rs = [tdist.shard_range(n, r, world) for r in range(world)]
            assert rs[0][0] == 0 and rs[-1][1] == n
            assert all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in rs]
            assert max(sizes) - min(sizes) <= 1


def test_plain_bench_command_refuses_without_the_gpus():
    """`python bench.py --gpus 8` where 8 GPUs are not there (here: none) exits non-zero without a JSON line — the driver's plain
    invocation can never be answered with a silent N = 1 measurement."""
    import subprocess
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "TIP_BENCH_SHARE_GPU"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode != 0
    assert not [l for l in res.stdout.splitlines() if l.startswith("{")], res.stdout
    assert "refusing" in res.stderr

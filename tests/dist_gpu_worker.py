"""Worker of tests/test_dist_gpu.py::test_two_ranks_one_gpu_shards_equal_single_process (launched by torch.distributed.run,
world_size 2, backend gloo, BOTH ranks on cuda:0).  Rank 0 holds the weights; ONE broadcast of the packed image; every rank
runs its contiguous shard of the batch on the HIP engine (plan fusedh + rnn_cluster 1: no cooperating kernels, so two processes
can share the GPU) and writes it to <out>/y<rank>.npy."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from tip_amd import synth, dist as tdist  # noqa: E402
from test_host_cpu import make_model, load_synth  # noqa: E402


def main():
    out, B, T = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, _, world = tdist.env_rank()
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    cfg = synth.PAPER
    torch.manual_seed(100 + rank)
    m = make_model(cfg)                       # every rank starts from its own random init ...
    if rank == 0:
        load_synth(m, cfg, 0)                 # ... only rank 0 holds the real weights
    m = m.to(dev).eval()
    packed = tdist.broadcast_packed(m, src=0, device=dev)
    m.set_plan("fusedh", rnn_cluster=1)
    x_imu, x_s = synth.make_inputs(cfg, B, T, seed=31)
    lo, hi = tdist.shard_range(B, rank, world)
    with torch.no_grad():
        y = m(torch.tensor(x_imu[lo:hi]).to(dev), torch.tensor(x_s[lo:hi]).to(dev))
        yl = m.forward_last(torch.tensor(x_imu[lo:hi]).to(dev), torch.tensor(x_s[lo:hi]).to(dev))
    torch.cuda.synchronize()
    assert m.hip_forward_count() == 2
    m.check_handoffs()
    np.save(os.path.join(out, f"y{rank}.npy"), y.cpu().numpy())
    np.save(os.path.join(out, f"yl{rank}.npy"), yl.cpu().numpy())
    np.save(os.path.join(out, f"r{rank}.npy"), np.array([lo, hi]))
    np.save(os.path.join(out, f"img{rank}.npy"), packed.view(torch.int32).to(torch.int64).sum().cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

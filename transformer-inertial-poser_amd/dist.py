"""Multi-GPU: independent IMU streams shard on the batch axis; the only collective is a one-time broadcast of
the packed weight image from rank 0 (RCCL over xGMI when the backend is "nccl").  No per-step communication
(no op in simple_transformer_with_state.py:60-102 crosses batch elements)."""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def env_rank() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torch.distributed.run environment; (0, 0, 1) when absent."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_range(n_streams: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous split of the stream (batch) axis; the first n % world ranks get one extra stream."""
    q, r = divmod(n_streams, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def broadcast_packed(model, src: int = 0, device=None) -> torch.Tensor:
    """Rank `src` packs its parameters into the kernel weight image; every rank receives it with ONE broadcast
    and (on a GPU) attaches it to its handle.  Returns the packed uint8 tensor on `device`."""
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    if device is None:
        device = next(model.parameters()).device
    device = torch.device(device)
    nbytes = model._ensure_handle().packed_bytes()
    if rank == src:
        # parameters already on this GPU: pack there (bit-identical image, no host loops — matters for the 609-MB scaled model)
        on_dev = device.type == "cuda" and all(p.is_cuda and p.device == device for p in model.parameters())
        packed = model.pack_device(device) if on_dev else model.pack_host().to(device)
    else:
        packed = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if world > 1:
        dist.broadcast(packed, src=src)
    if device.type == "cuda":
        model.attach_packed(packed)
        model.freeze_packed(True)  # parameters on non-src ranks are not the source of truth any more
    return packed

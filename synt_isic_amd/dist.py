"""Multi-GPU: independent images shard across ranks, one gather of finished samples at the end.

The reference has no distributed code; the path shards naturally because every
image's chain depends only on its own seed and the replicated weights (section 8e).  One process per GPU
(``torch.distributed``, backend "nccl" = RCCL over xGMI on ROCm); no collective inside the loop;
exactly one collective afterwards: gather the uint8 images (B_local*H*W*3 bytes per rank) to rank 0.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_range(n_total: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous block [lo, hi) of image indices for ``rank`` (first n_total % world ranks get one more)."""
    if not (0 <= rank < world_size):
        raise ValueError(f"rank {rank} outside world of {world_size}")
    q, r = divmod(n_total, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_seeds(seeds: Sequence[int], world_size: int, rank: int) -> List[int]:
    lo, hi = shard_range(len(seeds), world_size, rank)
    return list(seeds[lo:hi])


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torch.distributed.run environment; initialises the process
    group when WORLD_SIZE > 1.  HSA_ENABLE_IPC_MODE_LEGACY=0 is required on this pool (dmabuf IPC)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if share_gpu_rehearsal() and torch.cuda.is_available():
        # SISIC_SHARE_GPU=1: a REHEARSAL of the N-rank path on a box with fewer GPUs than ranks (the builder's boxes have one):
        # ranks take device LOCAL_RANK % device_count and the collectives go through gloo on host copies (RCCL refuses two
        # ranks on one device).  Launcher, rendezvous, sharding, gather and the max-over-ranks are the real ones; the
        # throughput of such a run means nothing.
        local = local % torch.cuda.device_count()
        backend = backend or "gloo"
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def share_gpu_rehearsal() -> bool:
    return os.environ.get("SISIC_SHARE_GPU", "0") == "1"


def _host_collectives() -> bool:
    """gloo moves host memory: device tensors are staged through the host for its collectives (the rehearsal mode)"""
    return dist.is_initialized() and dist.get_backend() == "gloo"


def gather_images(local: torch.Tensor, n_total: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Gather per-rank image blocks [B_local, ...] into [n_total, ...] on ``dst`` (None elsewhere).

    Blocks may differ in length by one (``shard_range``); they are padded to the largest block for the
    collective and trimmed afterwards, so one gather serves ragged shards too."""
    if not dist.is_initialized():
        if local.shape[0] != n_total:
            raise ValueError("single-process gather expects the full batch")
        return local
    # an initialised group of one rank still goes through the collective (RCCL on the GPU): the N>1 code path is the
    # one that runs, whatever the world size
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(n_total, world, rank)
    if local.shape[0] != hi - lo:
        raise ValueError(f"rank {rank} holds {local.shape[0]} images, its shard is [{lo},{hi})")
    b_max = -(-n_total // world)
    send = local
    if local.shape[0] < b_max:
        pad = torch.zeros((b_max - local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send = torch.cat([local, pad], dim=0)
    send = send.contiguous()
    dev = send.device
    if _host_collectives() and dev.type != "cpu":
        send = send.cpu()
    bufs = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send, gather_list=bufs, dst=dst)
    if rank != dst:
        return None
    parts = []
    for r in range(world):
        rlo, rhi = shard_range(n_total, world, r)
        parts.append(bufs[r][: rhi - rlo])
    return torch.cat(parts, dim=0).to(dev)


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if _host_collectives() else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

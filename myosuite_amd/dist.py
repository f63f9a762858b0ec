"""Multi-GPU sharding: one process per GPU, envs partitioned with NO collective on the step path.

Environments are independent, so the physics step never communicates.  The only
exchange is the reporting step of a rollout: an all-gather of per-env episode statistics
(return, length, solved) over RCCL/xGMI (backend "nccl" is RCCL on ROCm; "gloo" for CPU tests).
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """(rank, world_size, local_rank) from torchrun-style env vars; initialises the process group if needed."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def backend() -> str:
    """backend of the default process group ("nccl" = RCCL on ROCm, "gloo"), or "" without a group"""
    return dist.get_backend() if dist.is_initialized() else ""


def shard_envs(total_envs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [start, start+count) of a global env index range (remainder to the low ranks)."""
    base, rem = divmod(total_envs, world)
    count = base + (1 if rank < rem else 0)
    start = rank * base + min(rank, rem)
    return start, count


def gather_episode_stats(stats: torch.Tensor, always_collective: bool = False) -> torch.Tensor:
    """All-gather [E_local, 3] (episode_return, episode_length, solved) -> [world*E_local, 3] on every rank.
    Shards must be equal-sized (weak scaling: fixed envs per GPU).  `always_collective` issues the RCCL / gloo call even in a
    one-rank group (the single-GPU smoke test of the collective path)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not always_collective):
        return stats
    world = dist.get_world_size()
    if stats.is_cuda and dist.get_backend() == "gloo":
        # gloo group over device tensors (bench.py --oversubscribe, CPU-only launch tests): stage through the host
        parts = [torch.empty(stats.shape, dtype=stats.dtype) for _ in range(world)]
        dist.all_gather(parts, stats.detach().cpu().contiguous())
        return torch.cat(parts, 0).to(stats.device)
    out = torch.empty((world * stats.shape[0],) + tuple(stats.shape[1:]), dtype=stats.dtype, device=stats.device)
    dist.all_gather_into_tensor(out, stats.contiguous())
    return out


def max_over_ranks(x: float, device=None) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()

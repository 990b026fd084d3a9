"""Sharding of independent local-BA windows across ranks (BASELINE cfg5).

The path shards across windows only: one process per GPU, each rank solves its own
windows with no data-path collective; the single exchange step is an all-gather of the
optimised keyframe poses (RCCL over xGMI on GPUs, gloo in the CPU tests).  A single
window is never split across ranks (the reduced system couples every keyframe pair).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def windows_for_rank(n_windows: int, rank: int, world: int) -> list[int]:
    """Round-robin assignment of window ids to ranks."""
    return list(range(rank, n_windows, world))


def window_seed(window_id: int, base_seed: int = 2000) -> int:
    """cfg5 seeds 2000..2007; window 0 of a 1-GPU run is cfg3 itself (seed 1003)."""
    return base_seed + window_id


def gather_poses(local: torch.Tensor, group=None, force_collective: bool = False) -> torch.Tensor:
    """local: (n_local, NP, 7) f64 on this rank's device -> (world, n_local, NP, 7) on every rank.
    force_collective: run the all-gather even in a one-rank group (a real RCCL call on a box with one GPU)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1 and not (force_collective and dist.is_initialized()):
        return local.unsqueeze(0).clone()
    flat = local.contiguous().view(-1)
    out = torch.empty(world * flat.numel(), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, flat, group=group)
    return out.view((world,) + tuple(local.shape))

"""Multi-GPU: independent video sequences, one process per GPU.

A sequence is strictly serial (frame t needs frame t-1's disparity, hidden states and features) and
a 120x160 grid is far too small to split, so the only parallel axis is across sequences: rank r takes
sequences r, r+world, ...  Every rank holds a full weight replica (67 MB).  The data path has no
collective at all; ONE all_gather of a 6-double vector per rank at the end reproduces the reference's
EPE/D1/D3 reduction (evaluate_stereo.py:214-220) on every rank.  On the GPU box the backend is
"nccl" (= RCCL over xGMI); CPU tests use gloo.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def init_from_env(force_backend: str | None = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's environment; initialises the process group when world > 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # TCS_MI355_DIST_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks
        backend = force_backend or os.environ.get("TCS_MI355_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def shard(items: Sequence, rank: int, world: int) -> List:
    """Round-robin: sequence i belongs to rank i % world."""
    return [it for i, it in enumerate(items) if i % world == rank]


def barrier():
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def gather_vectors(vec: np.ndarray, device=None) -> List[np.ndarray]:
    """all_gather of one small float64 vector per rank (the only collective of a run)."""
    vec = np.asarray(vec, np.float64)
    if not (dist.is_available() and dist.is_initialized()):
        return [vec]
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    mine = torch.from_numpy(vec).to(device)
    out = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [t.cpu().numpy() for t in out]


def max_over_ranks(value: float, device=None) -> float:
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())

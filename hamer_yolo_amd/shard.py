"""Crop-shard data parallelism: one process per GPU, crops partitioned in
contiguous ranges, no collective on the data path.  RCCL (torch.distributed backend "nccl"
on ROCm) carries two things only: the one-off weight broadcast from rank 0 and the per-batch
gather of per-hand MANO parameters (157 floats: 16 rotation matrices, 10 betas, 3 camera).
The same code runs on gloo/CPU tensors for the world_size-2 tests.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

PARAMS_PER_HAND = 16 * 9 + 10 + 3


def init_distributed(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torchrun contract)."""
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
    return rank, local, world


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous range of rank `rank` when n items are split over `world` ranks (ceil(n/world) each)."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def broadcast_state_dict(sd: Optional[Dict[str, torch.Tensor]], keys: List[str], shapes: Dict[str, tuple],
                         device, src: int = 0) -> Dict[str, torch.Tensor]:
    """Rank `src` holds `sd`; everybody returns the same tensors (one broadcast per tensor)."""
    out = {}
    for k in keys:
        if dist.is_initialized() and dist.get_world_size() > 1:
            t = sd[k].to(device).contiguous() if dist.get_rank() == src else torch.empty(shapes[k], dtype=torch.float32, device=device)
            dist.broadcast(t, src=src)
        else:
            t = sd[k].to(device)
        out[k] = t
    return out


def pack_mano(out: Dict[str, torch.Tensor]) -> torch.Tensor:
    """(B, 157) = [rotmats(144) | betas(10) | cam(3)]."""
    B = out["betas"].shape[0]
    return torch.cat([out["rotmats"].reshape(B, 144), out["betas"], out["pred_cam"]], dim=1).contiguous()


def gather_mano(packed: torch.Tensor, dst: int = 0) -> Optional[torch.Tensor]:
    """All ranks hold (B_local, 157) with equal B_local; rank `dst` returns (world*B_local, 157)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return packed
    world = dist.get_world_size()
    full = torch.empty(world * packed.shape[0], packed.shape[1], dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(full, packed)
    return full if dist.get_rank() == dst else None

"""Crop-shard data parallelism: one process per GPU, crops partitioned in
contiguous ranges, no collective on the data path.  RCCL (torch.distributed backend "nccl"
on ROCm) carries two things only: the one-off weight broadcast from rank 0 (two flat buffers:
the GEMM matrices already rounded to the 16-bit operand type, everything else fp32) and the
gather of per-hand MANO parameters (157 floats: 16 rotation matrices, 10 betas, 3 camera).
The same code runs on gloo/CPU tensors for the world_size-2 tests.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

PARAMS_PER_HAND = 16 * 9 + 10 + 3

# state-dict entries HamerEngine converts to its 16-bit operand type (engine.py w16): they travel as 16-bit
GEMM_WEIGHT_SUFFIXES = ("attn.qkv.weight", "attn.proj.weight", "mlp.fc1.weight", "mlp.fc2.weight", "patch_embed.proj.weight",
                        "fn.to_kv.weight")


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


def _multi() -> bool:
    return dist.is_initialized() and dist.get_world_size() > 1


def broadcast_state_dict(sd: Optional[Dict[str, torch.Tensor]], device, src: int = 0,
                         half_dtype: torch.dtype = torch.float16) -> Dict[str, torch.Tensor]:
    """Rank `src` holds the fp32 state dict `sd`; every rank returns it.  Three collectives in all: the key/shape table,
    ONE flat buffer of the GEMM matrices rounded to `half_dtype` (what the engine would round them to anyway: 1.3 GB instead
    of 2.6 GB for ViT-H) and ONE flat fp32 buffer of everything else.  The returned tensors are views of the two buffers;
    GEMM matrices come back in `half_dtype`."""
    if not _multi():
        return {k: v.to(device) for k, v in sd.items()}
    rank = dist.get_rank()
    meta = [[(k, tuple(v.shape), k.endswith(GEMM_WEIGHT_SUFFIXES)) for k, v in sd.items()]] if rank == src else [None]
    dist.broadcast_object_list(meta, src=src)
    out: Dict[str, torch.Tensor] = {}
    for is_half, dt in ((True, half_dtype), (False, torch.float32)):
        group = [(k, shp) for k, shp, h in meta[0] if h == is_half]
        total = sum(int(torch.Size(shp).numel()) for _, shp in group)
        if total == 0:
            continue
        if rank == src:
            flat = torch.cat([sd[k].detach().to(device, torch.float32).to(dt).reshape(-1) for k, _ in group])
        else:
            flat = torch.empty(total, dtype=dt, device=device)
        dist.broadcast(flat, src=src)
        off = 0
        for k, shp in group:
            n = int(torch.Size(shp).numel())
            out[k] = flat[off:off + n].view(shp)
            off += n
    return {k: out[k] for k, _, _ in meta[0]}


def pack_mano(out: Dict[str, torch.Tensor]) -> torch.Tensor:
    """(B, 157) = [rotmats(144) | betas(10) | cam(3)]."""
    B = out["betas"].shape[0]
    return torch.cat([out["rotmats"].reshape(B, 144), out["betas"], out["pred_cam"]], dim=1).contiguous()


def gather_mano(packed: torch.Tensor, dst: int = 0, n_total: Optional[int] = None) -> Optional[torch.Tensor]:
    """Rank `dst` returns the rows of all ranks in rank order, the others None.  Without `n_total` every rank holds the same
    number of rows.  With `n_total` the ranks hold their shard_range(n_total, rank, world) rows (the last ranks may hold
    fewer, or none): rows are padded to ceil(n_total / world) for the one all_gather and trimmed again on `dst`."""
    if not _multi():
        return packed
    world, rank = dist.get_world_size(), dist.get_rank()
    if n_total is None:
        full = torch.empty(world * packed.shape[0], packed.shape[1], dtype=packed.dtype, device=packed.device)
        dist.all_gather_into_tensor(full, packed)
        return full if rank == dst else None
    per = (n_total + world - 1) // world
    lo, hi = shard_range(n_total, rank, world)
    assert packed.shape[0] == hi - lo, "this rank must hold exactly its shard"
    padded = packed if hi - lo == per else torch.cat([packed, packed.new_zeros(per - (hi - lo), packed.shape[1])])
    full = torch.empty(world * per, packed.shape[1], dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(full, padded.contiguous())
    return full[:n_total] if rank == dst else None         # shards are contiguous and only trailing ranks are short

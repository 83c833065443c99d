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


# Opt-in: run the collectives even when the group has ONE rank.  A one-GPU box can then push the real buffers (the 1.3 GB
# flat weight broadcast, the MANO gather) through RCCL itself -- init_process_group("nccl"), broadcast, all_gather_into_tensor
# -- instead of taking the single-process shortcut (tests/test_gpu_shard.py; `HAMER_RCCL_AT_WORLD1=1 python bench.py`).
FORCE_COLLECTIVES = False


def init_distributed(backend: Optional[str] = None, force: Optional[bool] = None) -> Tuple[int, int, int]:
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment (torchrun contract).  A process group is created
    when WORLD_SIZE > 1 -- or, with ``force`` (default: HAMER_RCCL_AT_WORLD1=1), at world size 1 too, in which case every
    collective of this module runs through the backend instead of being skipped."""
    global FORCE_COLLECTIVES
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if force is None:
        force = os.environ.get("HAMER_RCCL_AT_WORLD1", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if force and world == 1:
        FORCE_COLLECTIVES = True
    return rank, local, world


def shutdown_distributed() -> None:
    global FORCE_COLLECTIVES
    FORCE_COLLECTIVES = False
    if dist.is_initialized():
        dist.destroy_process_group()


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous range of rank `rank` when n items are split over `world` ranks (ceil(n/world) each)."""
    per = (n + world - 1) // world
    lo = min(n, rank * per)
    return lo, min(n, lo + per)


def _multi() -> bool:
    return dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


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


class ShardJob:
    """BASELINE configs[3] on one rank: this rank's contiguous share of an n_total-crop job, resident in HBM, run as forwards
    of `batch` crops on alternating in-flight contexts (a ragged last forward gets its own outputs on the same context's
    stream and workspace), per-hand MANO parameters packed into one [n_mine][157] buffer, ONE gather per job.  bench.py
    (--workload shard1024) and tests/test_gpu_shard.py run the same step() -- the reference's counterpart is the serial
    one-hand loop at hamer/infer.py:1268-1274."""

    def __init__(self, eng, crops: Optional[torch.Tensor], n_total: int, batch: int = 64, in_flight: int = 2, contexts=None):
        import torch as _t
        self.eng, self.n_total, self.batch = eng, n_total, batch
        self.crops = crops                                        # (n_mine, 3, 256, 256) f32 on the device, or None when this rank holds nothing
        n = 0 if crops is None else crops.shape[0]
        # (contexts: reuse a caller's (stream, workspace, outputs) triples of the same batch size.  HIP maps streams onto a few
        # hardware queues; a process that has already created a pair gets another mapping for the next pair, and two pairs do
        # not overlap alike -- measured: the same job 6 % slower on a second pair of streams)
        self.ctxs = contexts if contexts is not None else eng.contexts(batch, in_flight)
        self.pieces = [(a, min(a + batch, n)) for a in range(0, n, batch)]
        self.packed = _t.zeros(n, PARAMS_PER_HAND, device=eng.device)
        self._tail_out = None

    def step(self) -> Optional[torch.Tensor]:
        eng, dev, B = self.eng, self.eng.device, self.batch
        for j, (a, b) in enumerate(self.pieces):
            c = self.ctxs[j % len(self.ctxs)]
            if b - a == B:
                eng.forward_on(c, self.crops[a:b])
                res = c.out
            else:                                                  # ragged last forward of the shard
                if self._tail_out is None:
                    self._tail_out = eng.alloc_outputs(b - a)
                c.stream.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(c.stream):
                    res = eng.forward(self.crops[a:b].contiguous(), self._tail_out, workspace=c.workspace)
            with torch.cuda.stream(c.stream):
                self.packed[a:b] = pack_mano(res)
        for c in self.ctxs:
            torch.cuda.current_stream(dev).wait_stream(c.stream)
        return gather_mano(self.packed, dst=0, n_total=self.n_total)     # one collective per job (0.64 MB at 1024 hands)

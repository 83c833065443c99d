"""ORACLE (test infrastructure): CPU restatement of the reference's token-merging ("ToMe") backbone variant --
``HAMER_INFER(token_merge=True)`` (hamer/hamer/models/hamer.py:468-483): ``apply_patch(backbone)`` with
``backbone.r = (8, -1)`` from hamer/hamer/models/backbones/selective_vit_adapter.py.

Pinned (tools/gen_golden.py, build container) against that module itself -- ``apply_patch`` on the reference ``ViT``,
``bipartite_soft_matching`` (:17-96), ``merge_wavg`` (:98-113), ``parse_r`` (:132-157), ``ToMeAttention`` (:159-198),
``ToMeBlock`` (:200-235) -- on seeded weights; outputs committed as tests/golden/hamer_tome.npz.

What the variant does per block: proportional attention (``+ log(size)`` on the key axis), the residual add, then
``r_i`` tokens are merged away by bipartite soft matching on the head-averaged keys (tokens alternate between the sets
A (even) and B (odd); every A token proposes its most similar B token, the ``r_i`` best proposals are merged,
size-weighted), then the MLP on the shorter sequence.  No class token, no source tracing (apply_patch defaults).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

from .hamer_ref import _linear, _q, patch_embed, vit_mlp

Tensor = torch.Tensor


def parse_r(num_layers: int, r) -> List[int]:
    """selective_vit_adapter.py:132-157: int, (r, inflection) or a list -> tokens to remove per layer."""
    inflect = 0
    if isinstance(r, list):
        return list(r) + [0] * max(0, num_layers - len(r))
    if isinstance(r, tuple):
        r, inflect = r
    min_val = int(r * (1.0 - inflect))
    max_val = 2 * r - min_val
    step = (max_val - min_val) / (num_layers - 1)
    return [int(min_val + step * i) for i in range(num_layers)]


def bipartite_match(metric: Tensor, r: int) -> Tuple[Tensor, Tensor, Tensor]:
    """selective_vit_adapter.py:17-66 without class / distillation token.  metric (B, N, C) -> (unm_idx (B, Na - r),
    src_idx (B, r), dst_idx (B, r)): indices into the A (even) tokens that stay / are merged away, and, for the merged
    ones, the B (odd) token they join.  Equal proposal scores keep the lower A index first (a stable descending sort;
    the reference's argsort leaves ties unspecified)."""
    m = metric / metric.norm(dim=-1, keepdim=True)
    a, b = m[..., ::2, :], m[..., 1::2, :]
    scores = a @ b.transpose(-1, -2)
    node_max, node_idx = scores.max(dim=-1)
    edge_idx = torch.argsort(node_max, dim=-1, descending=True, stable=True)
    unm_idx, src_idx = edge_idx[..., r:], edge_idx[..., :r]
    dst_idx = node_idx.gather(dim=-1, index=src_idx)
    return unm_idx, src_idx, dst_idx


def merge_tokens(x: Tensor, size: Tensor, unm_idx: Tensor, src_idx: Tensor, dst_idx: Tensor) -> Tuple[Tensor, Tensor]:
    """merge_wavg (:98-113) over merge(mode="sum") (:68-80): size-weighted mean of every merged group.
    x (B, N, C), size (B, N, 1) -> (B, N - r, C), (B, N - r, 1): [unmerged A tokens in proposal-score order | all B tokens]."""
    def merge_sum(t):
        src, dst = t[..., ::2, :], t[..., 1::2, :].clone()
        c = t.shape[-1]
        unm = src.gather(dim=-2, index=unm_idx[..., None].expand(*unm_idx.shape, c))
        moved = src.gather(dim=-2, index=src_idx[..., None].expand(*src_idx.shape, c))
        dst.scatter_add_(-2, dst_idx[..., None].expand(*dst_idx.shape, c), moved)
        return torch.cat([unm, dst], dim=1)
    xs = merge_sum(x * size)
    ns = merge_sum(size)
    return xs / ns, ns


def tome_attention(sd, h: Tensor, p: str, vit, size, emu=False) -> Tuple[Tensor, Tensor]:
    """ToMeAttention.forward (:166-198): returns (attention output, head-averaged keys)."""
    B, N, C = h.shape
    qkv = _q(_linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"], emu), emu)
    qkv = qkv.reshape(B, N, 3, vit.heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = (q @ k.transpose(-2, -1)) * (vit.head_dim ** -0.5)
    if size is not None:
        attn = attn + size.log()[:, None, None, :, 0]
    attn = attn.softmax(dim=-1)
    o = _q((attn @ v).transpose(1, 2).reshape(B, N, C), emu)
    metric = k.mean(1)                                   # the reference: head-averaged keys (:198)
    if emu:
        # the kernels' arithmetic (round 3): k.mean(heads) is linear in h, so the build forms the metric entirely in fp32 (an
        # fp32 LayerNorm output, the fp32 head-mean of the key weights) instead of averaging 16-bit keys -- i.e. exactly the
        # reference's expression above evaluated on unquantised operands
        D = vit.embed_dim
        k32 = F.linear(h, sd[p + "attn.qkv.weight"][D:2 * D], sd[p + "attn.qkv.bias"][D:2 * D])
        metric = k32.reshape(B, N, vit.heads, -1).mean(2)
    return _linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"], emu), metric


def vit_forward_tome(sd, x: Tensor, vit, r=(8, -1), emu=False, prefix="backbone.", trace: Dict = None) -> Tensor:
    """ViT.forward_features (vit.py:320-339) with ToMeBlock.forward (:210-235) in place of Block.forward.
    Returns (B, N_final, D).  ``trace`` (optional dict) receives the per-layer matchings and token counts."""
    D = vit.embed_dim
    t = patch_embed(sd, x, vit, emu, prefix)
    pos = sd[prefix + "pos_embed"]
    t = t + pos[:, 1:] + pos[:, :1]
    size = None
    rs = parse_r(vit.depth, r)
    for i in range(vit.depth):
        p = f"{prefix}blocks.{i}."
        h = F.layer_norm(t, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], vit.ln_eps)
        a, metric = tome_attention(sd, h, p, vit, size, emu)
        t = t + a
        ri = min(rs[i], t.shape[1] // 2)
        if ri > 0:
            unm, src, dst = bipartite_match(metric, ri)
            if size is None:
                size = torch.ones_like(t[..., :1])
            t, size = merge_tokens(t, size, unm, src, dst)
            if trace is not None:
                trace.setdefault("match", []).append((unm, src, dst))
        if trace is not None:
            trace.setdefault("tokens", []).append(t.shape[1])
        h = F.layer_norm(t, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], vit.ln_eps)
        t = t + vit_mlp(sd, h, p, emu)
    return F.layer_norm(t, (D,), sd[prefix + "last_norm.weight"], sd[prefix + "last_norm.bias"], vit.ln_eps)

"""Load-time weight folding for YOLOv7 (host side, once per model):
  * Conv2d(bias=False) + BatchNorm2d -> Conv2d(bias=True)   (fuse_conv_and_bn, utils/torch_utils.py:181-201)
  * RepConv 3x3 + 1x1 (+ identity BN) branches -> one 3x3  (RepConv.fuse_repvgg_block, models/common.py:588-647)
  * IDetect ImplicitA / ImplicitM -> detect conv weight+bias (IDetect.fuse, models/yolo.py:186-198)
Input: an UNFUSED state dict keyed like the reference ``Model`` (``model.<i>.conv.weight``,
``model.<i>.bn.*``, ``model.<i>.rbr_dense.0.weight`` ...).  Output: ``{name: (weight, bias)}`` keyed as in
``arch.conv_specs``.  BatchNorm eps is 1e-3: ``initialize_weights`` (utils/torch_utils.py:170-178) sets it on
every BatchNorm2d of the model, and it is a module attribute, not part of the state dict.
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

BN_EPS = 1e-3


def fuse_conv_bn(w: torch.Tensor, bn_w, bn_b, bn_mean, bn_var, eps: float = BN_EPS) -> Tuple[torch.Tensor, torch.Tensor]:
    """torch_utils.py:181-201 (conv has no bias)."""
    co = w.shape[0]
    w_bn = torch.diag(bn_w.div(torch.sqrt(eps + bn_var)))
    wf = torch.mm(w_bn, w.reshape(co, -1)).view(w.shape)
    b_bn = bn_b - bn_w.mul(bn_mean).div(torch.sqrt(bn_var + eps))
    bf = torch.mm(w_bn, torch.zeros(co, 1, dtype=w.dtype)).reshape(-1) + b_bn
    return wf, bf


def _rep_branch(w, bn_w, bn_b, bn_mean, bn_var, eps):
    """RepConv.fuse_conv_bn, common.py:562-586."""
    std = (bn_var + eps).sqrt()
    return w * (bn_w / std).reshape(-1, 1, 1, 1), bn_b - bn_mean * bn_w / std


def fuse_state_dict(sd: Dict[str, torch.Tensor], specs: Dict[str, tuple], eps: float = BN_EPS) -> Dict[str, Tuple[torch.Tensor, torch.Tensor]]:
    out: Dict[str, Tuple[torch.Tensor, torch.Tensor]] = {}
    sd = {k: v.detach().float().cpu() for k, v in sd.items()}
    for name in specs:
        if name.endswith(".conv"):
            base = name[:-len(".conv")]
            if base + ".conv.bias" in sd:                      # already fused checkpoint
                out[name] = (sd[name + ".weight"], sd[name + ".bias"])
            else:
                out[name] = fuse_conv_bn(sd[name + ".weight"], sd[base + ".bn.weight"], sd[base + ".bn.bias"],
                                         sd[base + ".bn.running_mean"], sd[base + ".bn.running_var"], eps)
        elif name.endswith(".rbr_reparam"):
            base = name[:-len(".rbr_reparam")]
            if name + ".weight" in sd:
                out[name] = (sd[name + ".weight"], sd[name + ".bias"])
                continue
            bn = lambda p: (sd[p + ".weight"], sd[p + ".bias"], sd[p + ".running_mean"], sd[p + ".running_var"])
            w3, b3 = _rep_branch(sd[base + ".rbr_dense.0.weight"], *bn(base + ".rbr_dense.1"), eps)
            w1, b1 = _rep_branch(sd[base + ".rbr_1x1.0.weight"], *bn(base + ".rbr_1x1.1"), eps)
            w, b = w3 + torch.nn.functional.pad(w1, [1, 1, 1, 1]), b3 + b1
            if base + ".rbr_identity.weight" in sd:           # only when c1 == c2 (not in yolov7.yaml)
                c = w3.shape[0]
                ident = torch.zeros(c, c, 1, 1)
                ident[torch.arange(c), torch.arange(c), 0, 0] = 1.0
                wi, bi = _rep_branch(ident, *bn(base + ".rbr_identity"), eps)
                w, b = w + torch.nn.functional.pad(wi, [1, 1, 1, 1]), b + bi
            out[name] = (w, b)
        else:                                                  # detect conv  model.<i>.m.<l>
            base, l = name.rsplit(".m.", 1)
            w, b = sd[name + ".weight"].clone(), sd[name + ".bias"].clone()
            ia, im = sd.get(f"{base}.ia.{l}.implicit"), sd.get(f"{base}.im.{l}.implicit")
            if ia is not None:                                 # IDetect.fuse, yolo.py:186-198
                c1, c2 = w.shape[0], w.shape[1]
                b = b + torch.matmul(w.reshape(c1, c2), ia.reshape(c2, 1)).squeeze(1)
                b = b * im.reshape(c1)
                w = w * im.transpose(0, 1)
            out[name] = (w, b)
    return out

"""ORACLE (test infrastructure, not product code): CPU restatement of the reference's
HaMeR forward -- ViT-H/16 backbone, transformer-decoder MANO head, rot6d, MANO LBS,
projection -- in plain fp32 PyTorch on the CPU.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this package; the product path (``hamer_yolo_amd``) never does.

Pinning: the reference has no golden vectors for this path.  The
functions here are checked (tools/gen_golden.py, run in the build container) against the
reference's own modules loaded by file path on seeded weights --
``hamer/hamer/models/backbones/vit.py``, ``components/pose_transformer.py``,
``hamer/hamer/utils/geometry.py`` -- and, for MANO LBS (reference dependency
``smplx==0.1.28``, absent from the tree), against the in-tree ``manopth`` ManoLayer
(rootnet/KeypointFusion/manopth/manopth/manolayer.py:112-276) on the same parameters;
the outputs of those runs are committed under tests/golden/.

Every function takes a flat ``state_dict`` keyed like the reference checkpoint.
``emu`` selects the 16-bit emulation used for tight kernel-level checks (True / "bf16": bfloat16,
"fp16": half, "fp8": the MXFP8 configuration): activations and weights are rounded exactly where
the HIP path rounds them, arithmetic stays fp32.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def _q(x: Tensor, emu) -> Tensor:
    """Round to the 16-bit operand type the HIP path uses: emu == "fp16" -> half, any other true value -> bfloat16."""
    if not emu:
        return x
    return x.to(torch.float16 if emu == "fp16" else torch.bfloat16).to(torch.float32)


def _linear(x: Tensor, w: Tensor, b: Optional[Tensor], emu) -> Tensor:
    return F.linear(_q(x, emu), _q(w, emu), b)


def _linear_fp8(x: Tensor, w: Tensor, b: Optional[Tensor]) -> Tensor:
    """The fp8 GEMMs of emu == "fp8" (BASELINE configs[4]; qkv, fc1, fc2): X in MXFP8 blocks of 32 along K, W in e4m3
    with one scale per output channel (oracle/fp8_ref.py), fp32 accumulation."""
    from . import fp8_ref as Q
    w8, ws = Q.quantize_weight(w)
    return F.linear(Q.fake_quant_mx8(x), Q.dequantize_weight(w8, ws), b)


# ----------------------------------------------------------------------------- ViT backbone
def patch_embed(sd, x: Tensor, vit, emu=False, prefix="backbone.") -> Tensor:
    """PatchEmbed.forward, vit.py:170-176: Conv2d(3, D, k16, s16, pad 2) -> flatten(2).transpose(1,2)."""
    w = sd[prefix + "patch_embed.proj.weight"]
    b = sd[prefix + "patch_embed.proj.bias"]
    y = F.conv2d(_q(x, emu), _q(w, emu), b, stride=vit.patch, padding=vit.pad)
    return y.flatten(2).transpose(1, 2)


def vit_attention(sd, h: Tensor, p: str, vit, emu=False) -> Tensor:
    """Attention.forward, vit.py:110-126."""
    B, N, C = h.shape
    if emu == "fp8":
        qkv = _q(_linear_fp8(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]), True)   # fp8 path: bf16 outputs
    else:
        qkv = _q(_linear(h, sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"], emu), emu)
    qkv = qkv.reshape(B, N, 3, vit.heads, -1).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    scale = vit.head_dim ** -0.5
    if emu:
        # the HIP kernel scales the fp32 scores (not q), rounds the unnormalised P to bf16 for
        # the PV MFMA and keeps the row sum in fp32
        s = (q @ k.transpose(-2, -1)) * scale
        pun = torch.exp(s - s.amax(dim=-1, keepdim=True))
        o = _q((_q(pun, emu) @ v) / pun.sum(dim=-1, keepdim=True), emu)
    else:
        q = q * scale
        attn = (q @ k.transpose(-2, -1)).softmax(dim=-1)
        o = attn @ v
    if emu == "fp8" and vit.head_dim == 80:
        # fp8 proj: the attention output is MXFP8 with heads widened to 96 columns (3 scale blocks per head, the last
        # half empty); quantising the zero-padded heads and dropping the padding again is the same arithmetic
        from . import fp8_ref as Q
        o32 = (_q(pun, True) @ v) / pun.sum(dim=-1, keepdim=True)             # (B, H, N, 80) fp32, before any rounding
        opad = torch.zeros(B, vit.heads, N, 96)
        opad[..., :80] = o32
        oq = Q.fake_quant_mx8(opad)[..., :80].transpose(1, 2).reshape(B, N, -1)
        w8, ws = Q.quantize_weight(sd[p + "attn.proj.weight"])
        return F.linear(oq, Q.dequantize_weight(w8, ws), sd[p + "attn.proj.bias"])
    o = o.transpose(1, 2).reshape(B, N, -1)
    return _linear(o, sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"], emu)


def vit_mlp(sd, h: Tensor, p: str, emu=False) -> Tensor:
    """Mlp.forward, vit.py:82-87 (exact-erf GELU)."""
    if emu == "fp8":
        y = F.gelu(_linear_fp8(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        return _linear_fp8(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    y = F.gelu(_linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"], emu))
    return _linear(y, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"], emu)


def vit_forward(sd, x: Tensor, vit, emu=False, prefix="backbone.") -> Tensor:
    """ViT.forward_features, vit.py:320-339.  x: (B,3,256,192) -> (B,192,D)."""
    D = vit.embed_dim
    t = patch_embed(sd, x, vit, emu, prefix)
    pos = sd[prefix + "pos_embed"]
    t = t + pos[:, 1:] + pos[:, :1]
    for i in range(vit.depth):
        p = f"{prefix}blocks.{i}."
        h = F.layer_norm(t, (D,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], vit.ln_eps)
        t = t + vit_attention(sd, h, p, vit, emu)
        h = F.layer_norm(t, (D,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], vit.ln_eps)
        t = t + vit_mlp(sd, h, p, emu)
    return F.layer_norm(t, (D,), sd[prefix + "last_norm.weight"], sd[prefix + "last_norm.bias"], vit.ln_eps)


# ----------------------------------------------------------------------------- decoder head
def decoder_forward(sd, ctx: Tensor, dec, emu=False, prefix="mano_head.transformer.") -> Tensor:
    """TransformerDecoder.forward (pose_transformer.py:349-357) on the zero token
    (mano_head.py:86): 6 x [PreNorm self-attn, PreNorm cross-attn, PreNorm FF] (:191-201)."""
    B = ctx.shape[0]
    dim, Hh, dh = dec.dim, dec.heads, dec.dim_head
    token = torch.zeros(B, 1, 1, dtype=ctx.dtype)
    x = F.linear(token, sd[prefix + "to_token_embedding.weight"], sd[prefix + "to_token_embedding.bias"])
    x = x + sd[prefix + "pos_embedding"][:, :1]
    scale = dh ** -0.5
    for i in range(dec.depth):
        p = f"{prefix}transformer.layers.{i}."
        # self attention (pose_transformer.py:75-86)
        h = F.layer_norm(x, (dim,), sd[p + "0.norm.weight"], sd[p + "0.norm.bias"], dec.ln_eps)
        q, k, v = F.linear(h, sd[p + "0.fn.to_qkv.weight"]).chunk(3, dim=-1)
        sp = lambda t_: t_.reshape(B, -1, Hh, dh).transpose(1, 2)
        q, k, v = sp(q), sp(k), sp(v)
        a = ((q @ k.transpose(-1, -2)) * scale).softmax(dim=-1)
        o = (a @ v).transpose(1, 2).reshape(B, -1, Hh * dh)
        x = F.linear(o, sd[p + "0.fn.to_out.0.weight"], sd[p + "0.fn.to_out.0.bias"]) + x
        # cross attention (pose_transformer.py:111-124), to_kv has no bias
        h = F.layer_norm(x, (dim,), sd[p + "1.norm.weight"], sd[p + "1.norm.bias"], dec.ln_eps)
        kv = _linear(ctx, sd[p + "1.fn.to_kv.weight"], None, emu)
        k, v = kv.chunk(2, dim=-1)
        q = F.linear(h, sd[p + "1.fn.to_q.weight"])
        q, k, v = sp(q), sp(k), sp(v)
        a = ((q @ k.transpose(-1, -2)) * scale).softmax(dim=-1)
        o = (a @ v).transpose(1, 2).reshape(B, -1, Hh * dh)
        x = F.linear(o, sd[p + "1.fn.to_out.0.weight"], sd[p + "1.fn.to_out.0.bias"]) + x
        # feed forward (pose_transformer.py:40-52)
        h = F.layer_norm(x, (dim,), sd[p + "2.norm.weight"], sd[p + "2.norm.bias"], dec.ln_eps)
        h = F.gelu(F.linear(h, sd[p + "2.fn.net.0.weight"], sd[p + "2.fn.net.0.bias"]))
        x = F.linear(h, sd[p + "2.fn.net.3.weight"], sd[p + "2.fn.net.3.bias"]) + x
    return x.squeeze(1)


def mano_head_forward(sd, ctx: Tensor, dec, emu=False) -> Tuple[Tensor, Tensor, Tensor]:
    """MANOTransformerDecoderHead.forward, mano_head.py:61-95 (IEF_ITERS = 1).
    Returns pose6d (B,96), betas (B,10), cam (B,3)."""
    t = decoder_forward(sd, ctx, dec, emu)
    pose = F.linear(t, sd["mano_head.decpose.weight"], sd["mano_head.decpose.bias"]) + sd["mano_head.init_hand_pose"]
    betas = F.linear(t, sd["mano_head.decshape.weight"], sd["mano_head.decshape.bias"]) + sd["mano_head.init_betas"]
    cam = F.linear(t, sd["mano_head.deccam.weight"], sd["mano_head.deccam.bias"]) + sd["mano_head.init_cam"]
    return pose, betas, cam


def rot6d_to_rotmat(x: Tensor) -> Tensor:
    """geometry.py:47-70: a1 = x[0:3], a2 = x[3:6]; Gram-Schmidt; columns (b1, b2, b1 x b2)."""
    x = x.reshape(-1, 2, 3)
    a1, a2 = x[:, 0], x[:, 1]
    b1 = a1 / a1.norm(dim=1, keepdim=True).clamp_min(1e-12)
    u = a2 - (b1 * a2).sum(1, keepdim=True) * b1
    b2 = u / u.norm(dim=1, keepdim=True).clamp_min(1e-12)
    b3 = torch.linalg.cross(b1, b2, dim=1)
    return torch.stack((b1, b2, b3), dim=-1)


# ----------------------------------------------------------------------------- MANO
MANO_TIP_VERTS = [744, 320, 443, 554, 671]   # smplx vertex_ids['mano'], used at mano_wrapper.py:23
MANO_JOINT_MAP = [0, 13, 14, 15, 16, 1, 2, 3, 17, 4, 5, 6, 18, 10, 11, 12, 19, 7, 8, 9, 20]  # mano_wrapper.py:24


def mano_forward(mp: Dict[str, Tensor], betas: Tensor, rotmats: Tensor) -> Tuple[Tensor, Tensor]:
    """smplx.lbs.lbs (smplx==0.1.28, called with pose2rot=False from mano_wrapper.py:36) followed by
    the HaMeR wrapper's joint assembly (mano_wrapper.py:37-39).  Same arithmetic as the in-tree
    manopth layer (manolayer.py:172-262): blend shapes, joint regression, (R - I) pose map,
    kinematic chain, A - pack(A J), T = W A, v = T [v_posed; 1].
    betas (B,10), rotmats (B,16,3,3) -> vertices (B,V,3), joints (B,21,3) in metres."""
    B = betas.shape[0]
    vt, sdirs, pdirs = mp["v_template"], mp["shapedirs"], mp["posedirs"]
    Jr, W, parents = mp["J_regressor"], mp["lbs_weights"], [int(i) for i in mp["parents"]]
    V = vt.shape[0]
    v_shaped = vt[None] + torch.einsum("bl,mkl->bmk", betas, sdirs)
    J = torch.einsum("bik,ji->bjk", v_shaped, Jr)
    ident = torch.eye(3, dtype=betas.dtype)
    pose_feature = (rotmats[:, 1:] - ident).reshape(B, -1)
    v_posed = v_shaped + (pose_feature @ pdirs).reshape(B, V, 3)
    rel = J.clone()
    rel[:, 1:] = J[:, 1:] - J[:, parents[1:]]
    tm = torch.zeros(B, 16, 4, 4, dtype=betas.dtype)
    tm[:, :, :3, :3] = rotmats
    tm[:, :, :3, 3] = rel
    tm[:, :, 3, 3] = 1.0
    chain = [tm[:, 0]]
    for i in range(1, 16):
        chain.append(chain[parents[i]] @ tm[:, i])
    G = torch.stack(chain, dim=1)
    posed_joints = G[:, :, :3, 3]
    Jh = F.pad(J, (0, 1)).unsqueeze(-1)                    # (B,16,4,1), homogeneous coord 0
    A = G - F.pad(G @ Jh, (3, 0))
    T = (W @ A.reshape(B, 16, 16)).reshape(B, V, 4, 4)
    vh = F.pad(v_posed, (0, 1), value=1.0).unsqueeze(-1)
    verts = (T @ vh)[:, :, :3, 0]
    joints = torch.cat([posed_joints, verts[:, MANO_TIP_VERTS]], dim=1)[:, MANO_JOINT_MAP]
    return verts, joints


def perspective_projection(points: Tensor, translation: Tensor, focal_length: Tensor,
                           camera_center: Optional[Tensor] = None) -> Tensor:
    """geometry.py:72-110 with identity rotation."""
    B = points.shape[0]
    if camera_center is None:
        camera_center = torch.zeros(B, 2, dtype=points.dtype)
    p = points + translation.unsqueeze(1)
    p = p / p[:, :, -1:].clone()
    return p[:, :, :2] * focal_length.unsqueeze(1) + camera_center.unsqueeze(1) * p[:, :, 2:]


def hamer_forward(sd, mp, img: Tensor, cfg, emu=False) -> Dict[str, Tensor]:
    """HAMER.forward_step, hamer.py:99-156.  img: (B,3,256,256) normalised fp32."""
    B = img.shape[0]
    x = img[:, :, :, 32:-32]
    feats = vit_forward(sd, x, cfg.vit, emu)
    ctx = _q(feats, emu)
    pose6d, betas, cam = mano_head_forward(sd, ctx, cfg.dec, emu)
    R = rot6d_to_rotmat(pose6d).view(B, 16, 3, 3)
    focal = cfg.focal_length * torch.ones(B, 2)
    cam_t = torch.stack([cam[:, 1], cam[:, 2], 2 * focal[:, 0] / (cfg.image_size * cam[:, 0] + 1e-9)], dim=-1)
    verts, joints = mano_forward(mp, betas, R)
    kp2d = perspective_projection(joints, cam_t, focal / cfg.image_size)
    return {
        "tokens": feats, "pose6d": pose6d, "pred_cam": cam, "pred_cam_t": cam_t, "focal_length": focal,
        "pred_keypoints_3d": joints, "pred_vertices": verts, "pred_keypoints_2d": kp2d,
        "global_orient": R[:, :1], "hand_pose": R[:, 1:], "betas": betas,
    }

#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own modules (read in place from
/root/reference, never copied) on seeded synthetic weights.  Build-container only: the GPU
box has no /root/reference; only the small .npz outputs travel.

What is loaded from the reference (by file path, under a synthetic package so the relative
imports resolve):
  hamer/hamer/models/backbones/vit.py           -> ViT (vit.py:207-348)
  hamer/hamer/models/components/t_cond_mlp.py   -> normalization_layer
  hamer/hamer/models/components/pose_transformer.py -> TransformerDecoder (:301-357)
  hamer/hamer/utils/geometry.py                 -> rot6d_to_rotmat, perspective_projection
  rootnet/KeypointFusion/manopth/manopth/{manolayer,tensutils,rodrigues_layer}.py
                                                -> ManoLayer.forward (LBS arithmetic)
Import-time shims (no arithmetic on the inference path): ``timm.models.layers``
{to_2tuple, trunc_normal_, drop_path (identity in eval)} and an empty stand-in for manopth's
chumpy-based pickle reader (ManoLayer.__init__ is bypassed; buffers are set directly).

Usage:  python tools/gen_golden.py [--full]     (--full also runs the ViT-H geometry, ~1 min)
"""
import argparse
import importlib.util
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get("HAMER_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

from hamer_yolo_amd import synth  # noqa: E402


def _load(modname, path, package=None):
    spec = importlib.util.spec_from_file_location(modname, path)
    mod = importlib.util.module_from_spec(spec)
    if package:
        mod.__package__ = package
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def load_reference_modules():
    # timm shim: names only (vit.py:10)
    timm = types.ModuleType("timm"); timm_m = types.ModuleType("timm.models"); timm_l = types.ModuleType("timm.models.layers")
    timm_l.to_2tuple = lambda x: x if isinstance(x, tuple) else (x, x)
    timm_l.trunc_normal_ = torch.nn.init.trunc_normal_
    timm_l.drop_path = lambda x, p=0.0, training=False: x
    sys.modules.update({"timm": timm, "timm.models": timm_m, "timm.models.layers": timm_l})
    for pkg in ("refhamer", "refhamer.models", "refhamer.models.backbones", "refhamer.models.components", "refhamer.utils"):
        m = types.ModuleType(pkg); m.__path__ = []; sys.modules[pkg] = m
    base = os.path.join(REF, "hamer", "hamer")
    vit = _load("refhamer.models.backbones.vit", os.path.join(base, "models/backbones/vit.py"), "refhamer.models.backbones")
    _load("refhamer.models.components.t_cond_mlp", os.path.join(base, "models/components/t_cond_mlp.py"), "refhamer.models.components")
    pt = _load("refhamer.models.components.pose_transformer", os.path.join(base, "models/components/pose_transformer.py"), "refhamer.models.components")
    geo = _load("refhamer.utils.geometry", os.path.join(base, "utils/geometry.py"), "refhamer.utils")
    return vit, pt, geo


def load_reference_tome():
    """selective_vit_adapter.py (ToMe): ``from .vit import Attention, Block, ViT`` resolves to the module loaded above."""
    base = os.path.join(REF, "hamer", "hamer")
    return _load("refhamer.models.backbones.selective_vit_adapter", os.path.join(base, "models/backbones/selective_vit_adapter.py"),
                 "refhamer.models.backbones")


def load_manopth():
    base = os.path.join(REF, "rootnet/KeypointFusion/manopth")
    for pkg, path in (("refmano", []), ("refmano.manopth", [os.path.join(base, "manopth")]),
                      ("refmano.mano", []), ("refmano.mano.webuser", [])):
        m = types.ModuleType(pkg); m.__path__ = path; sys.modules[pkg] = m
    stub = types.ModuleType("refmano.mano.webuser.smpl_handpca_wrapper_HAND_only")
    stub.ready_arguments = None   # chumpy pickle reader: never called (ManoLayer.__init__ bypassed)
    sys.modules[stub.__name__] = stub
    return importlib.import_module("refmano.manopth.manolayer")


def build_ref_vit(vitmod, cfg, sd):
    v = cfg.vit
    m = vitmod.ViT(img_size=(v.img_h, v.img_w), patch_size=v.patch, embed_dim=v.embed_dim, depth=v.depth,
                   num_heads=v.heads, ratio=1, use_checkpoint=False, mlp_ratio=v.mlp_ratio, qkv_bias=True,
                   drop_path_rate=0.55)
    missing = m.load_state_dict({k[len("backbone."):]: t for k, t in sd.items() if k.startswith("backbone.")}, strict=True)
    torch.nn.Module.train(m, False)   # ViT.train() returns None (vit.py:345-348), so call the base method
    return m


def build_ref_decoder(ptmod, cfg, sd):
    d = cfg.dec
    m = ptmod.TransformerDecoder(num_tokens=1, token_dim=1, dim=d.dim, depth=d.depth, heads=d.heads, mlp_dim=d.mlp_dim,
                                 dim_head=d.dim_head, dropout=0.0, emb_dropout=0.0, norm="layer", context_dim=d.context_dim)
    pre = "mano_head.transformer."
    m.load_state_dict({k[len(pre):]: t for k, t in sd.items() if k.startswith(pre)}, strict=True)
    return m.eval()


def run_reference_hamer(cfg, sd, img, vitmod, ptmod, geo):
    """HAMER.forward_step (hamer.py:99-156) assembled from the reference sub-modules that are importable."""
    with torch.no_grad():
        vit = build_ref_vit(vitmod, cfg, sd)
        feats = vit(img[:, :, :, 32:-32])
        dec = build_ref_decoder(ptmod, cfg, sd)
        B = img.shape[0]
        tok = dec(torch.zeros(B, 1, 1), context=feats).squeeze(1)          # mano_head.py:86-90
        F = torch.nn.functional
        pose = F.linear(tok, sd["mano_head.decpose.weight"], sd["mano_head.decpose.bias"]) + sd["mano_head.init_hand_pose"]
        betas = F.linear(tok, sd["mano_head.decshape.weight"], sd["mano_head.decshape.bias"]) + sd["mano_head.init_betas"]
        cam = F.linear(tok, sd["mano_head.deccam.weight"], sd["mano_head.deccam.bias"]) + sd["mano_head.init_cam"]
        R = geo.rot6d_to_rotmat(pose).view(B, 16, 3, 3)                     # mano_head.py:110
    return feats, tok, pose, betas, cam, R


def manopth_forward(mlmod, mp, betas, axisang):
    """Drive ManoLayer.forward (manolayer.py:112-276) with axis-angle input on the given parameters."""
    layer = mlmod.ManoLayer.__new__(mlmod.ManoLayer)
    torch.nn.Module.__init__(layer)
    layer.center_idx = None; layer.robust_rot = False; layer.rot = 3; layer.flat_hand_mean = True
    layer.side = "right"; layer.use_pca = False; layer.joint_rot_mode = "axisang"; layer.root_rot_mode = "axisang"
    layer.ncomps = 45
    V = mp["v_template"].shape[0]
    layer.register_buffer("th_shapedirs", mp["shapedirs"].clone())                       # (V,3,10)
    layer.register_buffer("th_posedirs", mp["posedirs"].t().reshape(V, 3, 135).clone())  # (V,3,135)
    layer.register_buffer("th_v_template", mp["v_template"][None].clone())
    layer.register_buffer("th_J_regressor", mp["J_regressor"].clone())
    layer.register_buffer("th_weights", mp["lbs_weights"].clone())
    layer.register_buffer("th_hands_mean", torch.zeros(1, 45))
    layer.register_buffer("th_betas", torch.zeros(1, 10))
    with torch.no_grad():
        verts_mm, jtr_mm = layer(axisang, th_betas=betas)
    return verts_mm / 1000.0, jtr_mm / 1000.0


def rodrigues(aa):
    """Axis-angle -> rotation matrix in float64 (independent of both implementations under test)."""
    aa = aa.double()
    th = aa.norm(dim=-1, keepdim=True).clamp_min(1e-12)
    k = aa / th
    K = torch.zeros(*aa.shape[:-1], 3, 3, dtype=torch.float64)
    K[..., 0, 1], K[..., 0, 2], K[..., 1, 0] = -k[..., 2], k[..., 1], k[..., 2]
    K[..., 1, 2], K[..., 2, 0], K[..., 2, 1] = -k[..., 0], -k[..., 1], k[..., 0]
    s, c = torch.sin(th)[..., None], torch.cos(th)[..., None]
    return torch.eye(3, dtype=torch.float64) + s * K + (1 - c) * (K @ K)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    vitmod, ptmod, geo = load_reference_modules()

    # ---- tiny geometry: every intermediate
    cfg = synth.tiny_config()
    sd = synth.hamer_state_dict(cfg, seed=1)          # fp32 master weights, as a real checkpoint holds them (models/__init__.py:46)
    img = synth.normalize_crops(synth.crops_u8(3, seed0=10))
    feats, tok, pose, betas, cam, R = run_reference_hamer(cfg, sd, img, vitmod, ptmod, geo)
    np.savez_compressed(os.path.join(OUT, "hamer_tiny.npz"), seed=1, crop_seed0=10, bf16_representable=0,
                        tokens=feats.numpy(), token_out=tok.numpy(), pose6d=pose.numpy(), betas=betas.numpy(),
                        cam=cam.numpy(), rotmats=R.numpy())
    print("tiny: tokens", feats.shape, "pose", pose.shape)

    # ---- geometry helpers on seeded inputs
    x6 = synth.uniform("golden.rot6d", (32, 6), 1.0, seed=3)
    pts = synth.uniform("golden.pts", (4, 21, 3), 0.1, seed=3)
    tr = synth.uniform("golden.tr", (4, 3), 0.2, seed=3) + torch.tensor([0, 0, 40.0])
    fl = torch.full((4, 2), 5000.0 / 256)
    with torch.no_grad():
        np.savez_compressed(os.path.join(OUT, "geometry.npz"), x6=x6.numpy(), rotmat=geo.rot6d_to_rotmat(x6).numpy(),
                            pts=pts.numpy(), tr=tr.numpy(), fl=fl.numpy(),
                            proj=geo.perspective_projection(pts, tr, fl).numpy())

    # ---- MANO LBS vs in-tree manopth on synthetic MANO-shaped parameters
    ml = load_manopth()
    mp = synth.mano_params(seed=5)
    aa = synth.uniform("golden.aa", (4, 48), 0.8, seed=5)
    bt = synth.uniform("golden.betas", (4, 10), 1.5, seed=5)
    verts, jtr = manopth_forward(ml, mp, bt, aa)
    Rm = rodrigues(aa.view(4, 16, 3)).float()
    np.savez_compressed(os.path.join(OUT, "mano_manopth.npz"), mano_seed=5, axisang=aa.numpy(), betas=bt.numpy(),
                        rotmats=Rm.numpy(), verts=verts.numpy(), joints16_tips_manopth=jtr.numpy())
    print("manopth verts", verts.shape, float(verts.abs().max()))

    # ---- container-only cross-check on the REAL MANO model (licensed arrays: nothing is saved)
    real = os.path.join(REF, "rootnet/KeypointFusion/MANO/MANO_RIGHT.pkl")
    if os.path.exists(real):
        from hamer_yolo_amd.hamer.models.mano_wrapper import MANO
        from oracle import hamer_ref as R
        rp = MANO.from_pkl(real).params
        v_ref, _ = manopth_forward(ml, rp, bt, aa)
        v_or, _ = R.mano_forward(rp, bt, Rm)
        err = float((v_ref - v_or).abs().max())
        print(f"real MANO_RIGHT.pkl: oracle LBS vs manopth max |dv| = {err:.2e} m")
        assert err < 5e-6

    # ---- token merging (HAMER_INFER(token_merge=True), hamer.py:481-483: apply_patch + r = (8, -1)) on a 6-block geometry
    tome = load_reference_tome()
    cfg = synth.tome_tiny_config()
    sd = synth.hamer_state_dict(cfg, seed=7)
    img = synth.normalize_crops(synth.crops_u8(3, seed0=70))
    with torch.no_grad():
        vit = build_ref_vit(vitmod, cfg, sd)
        tome.apply_patch(vit)                      # trace_source=False, prop_attn=True (the defaults HAMER_INFER uses)
        vit.r = (8, -1)
        feats = vit(img[:, :, :, 32:-32])
        dec = build_ref_decoder(ptmod, cfg, sd)
        tok = dec(torch.zeros(3, 1, 1), context=feats).squeeze(1)
        F = torch.nn.functional
        pose = F.linear(tok, sd["mano_head.decpose.weight"], sd["mano_head.decpose.bias"]) + sd["mano_head.init_hand_pose"]
        betas = F.linear(tok, sd["mano_head.decshape.weight"], sd["mano_head.decshape.bias"]) + sd["mano_head.init_betas"]
        cam = F.linear(tok, sd["mano_head.deccam.weight"], sd["mano_head.deccam.bias"]) + sd["mano_head.init_cam"]
        # the matching of the first block alone, on the metric the reference computes there
        metric = synth.uniform("golden.tome_metric", (2, 192, 80), 1.0, seed=9)
        merge, _ = tome.bipartite_soft_matching(metric, 16)
        xs = synth.uniform("golden.tome_x", (2, 192, 24), 1.0, seed=9)
        merged, msize = tome.merge_wavg(merge, xs)
    np.savez_compressed(os.path.join(OUT, "hamer_tome.npz"), seed=7, crop_seed0=70, r=np.array([8, -1]),
                        r_list=np.array(tome.parse_r(cfg.vit.depth, (8, -1))), r_list_vith=np.array(tome.parse_r(32, (8, -1))),
                        tokens=feats.numpy(), token_out=tok.numpy(), pose6d=pose.numpy(), betas=betas.numpy(), cam=cam.numpy(),
                        merged=merged.numpy(), merged_size=msize.numpy())
    print("tome: tokens", feats.shape)

    if args.full:
        cfg = synth.HamerConfig()
        sd = synth.hamer_state_dict(cfg, seed=0)
        img = synth.normalize_crops(synth.crops_u8(4, seed0=0))
        feats, tok, pose, betas, cam, R = run_reference_hamer(cfg, sd, img, vitmod, ptmod, geo)
        np.savez_compressed(os.path.join(OUT, "hamer_vith.npz"), seed=0, crop_seed0=0, bf16_representable=0,
                            tokens_sub=feats[:, ::16, ::40].numpy(), tokens_mean=feats.mean((1, 2)).numpy(),
                            tokens_absmean=feats.abs().mean((1, 2)).numpy(),
                            token_out=tok.numpy(), pose6d=pose.numpy(), betas=betas.numpy(), cam=cam.numpy(),
                            rotmats=R.numpy())
        print("full: tokens", feats.shape, float(feats.abs().mean()))


if __name__ == "__main__":
    main()

"""CPU: the oracle restatement against fixtures produced by the reference's own modules
(tools/gen_golden.py, run in the build container)."""
import os

import numpy as np
import torch

from hamer_yolo_amd import synth
from oracle import hamer_ref as R


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_tiny_hamer_matches_reference_modules(golden_dir):
    g = _load(golden_dir, "hamer_tiny.npz")
    cfg = synth.tiny_config()
    sd = synth.hamer_state_dict(cfg, seed=int(g["seed"]), bf16_representable=bool(g["bf16_representable"]))
    img = synth.normalize_crops(synth.crops_u8(3, seed0=int(g["crop_seed0"])))
    with torch.no_grad():
        feats = R.vit_forward(sd, img[:, :, :, 32:-32], cfg.vit)
        tok = R.decoder_forward(sd, feats, cfg.dec)
        pose, betas, cam = R.mano_head_forward(sd, feats, cfg.dec)
        rot = R.rot6d_to_rotmat(pose).view(3, 16, 3, 3)
    np.testing.assert_allclose(feats.numpy(), g["tokens"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(tok.numpy(), g["token_out"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(pose.numpy(), g["pose6d"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(betas.numpy(), g["betas"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(cam.numpy(), g["cam"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(rot.numpy(), g["rotmats"], atol=2e-5, rtol=0)


def test_geometry_matches_reference(golden_dir):
    g = _load(golden_dir, "geometry.npz")
    rot = R.rot6d_to_rotmat(torch.from_numpy(g["x6"]))
    np.testing.assert_allclose(rot.numpy(), g["rotmat"], atol=1e-6, rtol=0)
    proj = R.perspective_projection(torch.from_numpy(g["pts"]), torch.from_numpy(g["tr"]), torch.from_numpy(g["fl"]))
    np.testing.assert_allclose(proj.numpy(), g["proj"], atol=1e-5, rtol=0)


def test_mano_lbs_matches_manopth(golden_dir):
    g = _load(golden_dir, "mano_manopth.npz")
    mp = synth.mano_params(seed=int(g["mano_seed"]))
    verts, joints = R.mano_forward(mp, torch.from_numpy(g["betas"]), torch.from_numpy(g["rotmats"]))
    np.testing.assert_allclose(verts.numpy(), g["verts"], atol=2e-6, rtol=0)
    # manopth's 16 posed joints are in the same openpose order; its finger tips use other
    # vertex ids (manolayer.py:252-253), so compare the 16 regressed joints only.
    jm = g["joints16_tips_manopth"]
    non_tip = [i for i, j in enumerate(R.MANO_JOINT_MAP) if j < 16]
    np.testing.assert_allclose(joints.numpy()[:, non_tip], jm[:, non_tip], atol=2e-6, rtol=0)
    tips = [i for i, j in enumerate(R.MANO_JOINT_MAP) if j >= 16]
    np.testing.assert_allclose(joints.numpy()[:, tips], verts.numpy()[:, R.MANO_TIP_VERTS], atol=0, rtol=0)


def test_mano_rest_pose_is_template():
    mp = synth.mano_params(seed=2)
    eye = torch.eye(3).expand(2, 16, 3, 3).contiguous()
    verts, joints = R.mano_forward(mp, torch.zeros(2, 10), eye)
    np.testing.assert_allclose(verts.numpy(), mp["v_template"][None].expand(2, -1, -1).numpy(), atol=1e-6)


def test_rootnet_head_bbox_and_k_match_reference(golden_dir):
    """ResRootNet.forward, process_bbox and calculate_k of the reference (captured by tools/gen_golden_rootnet.py) against the
    oracle restatement and against the product's host mirror (hamer_yolo_amd/rootnet/preprocessing.py)."""
    import os
    import numpy as np
    import torch
    from hamer_yolo_amd.rootnet.preprocessing import process_bbox
    from oracle import rootnet_ref as RR
    g = np.load(os.path.join(golden_dir, "rootnet_head.npz"))
    W, H = int(g["img_wh"][0]), int(g["img_wh"][1])
    ks = []
    for b, want in zip(g["boxes"], g["processed"]):
        got_o = RR.process_bbox(list(b), W, H, (256, 256), 1.5)
        got_p = process_bbox(b.copy(), W, H, (256, 256), 1.5)
        if np.isnan(want[0]):
            assert got_o is None and got_p is None
            continue
        np.testing.assert_allclose(got_o, want, rtol=0, atol=1e-4)
        np.testing.assert_array_equal(got_p, want)            # the mirror keeps the reference's exact arithmetic
        ks.append(RR.calculate_k(want, float(g["fx_fy"][0]), float(g["fx_fy"][1])))
    np.testing.assert_allclose(ks, g["k_of_processed"], rtol=1e-6)
    root = {"depth_layer.weight": torch.from_numpy(g["depth_w"]), "depth_layer.bias": torch.from_numpy(g["depth_b"])}
    d = RR.root_depth(root, torch.from_numpy(g["feats"]), torch.from_numpy(g["k_value"]))
    np.testing.assert_allclose(d.numpy(), g["depth"], rtol=1e-5, atol=1e-7)


def test_tome_matches_reference_module(golden_dir):
    """Token merging: oracle/tome_ref.py against hamer/hamer/models/backbones/selective_vit_adapter.py run on the same seeded
    weights (apply_patch + r = (8, -1), as HAMER_INFER(token_merge=True) sets it, hamer.py:481-483)."""
    from oracle import tome_ref as T
    g = _load(golden_dir, "hamer_tome.npz")
    assert T.parse_r(6, (8, -1)) == list(g["r_list"]) and T.parse_r(32, (8, -1)) == list(g["r_list_vith"])
    assert T.parse_r(4, 5) == [5, 5, 5, 5] and T.parse_r(3, [7, 2]) == [7, 2, 0]
    # the matching + size-weighted merge alone
    metric = synth.uniform("golden.tome_metric", (2, 192, 80), 1.0, seed=9)
    xs = synth.uniform("golden.tome_x", (2, 192, 24), 1.0, seed=9)
    unm, src, dst = T.bipartite_match(metric, 16)
    merged, msize = T.merge_tokens(xs, torch.ones(2, 192, 1), unm, src, dst)
    np.testing.assert_allclose(merged.numpy(), g["merged"], atol=1e-6, rtol=0)
    np.testing.assert_array_equal(msize.numpy(), g["merged_size"])
    assert merged.shape == (2, 176, 24) and float(msize.sum()) == 2 * 192
    # the whole backbone + head
    cfg = synth.tome_tiny_config()
    sd = synth.hamer_state_dict(cfg, seed=int(g["seed"]))
    img = synth.normalize_crops(synth.crops_u8(3, seed0=int(g["crop_seed0"])))
    trace = {}
    with torch.no_grad():
        feats = T.vit_forward_tome(sd, img[:, :, :, 32:-32], cfg.vit, (8, -1), trace=trace)
        pose, betas, cam = R.mano_head_forward(sd, feats, cfg.dec)
    assert trace["tokens"] == [176, 164, 155, 149, 146, 146] and feats.shape == (3, 146, 320)
    np.testing.assert_allclose(feats.numpy(), g["tokens"], atol=3e-5, rtol=0)
    np.testing.assert_allclose(pose.numpy(), g["pose6d"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(betas.numpy(), g["betas"], atol=2e-5, rtol=0)
    np.testing.assert_allclose(cam.numpy(), g["cam"], atol=2e-5, rtol=0)

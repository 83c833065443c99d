"""GPU: the reference-shaped Python API end to end (hamer_inference.estimate_from_rgb, .npy records)
against the oracle pipeline (crop_ref -> hamer_ref -> the reference's camera formulas)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from hamer_yolo_amd import synth
from hamer_yolo_amd.infer import (axis_angle_to_rotation_matrix_torch, hamer_inference, hand_record,
                                  reconstruct_and_save_obj_with_wrapper)
from oracle import crop_ref
from oracle import hamer_ref as R


class _Cfg:
    ckpt_path = "synthetic:0"
    model_cfg = None
    use_onnx = False
    onnx_path = None


@pytest.fixture(scope="module")
def hi():
    return hamer_inference(_Cfg)


def _oracle_pipeline(frame, dets, k_real):
    cfg = synth.HamerConfig()
    sd = synth.hamer_state_dict(cfg, seed=0)
    mp = synth.mano_params(seed=0)
    mean = 255.0 * np.array([0.485, 0.456, 0.406]); std = 255.0 * np.array([0.229, 0.224, 0.225])
    batch = crop_ref.prepare_batch_bbox(frame, dets, mean, std)
    with torch.no_grad():
        o = R.hamer_forward(sd, mp, torch.from_numpy(batch["img"]), cfg)
    # estimate_from_rgb, infer.py:381-476 (k_real branch)
    do_flip = torch.from_numpy(batch["do_flip"])
    cam = o["pred_cam"].clone()
    cam[:, 1] *= 1.0 - 2.0 * do_flip
    fx, fy, cx, cy = k_real[0, 0], k_real[1, 1], k_real[0, 2], k_real[1, 2]
    bs = torch.from_numpy(batch["box_size"]) * cam[:, 0] + 1e-9
    tz = 2 * fx / bs
    tx = 2 * (torch.from_numpy(batch["box_center"])[:, 0] - cx) / bs + cam[:, 1]
    ty = 2 * (torch.from_numpy(batch["box_center"])[:, 1] - cy) / bs + cam[:, 2]
    if fx != fy:
        ty = ty * (fx / fy)
    cam_t_full = torch.stack([tx, ty, tz], -1)
    kp3d = o["pred_keypoints_3d"].clone()
    kp3d[:, :, 0] *= do_flip[:, None]
    kc = kp3d + cam_t_full[:, None]
    depth = kc[:, :, 2:3] + 1e-9
    kp2d = torch.cat([kc[:, :, 0:1] / depth * fx + cx, kc[:, :, 1:2] / depth * fy + cy], -1)
    return batch, o, cam_t_full, kp2d


def test_estimate_from_rgb_matches_oracle_pipeline(hi):
    frame = synth.frame_u8(720, 1280, seed=21).numpy()
    dets = [["right", [300.0, 200.0, 460.0, 380.0]], ["left", [700.0, 300.0, 820.0, 470.0]], ["left", [1150.0, 600.0, 1300.0, 740.0]]]
    k_real = np.array([[900.0, 0, 640.0], [0, 910.0, 360.0], [0, 0, 1]], dtype=np.float32)
    out, params = hi.estimate_from_rgb(frame, dets, k_real)
    batch, o, cam_t_full, kp2d = _oracle_pipeline(frame, dets, k_real)
    assert np.array_equal(out["img"].cpu().numpy(), batch["img"])                      # crop: bit-exact
    np.testing.assert_allclose(out["trans"].cpu().numpy(), batch["trans"], rtol=1e-6, atol=1e-5)
    assert torch.equal(out["trans"], out["inv_trans"]) and out["do_flip"].tolist() == [0.0, 1.0, 1.0]
    mp = out["pred_mano_params"]
    assert mp["global_orient"].shape == (3, 1, 3, 3) and mp["hand_pose"].shape == (3, 15, 3, 3) and mp["betas"].shape == (3, 10)
    np.testing.assert_allclose(mp["global_orient"].cpu().numpy(), o["global_orient"].numpy(), atol=1e-3)
    np.testing.assert_allclose(mp["hand_pose"].cpu().numpy(), o["hand_pose"].numpy(), atol=1e-3)
    np.testing.assert_allclose(mp["betas"].cpu().numpy(), o["betas"].numpy(), atol=1e-3)
    np.testing.assert_allclose(out["pred_vertices"].cpu().numpy(), o["pred_vertices"].numpy(), atol=1e-3)
    np.testing.assert_allclose(params["trans"].cpu().numpy(), o["pred_cam_t"].numpy(), rtol=2e-3, atol=1e-3)
    np.testing.assert_allclose(out["pred_cam_t_full"].cpu().numpy(), cam_t_full.numpy(), rtol=2e-3, atol=2e-3)
    np.testing.assert_allclose(out["pred_keypoints_2d_full"].cpu().numpy(), kp2d.numpy(), rtol=2e-3, atol=0.5)
    assert (out["pred_keypoints_3d"][0, :, 0] == 0).all()                               # the reference's do_flip quirk


def test_estimate_from_rgb_default_intrinsics_and_errors(hi):
    frame = synth.frame_u8(480, 640, seed=3).numpy()
    out, _ = hi.estimate_from_rgb(frame, [["right", [200.0, 100.0, 330.0, 260.0]]])
    f = 5000.0 / 256.0 * 640.0
    assert abs(float(out["focal_length"][0]) - f) < 1e-3 and out["pred_cam_t_full"].shape == (1, 3)
    with pytest.raises(ValueError):
        hi.estimate_from_rgb(frame, [])
    with pytest.raises(ValueError):
        hi.estimate_from_rgb(frame, [["right", [1.0, 2.0, 3.0]]])
    out2, _ = hi.estimate_from_rgb(frame, [["right", [200.0, 100.0, 330.0, 260.0]]], depth_refine=0.6)
    assert abs(float(out2["pred_cam_t_full"][0, 2]) - 0.6) < 1e-6


def test_npy_record_and_obj_reconstruction(hi, tmp_path):
    frame = synth.frame_u8(480, 640, seed=4).numpy()
    dets = [["right", [100.0, 100.0, 230.0, 260.0]], ["left", [350.0, 200.0, 470.0, 330.0]]]
    out, _ = hi.estimate_from_rgb(frame, dets, np.array([[600.0, 0, 320], [0, 600.0, 240], [0, 0, 1]], np.float32))
    rec = {d[0]: hand_record(out, d[0] == "right", i) for i, d in enumerate(dets)}
    for r in rec.values():
        assert r["betas"].shape == (10,) and r["theta"].shape == (48,) and r["pose_hand"].shape == (45,) and r["cam_t"].shape == (3,)
    # theta (axis-angle) -> rotation matrices reproduces the network's matrices (infer.py:65-83 round trip)
    Rm = axis_angle_to_rotation_matrix_torch(torch.from_numpy(rec["right"]["theta"].reshape(16, 3)))
    ref = torch.cat([out["pred_mano_params"]["global_orient"][0], out["pred_mano_params"]["hand_pose"][0]], 0).cpu()
    np.testing.assert_allclose(Rm.numpy(), ref.numpy(), atol=2e-5)
    np.save(tmp_path / "img0.npy", rec)
    reconstruct_and_save_obj_with_wrapper(str(tmp_path), str(tmp_path / "obj"), hi)
    lines = open(tmp_path / "obj" / "img0.obj").read().splitlines()
    v = np.array([[float(t) for t in l.split()[1:]] for l in lines if l.startswith("v ")])
    assert v.shape == (2 * 778, 3) and sum(l.startswith("f ") for l in lines) == 2 * 1538
    np.testing.assert_allclose(v[:778], out["pred_vertices"][0].cpu().numpy() + rec["right"]["cam_t"], atol=2e-4)
    left = out["pred_vertices"][1].cpu().numpy().copy(); left[:, 0] *= -1
    np.testing.assert_allclose(v[778:], left + rec["left"]["cam_t"], atol=2e-4)


class _FixedDetector:
    """Detector stand-in for the batch drivers: the boxes are given (the detector has its own tests)."""

    def __init__(self, dets):
        self.dets = dets

    def detect(self, image):
        return [None], [self.dets]


def test_process_batch_npz_and_mask_driver(hi, tmp_path):
    """The two remaining savers of infer.py: process_batch (:908-1036, rotation matrices in one .npz per hand, suffix rule
    for repeated labels) and process_batch_manopara_with_mask (:1099-1220, box from label 3 of a mask, per-frame
    intrinsics), plus the OBJ -> reprojection helpers of hamer/reconstruct.py."""
    from PIL import Image
    from hamer_yolo_amd.hamer.reconstruct import load_obj, project_vertices
    from hamer_yolo_amd.infer import get_bbox_from_npy, process_batch, process_batch_manopara_with_mask
    img_dir, out_dir, mask_dir, k_dir = tmp_path / "rgb", tmp_path / "out", tmp_path / "mask", tmp_path / "k"
    for d in (img_dir, mask_dir, k_dir):
        d.mkdir()
    frame = synth.frame_u8(480, 640, seed=5).numpy()
    Image.fromarray(frame[:, :, ::-1]).save(img_dir / "f0.png")
    dets = [["right", [100.0, 120.0, 260.0, 300.0]], ["left", [380.0, 200.0, 520.0, 330.0]], ["right", [300.0, 20.0, 420.0, 150.0]]]
    K = np.array([[600.0, 0, 320], [0, 610.0, 240], [0, 0, 1]], np.float32)
    process_batch(str(img_dir), str(out_dir), K, hamer=hi, detector=_FixedDetector(dets))
    names = sorted(os.listdir(out_dir))
    assert names == ["f0_left.npz", "f0_right.npz", "f0_right_2.npz"]
    out, _ = hi.estimate_from_rgb(frame, dets, K)
    for name, i in (("f0_right.npz", 0), ("f0_left.npz", 1), ("f0_right_2.npz", 2)):
        z = np.load(out_dir / name)
        assert z["betas"].shape == (1, 10) and z["global_orient"].shape == (1, 1, 3, 3) and z["hand_pose"].shape == (1, 15, 3, 3)
        assert z["cam_t"].shape == (1, 3) and bool(z["is_right"]) == (dets[i][0] == "right")
        np.testing.assert_allclose(z["hand_pose"][0], out["pred_mano_params"]["hand_pose"][i].cpu().numpy(), atol=1e-6)
        np.testing.assert_allclose(z["cam_t"][0], out["pred_cam_t_full"][i].cpu().numpy(), atol=1e-6)

    # mask-driven: label 3 occupies rows 150..280, cols 200..330 -> box [200, 150, 330, 280]
    mask = np.zeros((480, 640), np.uint8)
    mask[150:281, 200:331] = 3
    mask[10:20, 10:20] = 2
    np.save(mask_dir / "f0.npy", mask)
    assert get_bbox_from_npy(str(mask_dir / "f0.npy")) == [200.0, 150.0, 330.0, 280.0]
    assert get_bbox_from_npy(str(mask_dir / "f0.npy"), target_val=7) is None
    assert get_bbox_from_npy(str(mask_dir / "missing.npy")) is None
    np.savetxt(k_dir / "f0.txt", K)
    mout = tmp_path / "mout"
    process_batch_manopara_with_mask(str(img_dir), str(mask_dir), str(mout), str(k_dir), hamer=hi)
    rec = np.load(mout / "f0.npy", allow_pickle=True).item()
    assert rec["left"] is None and rec["right"]["is_right"] is True
    ref, _ = hi.estimate_from_rgb(frame, [["right", [200.0, 150.0, 330.0, 280.0]]], K)
    want = hand_record(ref, True, 0)
    for k in ("betas", "theta", "cam_t"):
        np.testing.assert_allclose(rec["right"][k], want[k], atol=1e-6)

    # .npy -> OBJ -> reprojection: every projected vertex of the reconstructed mesh lands where the network's
    # camera-frame vertices project
    reconstruct_and_save_obj_with_wrapper(str(mout), str(tmp_path / "obj"), hi)
    v, f = load_obj(str(tmp_path / "obj" / "f0.obj"))
    assert v.shape == (778, 3) and f.shape == (1538, 3) and f.min() == 0 and f.max() == 777
    px, order = project_vertices(v, f, K)
    cam = ref["pred_vertices"][0].cpu().numpy().astype(np.float64) + want["cam_t"]
    uv = (K.astype(np.float64) @ cam.T).T
    np.testing.assert_allclose(px, (uv[:, :2] / uv[:, 2:3]).astype(np.int32), atol=1)
    assert sorted(order.tolist()) == list(range(1538))
    assert load_obj(str(tmp_path / "nope.obj")) == (None, None)


def test_one_shared_set_of_streams_per_device():
    """HIP deals streams onto a few hardware queues in creation order and two streams on one queue do not overlap (3 of 15 pairs,
    tools/probes/stream_pairs.py), so the package keeps ONE set per device: the engine's contexts and the folder drivers get the
    same stream objects, and 'cuda' / 'cuda:0' name the same set."""
    import torch
    from hamer_yolo_amd import infer, streams, synth
    from hamer_yolo_amd.engine import HamerEngine
    a = streams.get_streams("cuda", 2)
    b = streams.get_streams(torch.device("cuda", torch.cuda.current_device()), 2)
    assert a[0] is b[0] and a[1] is b[1] and a[0] is not a[1]
    assert streams.get_streams("cuda:0", 3)[:2] == a
    cfg = synth.tiny_config()
    eng = HamerEngine(synth.hamer_state_dict(cfg, seed=0), synth.mano_params(seed=0), cfg, device="cuda:0")
    ctxs = eng.contexts(2, 2)
    assert ctxs[0].stream is a[0] and ctxs[1].stream is a[1]
    assert infer._driver_streams(torch.device("cuda"), 2) == a

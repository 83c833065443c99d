"""Pipeline driver with the reference's entry-point surface (reference: hamer/infer.py;
``d_infer.py`` differs only by the ``depth_refine`` argument, which ``estimate_from_rgb`` accepts).

Kept in Python (drop-in): ``hamer_inference`` (:117-528), ``matrix_to_axis_angle`` (:1082-1096),
``axis_angle_to_rotation_matrix_torch`` (:65-83), ``process_batch_manopara`` (:1223-1318),
``reconstruct_and_save_obj_with_wrapper`` (:1321-1436), ``load_intrinsics`` (:1458-1477) and the CLI
(:1479-1536).  Pixel and tensor work is HIP: one ``hm_crop_batch`` launch for all hands of a frame
(the reference crops hand by hand on the CPU and copies the frame per hand, :208) and one
``hm_hamer_forward`` enqueue for the batch.  Out of scope: ONNX export/compare, pyrender overlays.
"""
from __future__ import annotations

import argparse
import glob
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import ops
from .config.hamer_config import hamer_opt
from .hamer.datasets.utils import expand_to_aspect_ratio, gen_trans_from_patch_cv
from .hamer.models import load_hamer
from .hamer.utils.geometry import perspective_projection
from .hamer.utils.renderer import custom_cam_crop_to_full

DEFAULT_MEAN = 255. * np.array([0.485, 0.456, 0.406])
DEFAULT_STD = 255. * np.array([0.229, 0.224, 0.225])


def axis_angle_to_rotation_matrix_torch(rvec_tensor: torch.Tensor) -> torch.Tensor:
    """infer.py:65-83 (Rodrigues, theta + 1e-8)."""
    theta = torch.norm(rvec_tensor, dim=1, keepdim=True) + 1e-8
    r_hat = rvec_tensor / theta
    cos = torch.cos(theta)
    z = torch.zeros(theta.shape[0], dtype=rvec_tensor.dtype, device=rvec_tensor.device)
    m = torch.stack([z, -r_hat[:, 2], r_hat[:, 1], r_hat[:, 2], z, -r_hat[:, 0], -r_hat[:, 1], r_hat[:, 0], z],
                    dim=1).reshape(-1, 3, 3)
    eye = torch.eye(3, dtype=rvec_tensor.dtype, device=rvec_tensor.device).unsqueeze(0).expand(rvec_tensor.shape[0], -1, -1)
    A = r_hat.unsqueeze(2) * r_hat.unsqueeze(1)
    return cos.unsqueeze(2) * eye + (1 - cos.unsqueeze(2)) * A + torch.sin(theta).unsqueeze(2) * m


def rodrigues_log(R: np.ndarray) -> np.ndarray:
    """Rotation matrix -> axis-angle (what cv2.Rodrigues(R)[0] returns; cv2 is not a dependency).
    Uses the antisymmetric part away from pi and the symmetric part near pi."""
    R = np.asarray(R, dtype=np.float64)
    c = np.clip((np.trace(R) - 1.0) * 0.5, -1.0, 1.0)
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]]) * 0.5
    s = np.linalg.norm(r)
    theta = np.arctan2(s, c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3, dtype=np.float32) if s == 0 else (r * (theta / s)).astype(np.float32)
        t = (np.diag(R) + 1.0) * 0.5
        v = np.sqrt(np.maximum(t, 0.0))
        k = int(np.argmax(v))
        sign = np.sign(np.array([R[k, 0] + R[0, k], R[k, 1] + R[1, k], R[k, 2] + R[2, k]]))
        sign[sign == 0] = 1.0
        v = v * sign
        return (v / np.linalg.norm(v) * theta).astype(np.float32)
    return (r * (theta / s)).astype(np.float32)


def matrix_to_axis_angle(rot_mats) -> np.ndarray:
    """infer.py:1082-1096: (N,3,3) or (3,3) -> flattened (N*3,) axis-angle."""
    if isinstance(rot_mats, np.ndarray) and rot_mats.ndim == 2:
        rot_mats = [rot_mats]
    return np.concatenate([rodrigues_log(m).flatten() for m in rot_mats])


class hamer_inference():
    def __init__(self, cfg=hamer_opt):
        """infer.py:118-146.  ``cfg.{ckpt_path, model_cfg, use_onnx, onnx_path}``."""
        self.use_onnx = bool(getattr(cfg, "use_onnx", False))
        if self.use_onnx:
            raise NotImplementedError("the ONNX Runtime path of the reference is out of scope (no ONNX export path)")
        self.device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        model, model_cfg_obj = load_hamer(cfg.ckpt_path)
        self.model = model.to(self.device)      # raises on a machine without an MI355X: no CPU fallback
        self.model.eval()
        self.cfg = model_cfg_obj
        self.ort_session = None
        self.mean = 255. * np.array(self.cfg.MODEL.IMAGE_MEAN)
        self.std = 255. * np.array(self.cfg.MODEL.IMAGE_STD)
        self.mano = self.model.mano

    def get_mesh_renderer(self):
        """infer.py:148-152 builds pyrender's MeshRenderer; rendering is out of scope, the MANO model
        (with ``.faces``) is what reconstruct_and_save_obj_with_wrapper needs."""
        self.mano = self.model.mano
        return None

    # ------------------------------------------------------------------ crop
    def prepare_batch_bbox(self, img_0: np.ndarray, bboxs: List) -> Dict[str, torch.Tensor]:
        """infer.py:154-259.  img_0: HxWx3 uint8 BGR; bboxs: [[label, [x1, y1, x2, y2]], ...]."""
        P = int(self.cfg.MODEL.IMAGE_SIZE)
        boxes, centers, sizes, flips, transs = [], [], [], [], []
        for bbox in bboxs:
            if not (isinstance(bbox, list) and len(bbox) == 2 and isinstance(bbox[1], list) and len(bbox[1]) == 4):
                raise ValueError(f"Invalid bbox format: Expected [class, [x1,y1,x2,y2]], got {bbox}")
            hand_cls, (x1, y1, x2, y2) = bbox
            do_flip = 0.0 if hand_cls == 'right' else 1.0
            center_x, center_y = (x1 + x2) / 2.0, (y1 + y2) / 2.0
            rescaling_factor = 2.5
            scale = np.array([rescaling_factor * (x2 - x1) / 200.0, rescaling_factor * (y2 - y1) / 200.0])
            BBOX_SHAPE = self.cfg.MODEL.get('BBOX_SHAPE', None)
            if BBOX_SHAPE is not None:
                final_bbox_size = float(expand_to_aspect_ratio(scale * 200, target_aspect_ratio=BBOX_SHAPE).max())
            else:
                final_bbox_size = float(max(x2 - x1, y2 - y1) * rescaling_factor)
            boxes.append((center_x, center_y, final_bbox_size, hand_cls != 'right'))
            centers.append([center_x, center_y]); sizes.append(final_bbox_size); flips.append(do_flip)
            transs.append(gen_trans_from_patch_cv(center_x, center_y, final_bbox_size, final_bbox_size, P, P, 1.0, 0))
        frame = torch.from_numpy(np.ascontiguousarray(img_0)).to(self.device)
        rec = ops.crop_boxes(boxes, P).to(self.device)
        img = ops.crop_batch(frame, rec, self.mean, self.std, P)
        n = len(bboxs)
        trans = torch.tensor(np.stack(transs), dtype=torch.float32)
        return {
            'img': img,                                                               # (B,3,P,P) on device
            'box_center': torch.tensor(centers, dtype=torch.float32),                 # (B,2)
            'box_size': torch.tensor(sizes, dtype=torch.float32),                     # (B,)
            'img_size': torch.tensor([[img_0.shape[1], img_0.shape[0]]] * n, dtype=torch.float32),
            'inv_trans': trans.clone(),                                               # == trans (datasets/utils.py:354-355)
            'trans': trans,
            'do_flip': torch.tensor(flips, dtype=torch.float32),
        }

    # ------------------------------------------------------------------ forward + camera maths
    @torch.no_grad()
    def estimate_from_rgb(self, img_0, detections, k_real=None, depth_refine=None):
        """infer.py:355-528 (d_infer.py:355 adds ``depth_refine``)."""
        if not isinstance(detections, list) or len(detections) == 0:
            raise ValueError("Invalid detections format")
        batch = self.prepare_batch_bbox(img_0, detections)
        for key in batch:
            if isinstance(batch[key], torch.Tensor):
                batch[key] = batch[key].to(self.device).float()
        out, params = self.model(batch)

        pred_cam = out['pred_cam']
        box_center, box_size, img_size = batch["box_center"], batch["box_size"], batch["img_size"]
        do_flip, trans, inv_trans = batch['do_flip'], batch['trans'], batch['inv_trans']

        pred_keypoints_3d = out['pred_keypoints_3d'].float()
        pred_keypoints_3d[:, :, 0] = pred_keypoints_3d[:, :, 0] * do_flip.unsqueeze(1)    # sic, infer.py:392
        flip_correction = 1.0 - 2.0 * do_flip.view(-1)
        pred_cam_corrected = pred_cam.clone()
        pred_cam_corrected[:, 1] = pred_cam_corrected[:, 1] * flip_correction

        if k_real is not None:
            if isinstance(k_real, np.ndarray):
                k_real = torch.from_numpy(k_real).float().to(self.device)
            elif isinstance(k_real, torch.Tensor):
                k_real = k_real.float().to(self.device)
            if k_real.dim() == 2:
                fx, fy, cx, cy = k_real[0, 0], k_real[1, 1], k_real[0, 2], k_real[1, 2]
            else:
                fx, fy, cx, cy = k_real[:, 0, 0], k_real[:, 1, 1], k_real[:, 0, 2], k_real[:, 1, 2]
            pred_cam_t_full = custom_cam_crop_to_full(pred_cam_corrected, box_center, box_size, img_size, fx, fy, cx, cy,
                                                      depth_refine=depth_refine)
            scaled_focal_length = fx.unsqueeze(0) if fx.dim() == 0 else fx
            kp_cam = pred_keypoints_3d + pred_cam_t_full.unsqueeze(1)
            depth = kp_cam[:, :, 2:3] + 1e-9
            x_norm, y_norm = kp_cam[:, :, 0:1] / depth, kp_cam[:, :, 1:2] / depth
            v = lambda t: t.view(-1, 1, 1) if torch.is_tensor(t) else t
            pred_keypoints_2d = torch.cat([x_norm * v(fx) + v(cx), y_norm * v(fy) + v(cy)], dim=-1)
        else:
            img_size_max = img_size.max(dim=1)[0] if img_size.dim() > 1 else img_size.max()
            scaled_focal_length = self.cfg.EXTRA.FOCAL_LENGTH / self.cfg.MODEL.IMAGE_SIZE * img_size_max
            pred_cam_t_full = custom_cam_crop_to_full(pred_cam_corrected, box_center, box_size, img_size,
                                                      scaled_focal_length, scaled_focal_length,
                                                      img_size[:, 0] / 2.0, img_size[:, 1] / 2.0, depth_refine=depth_refine)
            focal_length_2d = torch.stack([scaled_focal_length, scaled_focal_length], dim=1) \
                if scaled_focal_length.dim() == 1 else scaled_focal_length
            pred_keypoints_2d = perspective_projection(pred_keypoints_3d, translation=pred_cam_t_full,
                                                       focal_length=focal_length_2d)

        out['pred_keypoints_2d_full'] = pred_keypoints_2d
        out['pred_cam_t_full'] = pred_cam_t_full
        out['img'] = batch['img']
        out['focal_length'] = scaled_focal_length
        out['trans'] = trans
        out['do_flip'] = do_flip
        out['inv_trans'] = inv_trans
        return out, params


# ---------------------------------------------------------------------------------------- batch drivers
def _imread_bgr(path: str) -> Optional[np.ndarray]:
    """cv2.imread stand-in (infer.py:1252): HxWx3 uint8 BGR, None when unreadable."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])
    except Exception:
        return None


def hand_record(output: Dict, is_right: bool, index: int = 0) -> Dict:
    """The per-hand dict saved by process_batch_manopara (infer.py:1279-1303)."""
    mp = output['pred_mano_params']
    betas_np = mp['betas'][index].detach().cpu().numpy().squeeze()
    hand_pose_aa = matrix_to_axis_angle(mp['hand_pose'][index].detach().cpu().numpy())
    go = mp['global_orient'][index].detach().cpu().numpy().squeeze()
    if go.ndim == 3:
        go = go[0]
    global_orient_aa = rodrigues_log(go).flatten()
    cam_t_np = output['pred_cam_t_full'][index].detach().cpu().numpy().squeeze()
    return {'betas': betas_np, 'theta': np.concatenate((global_orient_aa, hand_pose_aa)), 'pose_hand': hand_pose_aa,
            'pose_global': global_orient_aa, 'cam_t': cam_t_np, 'is_right': is_right}


def process_batch_manopara(input_folder, output_folder, k_real=None, hamer=None, detector=None):
    """infer.py:1223-1318: per image, detect -> HaMeR -> save ``<stem>.npy`` holding
    ``{'left': None|hand, 'right': None|hand}``.  All hands of an image go through ONE forward."""
    os.makedirs(output_folder, exist_ok=True)
    if hamer is None:
        hamer = hamer_inference(hamer_opt)
    if detector is None:
        from .config.yolo_config import yolo_opt
        from .yolo.detector import Detector
        detector = Detector(yolo_opt)
    exts = ['*.jpg', '*.jpeg', '*.png', '*.bmp']
    image_paths = []
    for ext in exts:
        image_paths.extend(glob.glob(os.path.join(input_folder, ext)))
        image_paths.extend(glob.glob(os.path.join(input_folder, ext.upper())))
    image_paths = sorted(list(set(image_paths)))
    print(f"{len(image_paths)} images")
    for img_path in image_paths:
        file_name = os.path.splitext(os.path.basename(img_path))[0]
        image_results = {'left': None, 'right': None}
        try:
            image = _imread_bgr(img_path)
            if image is None:
                continue
            _, dets = detector.detect(image)
            detection_list = []
            if isinstance(dets, list) and len(dets) > 0:
                if isinstance(dets[0], list) and len(dets[0]) > 0 and isinstance(dets[0][0], list):
                    detection_list = dets[0]
                else:
                    detection_list = dets
            if not detection_list:
                continue
            try:
                output, _ = hamer.estimate_from_rgb(image, detection_list, k_real)
                for i, bbox in enumerate(detection_list):
                    image_results[bbox[0]] = hand_record(output, bbox[0] == 'right', i)
            except Exception as e:
                print(f"Error processing hand: {e}")
            np.save(os.path.join(output_folder, f"{file_name}.npy"), image_results)
        except Exception as e:
            print(f"Error processing file {img_path}: {e}")
            continue


def _list_images(input_folder):
    image_paths = []
    for ext in ['*.jpg', '*.jpeg', '*.png', '*.bmp']:
        image_paths.extend(glob.glob(os.path.join(input_folder, ext)))
        image_paths.extend(glob.glob(os.path.join(input_folder, ext.upper())))
    return sorted(list(set(image_paths)))


def _detection_list(dets):
    if isinstance(dets, list) and len(dets) > 0:
        if isinstance(dets[0], list) and len(dets[0]) > 0 and isinstance(dets[0][0], list):
            return dets[0]
        return dets
    return []


def process_batch(input_folder, output_folder, k_real=None, hamer=None, detector=None):
    """infer.py:908-1036: per detected hand one ``<stem>_<label>[_<n>].npz`` with the ROTATION-MATRIX form of the MANO
    parameters: ``betas (1,10)``, ``global_orient (1,1,3,3)``, ``hand_pose (1,15,3,3)``, ``cam_t (1,3)``, ``is_right``
    (= do_flip == 0).  A second hand with the same label gets the suffix ``_2``, ``_3`` ... (:996-998).  All hands of an
    image go through one forward; the saved arrays are the per-hand slices the reference's one-hand calls produce."""
    os.makedirs(output_folder, exist_ok=True)
    if hamer is None:
        hamer = hamer_inference(hamer_opt)
    if detector is None:
        from .config.yolo_config import yolo_opt
        from .yolo.detector import Detector
        detector = Detector(yolo_opt)
    for img_path in _list_images(input_folder):
        file_name = os.path.splitext(os.path.basename(img_path))[0]
        try:
            image = _imread_bgr(img_path)
            if image is None:
                continue
            _, dets = detector.detect(image)
            detection_list = _detection_list(dets)
            if not detection_list:
                continue
            output, _ = hamer.estimate_from_rgb(image, detection_list, k_real)
            mp = output['pred_mano_params']
            saved_counts = {'left': 0, 'right': 0}
            for i, bbox in enumerate(detection_list):
                hand_label = bbox[0]
                suffix = f"_{hand_label}"
                if saved_counts[hand_label] > 0:
                    suffix += f"_{saved_counts[hand_label] + 1}"
                np.savez(os.path.join(output_folder, f"{file_name}{suffix}.npz"),
                         betas=mp['betas'][i:i + 1].detach().cpu().numpy(),
                         global_orient=mp['global_orient'][i:i + 1].detach().cpu().numpy(),
                         hand_pose=mp['hand_pose'][i:i + 1].detach().cpu().numpy(),
                         cam_t=output['pred_cam_t_full'][i:i + 1].detach().cpu().numpy(),
                         is_right=bool(output['do_flip'][i] == 0))
                saved_counts[hand_label] += 1
        except Exception as e:
            print(f"Error processing file {img_path}: {e}")
            continue


def get_bbox_from_npy(npy_path, target_val=3):
    """infer.py:1040-1072: tight box [x1, y1, x2, y2] (floats) around the pixels of a label mask equal to
    ``target_val``; None when the file or the label is missing."""
    if not os.path.exists(npy_path):
        print(f"[Warning] mask not found: {npy_path}")
        return None
    mask = np.load(npy_path)
    rows, cols = np.where(mask == target_val)
    if len(rows) == 0:
        return None
    return [float(np.min(cols)), float(np.min(rows)), float(np.max(cols)), float(np.max(rows))]


def flip_axis_angle(rvec):
    """infer.py:1074-1080 (the reference's mirror is disabled: it returns the vector unchanged)."""
    return np.array([rvec[0], rvec[1], rvec[2]], dtype=rvec.dtype)


def process_batch_manopara_with_mask(input_folder, mask_folder, output_folder, intrinsics_path=None, hamer=None):
    """infer.py:1099-1220: no detector -- the hand box is the bounding box of label 3 in ``<mask_folder>/<stem>.npy``,
    always treated as a right hand; intrinsics from one txt file or from ``<intrinsics_path>/<stem>.txt`` per frame.
    Saves ``<stem>.npy`` with the same record as process_batch_manopara."""
    os.makedirs(output_folder, exist_ok=True)
    fixed_k, intrinsics_dir = None, None
    if intrinsics_path:
        if os.path.isfile(intrinsics_path):
            fixed_k = load_intrinsics(intrinsics_path)
        elif os.path.isdir(intrinsics_path):
            intrinsics_dir = intrinsics_path
        else:
            print(f"[Warning] invalid intrinsics path: {intrinsics_path}")
    if hamer is None:
        hamer = hamer_inference(hamer_opt)
    for img_path in _list_images(input_folder):
        file_name = os.path.splitext(os.path.basename(img_path))[0]
        image_results = {'left': None, 'right': None}
        bbox_coords = get_bbox_from_npy(os.path.join(mask_folder, f"{file_name}.npy"), target_val=3)
        if bbox_coords is None:
            continue
        current_k_real = None
        if fixed_k is not None:
            current_k_real = fixed_k
        elif intrinsics_dir is not None:
            txt_path = os.path.join(intrinsics_dir, f"{file_name}.txt")
            if os.path.exists(txt_path):
                current_k_real = load_intrinsics(txt_path)
        try:
            image = _imread_bgr(img_path)
            if image is None:
                continue
            try:
                output, _ = hamer.estimate_from_rgb(image, [['right', bbox_coords]], current_k_real)
                image_results['right'] = hand_record(output, True, 0)
            except Exception as e_inner:
                print(f"Error processing hand in {file_name}: {e_inner}")
            np.save(os.path.join(output_folder, f"{file_name}.npy"), image_results)
        except Exception as e:
            print(f"Error processing file {img_path}: {e}")
            continue


def write_obj(path: str, vertices: np.ndarray, faces: np.ndarray):
    with open(path, "w") as f:
        for v in vertices:
            f.write(f"v {v[0]:.8f} {v[1]:.8f} {v[2]:.8f}\n")
        for t in faces + 1:
            f.write(f"f {t[0]} {t[1]} {t[2]}\n")


def reconstruct_and_save_obj_with_wrapper(npy_folder, output_obj_folder, hamer_instance):
    """infer.py:1321-1436: .npy (axis-angle) -> rotation matrices -> MANO -> 778-vertex mesh; left hands are
    mirrored (x := -x, face winding flipped); += cam_t; one OBJ per image."""
    device = hamer_instance.device
    os.makedirs(output_obj_folder, exist_ok=True)
    mano = hamer_instance.mano
    mpd = {k: v.to(device) for k, v in mano.params.items() if v.dtype == torch.float32}
    for npy_path in sorted(glob.glob(os.path.join(npy_folder, '*.npy'))):
        file_name = os.path.splitext(os.path.basename(npy_path))[0]
        try:
            data = np.load(npy_path, allow_pickle=True).item()
            verts_all, faces_all, off = [], [], 0
            for hand_type in ['right', 'left']:
                hd = data[hand_type]
                if hd is None:
                    continue
                betas = torch.tensor(np.atleast_2d(hd['betas']), dtype=torch.float32, device=device)
                go = axis_angle_to_rotation_matrix_torch(torch.tensor(np.atleast_2d(hd['pose_global']), dtype=torch.float32, device=device))
                hp = axis_angle_to_rotation_matrix_torch(torch.tensor(hd['pose_hand'].reshape(-1, 3), dtype=torch.float32, device=device))
                R = torch.cat([go, hp], 0)                                   # (16,3,3)
                six = torch.cat([R[:, :, 0], R[:, :, 1]], dim=1).reshape(1, 96)   # rot6d of an exact rotation is itself
                o = ops.mano_forward(mpd, six, betas, torch.tensor([[1.0, 0.0, 0.0]], device=device))
                vertices = o["verts"][0].cpu().numpy()
                faces = mano.faces.astype(np.int32)
                if not hd['is_right']:
                    vertices[:, 0] = -vertices[:, 0]
                    faces = faces[:, [0, 2, 1]]
                vertices = vertices + hd['cam_t']
                verts_all.append(vertices); faces_all.append(faces + off); off += len(vertices)
            if verts_all:
                write_obj(os.path.join(output_obj_folder, f"{file_name}.obj"), np.concatenate(verts_all), np.concatenate(faces_all))
        except Exception as e:
            print(f"Error reconstructing {file_name}: {e}")
            continue


def load_intrinsics(txt_path):
    """infer.py:1458-1477."""
    if not os.path.exists(txt_path):
        raise FileNotFoundError(txt_path)
    try:
        k_real = np.loadtxt(txt_path, dtype=np.float32)
        if k_real.shape != (3, 3):
            raise ValueError(f"expected a 3x3 matrix, got {k_real.shape}")
        return k_real
    except Exception as e:
        print(f"cannot read intrinsics: {e}")
        return None


def main(argv=None):
    """CLI of infer.py:1479-1536: ``python -m hamer_yolo_amd.infer --input <RGB_dir> --output <out_dir>``."""
    ap = argparse.ArgumentParser(description="YOLOv7 -> HaMeR -> MANO parameters (.npy per image)")
    ap.add_argument('--input', type=str, required=True)
    ap.add_argument('--output', type=str, required=True)
    ap.add_argument('--intrinsics', type=str, default=None, help="3x3 camera matrix txt (the reference hard-codes its path)")
    ap.add_argument('--obj', type=str, default=None, help="also reconstruct OBJ meshes into this folder")
    args = ap.parse_args(argv)
    k_real = load_intrinsics(args.intrinsics) if args.intrinsics else None
    hamer = hamer_inference(hamer_opt)
    process_batch_manopara(args.input, args.output, k_real, hamer=hamer)
    if args.obj:
        reconstruct_and_save_obj_with_wrapper(args.output, args.obj, hamer)


if __name__ == '__main__':
    main()

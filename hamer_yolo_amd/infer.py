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
    def _box_scalars(self, bboxs: List):
        """Per-hand host scalars of infer.py:179-199: crop centre, square crop size S, flip flag, the 2x3 affine."""
        P = int(self.cfg.MODEL.IMAGE_SIZE)
        BBOX_SHAPE = self.cfg.MODEL.get('BBOX_SHAPE', None)
        boxes, centers, sizes, flips, transs = [], [], [], [], []
        for bbox in bboxs:
            if not (isinstance(bbox, list) and len(bbox) == 2 and isinstance(bbox[1], list) and len(bbox[1]) == 4):
                raise ValueError(f"Invalid bbox format: Expected [class, [x1,y1,x2,y2]], got {bbox}")
            hand_cls, (x1, y1, x2, y2) = bbox
            if not (x2 > x1 or y2 > y1):
                # a box clipped to nothing (scale_coords + round at the frame border) has crop size 0: the reference hands
                # cv2.getAffineTransform three coincident points and gets a meaningless patch; here it is an error
                raise ValueError(f"Invalid bbox format: empty box {bbox}")
            center_x, center_y = (x1 + x2) / 2.0, (y1 + y2) / 2.0
            rescaling_factor = 2.5
            scale = np.array([rescaling_factor * (x2 - x1) / 200.0, rescaling_factor * (y2 - y1) / 200.0])
            if BBOX_SHAPE is not None:
                final_bbox_size = float(expand_to_aspect_ratio(scale * 200, target_aspect_ratio=BBOX_SHAPE).max())
            else:
                final_bbox_size = float(max(x2 - x1, y2 - y1) * rescaling_factor)
            boxes.append((center_x, center_y, final_bbox_size, hand_cls != 'right'))
            centers.append([center_x, center_y]); sizes.append(final_bbox_size); flips.append(0.0 if hand_cls == 'right' else 1.0)
            transs.append(gen_trans_from_patch_cv(center_x, center_y, final_bbox_size, final_bbox_size, P, P, 1.0, 0))
        return boxes, centers, sizes, flips, transs

    def prepare_batch_frames(self, frames: List[torch.Tensor], dets_lists: List[List]) -> Dict[str, torch.Tensor]:
        """prepare_batch_bbox for the hands of SEVERAL frames at once: ``frames`` are (H,W,3) uint8 BGR device tensors,
        ``dets_lists[i]`` the boxes of frame i.  One hm_crop_batch launch per frame, all writing into one (sum B,3,P,P)
        batch tensor, so one HaMeR forward serves every hand of every frame.  ``frame_index`` says which frame a hand is from."""
        P = int(self.cfg.MODEL.IMAGE_SIZE)
        boxes, centers, sizes, flips, transs, img_sizes, fidx = [], [], [], [], [], [], []
        counts = []
        for i, (fr, bboxs) in enumerate(zip(frames, dets_lists)):
            b, c, sz, fl, tr = self._box_scalars(bboxs)
            boxes += b; centers += c; sizes += sz; flips += fl; transs += tr
            img_sizes += [[fr.shape[1], fr.shape[0]]] * len(bboxs)
            fidx += [i] * len(bboxs)
            counts.append(len(bboxs))
        n = len(boxes)
        if n == 0:
            raise ValueError("Invalid detections format")
        # one upload for all hands, from page-locked memory and asynchronous: a pageable host -> device copy makes the host wait
        # for everything already queued on its stream -- with several HaMeR batches queued per stream that was a whole forward
        # (tools/probes/e2e_trace.py: the third batch of a pass could not be enqueued before the first had finished)
        rec = self._up(ops.crop_boxes(boxes, P))
        rsz = rec.numel() // n
        img = torch.empty(n, 3, P, P, device=self.device, dtype=torch.float32)
        off = 0
        for fr, k in zip(frames, counts):
            if k:
                ops.crop_batch(fr, rec[off * rsz:(off + k) * rsz], self.mean, self.std, P, out=img[off:off + k])
                off += k
        trans = torch.tensor(np.stack(transs), dtype=torch.float32)
        return {
            'img': img,                                                               # (B,3,P,P) on device
            'box_center': torch.tensor(centers, dtype=torch.float32),                 # (B,2)
            'box_size': torch.tensor(sizes, dtype=torch.float32),                     # (B,)
            'img_size': torch.tensor(img_sizes, dtype=torch.float32),                 # (B,2) = (w, h) of the hand's frame
            'inv_trans': trans.clone(),                                               # == trans (datasets/utils.py:354-355)
            'trans': trans,
            'do_flip': torch.tensor(flips, dtype=torch.float32),
            'frame_index': torch.tensor(fidx, dtype=torch.long),
        }

    def _up(self, t: torch.Tensor) -> torch.Tensor:
        """Host tensor -> device without blocking the host: through page-locked memory, asynchronously on the current stream
        (the caching host allocator keeps the staging block alive until the copy has run)."""
        if self.device.type != "cuda" or t.is_cuda:
            return t.to(self.device)
        return t.pin_memory().to(self.device, non_blocking=True)

    def prepare_batch_bbox(self, img_0: np.ndarray, bboxs: List) -> Dict[str, torch.Tensor]:
        """infer.py:154-259.  img_0: HxWx3 uint8 BGR; bboxs: [[label, [x1, y1, x2, y2]], ...]."""
        frame = torch.from_numpy(np.ascontiguousarray(img_0)).to(self.device)
        batch = self.prepare_batch_frames([frame], [bboxs])
        del batch['frame_index']
        return batch

    # ------------------------------------------------------------------ forward + camera maths
    @torch.no_grad()
    def estimate_from_rgb(self, img_0, detections, k_real=None, depth_refine=None):
        """infer.py:355-528 (d_infer.py:355 adds ``depth_refine``)."""
        if not isinstance(detections, list) or len(detections) == 0:
            raise ValueError("Invalid detections format")
        return self._estimate(self.prepare_batch_bbox(img_0, detections), k_real, depth_refine)

    @torch.no_grad()
    def estimate_from_frames(self, frames: List[torch.Tensor], dets_lists: List[List], k_real=None, depth_refine=None):
        """estimate_from_rgb over the hands of several device-resident frames in ONE forward (the batched drivers below);
        ``out['frame_index'][h]`` is the frame of hand h, hands keep the order of ``dets_lists``."""
        batch = self.prepare_batch_frames(frames, dets_lists)
        fidx = batch.pop('frame_index')
        out, params = self._estimate(batch, k_real, depth_refine)
        out['frame_index'] = fidx
        return out, params

    def _estimate(self, batch, k_real=None, depth_refine=None):
        for key in batch:
            if isinstance(batch[key], torch.Tensor):
                batch[key] = self._up(batch[key]).float()
        # the intrinsics go up BEFORE the forward is queued: a pageable host -> device copy waits for everything already on the
        # stream, and behind `self.model(batch)` that is the whole HaMeR forward -- 19 ms per chunk during which the driver could
        # not start the next chunk (d_infer flow: 1960 -> hands/s see DESIGN.md); a numpy K is uploaded once and kept
        if isinstance(k_real, np.ndarray):
            kb = k_real.tobytes()
            if getattr(self, "_k_dev", (None, None))[0] != kb:
                self._k_dev = (kb, torch.from_numpy(np.ascontiguousarray(k_real)).float().to(self.device))
            k_real = self._k_dev[1]
        elif isinstance(k_real, torch.Tensor):
            k_real = k_real.float().to(self.device)
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
import threading as _threading
_scratch = _threading.local()          # per decoder thread: one reusable read buffer (see _read_bmp24)


def _read_bmp24(path: str, alloc=None, raw_rows: Optional[list] = None) -> Optional[np.ndarray]:
    """Uncompressed 24-bit BMP (what frame dumps usually are) straight into an HxWx3 BGR array: the file already holds BGR rows
    (bottom-up, padded to 4 bytes), so this is one strided copy instead of PIL's decode + RGB conversion + channel reversal
    (3 ms instead of 35 ms per 1080p frame on the build host).  None for anything else.  ``alloc(shape)`` may supply the
    destination array (the folder drivers hand out page-locked slots, so the strided copy IS the staging copy).
    ``raw_rows`` (a list, folder drivers): when the rows have no padding the pixel block is read with ONE readinto straight into the
    destination and a bottom-up file is left bottom-up -- the list then receives True, and the caller flips the rows where that
    is cheap (on the device, behind the upload: 16 frames are decoded in 2.0 ms instead of 4.7, tools/probes/decode_rate.py)."""
    import struct
    with open(path, "rb", buffering=0) as f:
        head = f.read(54)
        if len(head) < 54 or head[:2] != b"BM":
            return None
        off, = struct.unpack_from("<I", head, 10)
        hsize, w, h, planes, bpp, comp = struct.unpack_from("<IiiHHI", head, 14)
        if hsize < 40 or planes != 1 or bpp != 24 or comp != 0 or w <= 0 or h == 0:
            return None
        stride = (w * 3 + 3) & ~3
        dst = alloc((abs(h), w, 3)) if alloc is not None else None
        if dst is not None and raw_rows is not None and stride == w * 3:
            f.seek(off)
            view, got = memoryview(dst.reshape(-1)), 0
            while got < len(view):
                k = f.readinto(view[got:])
                if not k:
                    return None
                got += k
            raw_rows.append(h > 0)                       # True: the rows in `dst` run bottom-up
            return dst
        if dst is not None:
            # file -> this decoder thread's reusable scratch (readinto: no 6 MB allocation per frame, whose page faults serialise
            # the decoder threads on the process's memory map) -> one flipped copy into the destination slot.  Measured and
            # dropped: a scatter read straight into the slot (os.preadv, one iovec per row: 10x slower into page-locked memory)
            # and a read-only mapping of the file (mmap / munmap per frame: the same memory-map contention between threads).
            n = stride * abs(h)
            sc = getattr(_scratch, "buf", None)
            if sc is None or sc.size < n:
                sc = _scratch.buf = np.empty(n, dtype=np.uint8)
            f.seek(off)
            view, got = memoryview(sc)[:n], 0
            while got < n:
                k = f.readinto(view[got:])
                if not k:
                    return None
                got += k
            rows = sc[:n].reshape(abs(h), stride)[:, :w * 3].reshape(abs(h), w, 3)
            np.copyto(dst, rows[::-1] if h > 0 else rows)
            return dst
        f.seek(off)
        buf = f.read(stride * abs(h))
    if len(buf) < stride * abs(h):
        return None
    rows = np.frombuffer(buf, dtype=np.uint8).reshape(abs(h), stride)[:, :w * 3].reshape(abs(h), w, 3)
    if dst is None:
        return np.ascontiguousarray(rows[::-1] if h > 0 else rows)
    np.copyto(dst, rows[::-1] if h > 0 else rows)
    return dst


def _imread_bgr(path: str, alloc=None, raw_rows: Optional[list] = None) -> Optional[np.ndarray]:
    """cv2.imread stand-in (infer.py:1252): HxWx3 uint8 BGR, None when unreadable.  (``raw_rows``: see _read_bmp24.)"""
    try:
        if path.lower().endswith(".bmp"):
            im = _read_bmp24(path, alloc, raw_rows)
            if im is not None:
                return im
        from PIL import Image
        with Image.open(path) as im:
            return np.ascontiguousarray(np.asarray(im.convert("RGB"))[:, :, ::-1])
    except Exception:
        return None


def rodrigues_log_batch(R: np.ndarray) -> np.ndarray:
    """rodrigues_log over (N,3,3) -> (N,3) float32 (vectorised; the rare near-pi / near-identity rows take the scalar path)."""
    R = np.asarray(R, dtype=np.float64).reshape(-1, 3, 3)
    c = np.clip((np.trace(R, axis1=1, axis2=2) - 1.0) * 0.5, -1.0, 1.0)
    r = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], axis=1) * 0.5
    sn = np.linalg.norm(r, axis=1)
    theta = np.arctan2(sn, c)
    out = (r * (theta / np.maximum(sn, 1e-300))[:, None]).astype(np.float32)
    for i in np.nonzero(sn < 1e-5)[0]:
        out[i] = rodrigues_log(R[i])
    return out


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


def box_has_area(det) -> bool:
    """False for a detection whose box was clipped to nothing at the frame border (x2 <= x1 and y2 <= y1): it has no crop
    (size 0) and the folder drivers skip it -- the reference's per-hand try/except plays that role (infer.py:1306-1308)."""
    x1, y1, x2, y2 = det[1]
    return x2 > x1 or y2 > y1


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


# The reference walks the folder one image and one hand at a time (infer.py:1248-1316: imread -> detect -> one
# estimate_from_rgb per hand -> save).  Here the same per-image results come out of a pipeline that keeps the GPU full and
# hands each stage the batch shape it is tuned for (round 4: the two stages no longer share a chunk):
#   * images are decoded by a thread pool straight into page-locked slots, a rolling window ahead of the GPU;
#   * DETECTOR PASSES: runs of consecutive frames of one size -- `frames_per_step` for the first pass (so the first HaMeR
#     forward starts early), up to `det_frames` (48) afterwards (the 12x20 / 24x40 maps of YOLOv7 want many frames per launch: 394 TFLOP/s at 16 frames, 556 at 48, 591 at 64 -- but passes of 64 or 96 frames cost the 192-frame folder 1-4 % end to end, profiles/r04_e2e_det_frames_sweep.log) --
#     go up as one copy and through ONE batched YOLOv7 pass + NMS on the detector's own stream; one host sync per pass
#     returns all its box lists, and the hands join a queue in (file, detection) order;
#   * HaMeR BATCHES: whenever `hands_per_forward` hands are queued, exactly that many -- across frame and pass boundaries --
#     are cropped into one batch tensor (one crop launch per contributing frame) and go through ONE HaMeR forward and one
#     vectorised camera step, batches alternating between `in_flight` streams; the rest is flushed at the end of the folder;
#   * a file's result is emitted, in path order, once its last hand has come back.
# Multi-GPU (SURVEY 8e "frames round-robin per rank, YOLO run where the frame lives"): rank r of `world` takes files
# r, r + world, ...; every rank runs this whole pipeline on its own GPU and writes its own files -- no data-path collective.
FRAMES_PER_STEP = 16
DET_FRAMES = 48
HANDS_PER_FORWARD = 64


def _driver_streams(dev, n: int):
    """The folder drivers' HIP streams: the package's one shared set per device (streams.py), the same streams
    HamerEngine.contexts hands out -- a fresh pair per call lands on another pair of hardware queues each time, and some
    pairs share a queue (measured: the same pass 6-15 % slower)."""
    from .streams import get_streams
    return get_streams(dev, n)


_POOLS: Dict[int, object] = {}


def _decode_pool(nthreads: int):
    """The decoder threads, created once per thread count and kept for the life of the process: a decoder's read scratch
    (_read_bmp24) is thread-local, and a fresh pool per pass meant sixteen fresh 6 MB buffers whose first-touch page faults
    serialise on the process's memory map -- 8.4 ms until the first 16 frames of a pass were decoded, against 3-4 ms with warm
    threads (tools/probes/decode_rate.py, profiles/r04_decode_rate.log)."""
    from concurrent.futures import ThreadPoolExecutor
    pool = _POOLS.get(nthreads)
    if pool is None:
        pool = _POOLS[nthreads] = ThreadPoolExecutor(max_workers=nthreads, thread_name_prefix="hm-decode")
    return pool


class _Cpu:
    """Stand-ins for stream / event on a host-only run of the driver (the world_size-2 CPU test drives it with stub models)."""

    class Stream:
        def synchronize(self): pass
        def wait_event(self, ev): pass
        def wait_stream(self, st): pass

    class Event:
        def record(self, st=None): pass
        def synchronize(self): pass

    class ctx:
        def __init__(self, st): pass
        def __enter__(self): return self
        def __exit__(self, *a): return False


class _FrameRing:
    """Page-locked frame slots owned by ONE generator: file i decodes into slot i % n, and a detector pass goes up as
    asynchronous host -> device copies straight from the slots (no staging copy).  Every slot carries the event recorded
    behind the last copy that read it; a decoder waits for that event before it rewrites the slot, and files are only
    submitted for decoding once the slot's previous occupant has been uploaded -- so a slot is never rewritten under a
    pending copy, whatever `in_flight`, pass sizes or window the caller chose.  Nothing is cached at class
    level: two generators never share slots, and the memory goes back to torch's host allocator with the generator."""
    MAX_SHAPES = 2                       # a folder of many frame sizes: later sizes decode into ordinary arrays

    def __init__(self, n: int, pin: bool):
        import threading
        self.n, self.pin = n, pin
        self.bufs: Dict = {}
        self.events: List = [None] * n
        self.lock = threading.Lock()

    def slot(self, index: int, shape):
        """(tensor view, numpy view) of file `index`'s slot for frames of `shape`, safe to write; None when out of room."""
        s = index % self.n
        ev = self.events[s]
        if ev is not None:
            ev.synchronize()
        shape = tuple(shape)
        with self.lock:
            b = self.bufs.get(shape)
            if b is None:
                if len(self.bufs) >= self.MAX_SHAPES:
                    return None
                t = torch.empty((self.n,) + shape, dtype=torch.uint8, pin_memory=self.pin)
                b = self.bufs[shape] = (t, t.numpy())
        return b[0][s], b[1][s]


def shard_paths(paths: List[str], rank: int = 0, world: int = 1) -> List[str]:
    """Rank `rank`'s files of a folder job split over `world` ranks: round robin (r, r + world, ...), so that every rank's
    share spans the whole sequence and consecutive frames of one camera stay evenly spread."""
    if not 0 <= rank < world:
        raise ValueError(f"rank {rank} outside world {world}")
    return list(paths[rank::world])


def iter_folder_results(image_paths, hamer, detector, k_real=None, frames_per_step: int = FRAMES_PER_STEP, in_flight: int = 2,
                        decode_threads: Optional[int] = None, depth_model=None, hands_per_forward: int = HANDS_PER_FORWARD,
                        det_frames: Optional[int] = None, rank: int = 0, world: int = 1, stats: Optional[Dict] = None,
                        overlap_detector: bool = True, balance_tail: bool = True, det_ramp=None):
    """Yields ``(path, detection_list, hands)`` per image that has detections, in path order; ``hands`` is a dict of host
    numpy arrays for that image's hands in detection order: betas (n,10), global_orient (n,1,3,3), hand_pose (n,15,3,3),
    cam_t (n,3), do_flip (n,), plus the axis-angle forms pose_global (n,3) and pose_hand (n,45).
    ``depth_model`` (d_infer): a RootNet ``EstimateRGB``; every hand's root depth comes from ONE RootNet forward per HaMeR
    batch and enters the camera step as its ``depth_refine``; hands without a RootNet patch are dropped, as the reference's
    per-hand try/except drops them.
    ``rank`` / ``world``: this process handles ``shard_paths(image_paths, rank, world)`` only.
    ``stats`` (optional dict) receives ``images`` (files of this rank), ``frames`` (files with a result), ``hands`` (hands that
    went through HaMeR), ``det_passes`` and ``forwards`` (and their sizes, ``det_pass_sizes`` / ``forward_sizes``, in launch order).
    ``balance_tail=False``: the folder's last hands go out as forwards of ``hands_per_forward`` and one remainder (up to 1.25 x)
    instead of equal parts on all streams.
    ``overlap_detector=False`` puts the detector passes on the first HaMeR stream (with ``in_flight=1``: a strictly serial
    GPU timeline, what bench.py's per-launch profile pass wants)."""
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor
    image_paths = shard_paths(list(image_paths), rank, world)
    dev = torch.device(hamer.device)
    on_gpu = dev.type == "cuda"
    nthreads = decode_threads or max(1, min(16, len(os.sched_getaffinity(0))))
    H = max(1, int(hands_per_forward))
    first_pass = max(1, int(frames_per_step))
    big_pass = max(first_pass, int(det_frames) if det_frames else (DET_FRAMES if first_pass >= FRAMES_PER_STEP else first_pass))
    ramp = [max(1, int(x)) for x in det_ramp] if det_ramp else [first_pass]      # sizes of the first detector passes (tuning: e.g. (8, 8))
    n_streams = max(1, in_flight)
    if on_gpu:
        streams = _driver_streams(dev, n_streams + 1)         # [0 .. in_flight) HaMeR batches, the last one the detector's
        hstreams, dstream = streams[:n_streams], (streams[n_streams] if overlap_detector else streams[0])
        new_event, stream_ctx = (lambda: torch.cuda.Event()), torch.cuda.stream
    else:
        hstreams, dstream = [_Cpu.Stream() for _ in range(n_streams)], _Cpu.Stream()
        new_event, stream_ctx = _Cpu.Event, _Cpu.ctx
    st = stats if stats is not None else {}
    st.update(images=len(image_paths), frames=0, hands=0, det_passes=0, forwards=0, forward_sizes=[], det_pass_sizes=[])
    import time as _time
    _t0 = _time.perf_counter()
    trace = st.setdefault("trace", []) if os.environ.get("HAMER_E2E_TRACE") == "1" else None      # (host timeline of a pass, ms: tuning runs)

    def mark(what):
        if trace is not None:
            trace.append((round((_time.perf_counter() - _t0) * 1e3, 2), what))

    ahead = max(4 * first_pass, big_pass + first_pass)                       # decode-ahead window, in files
    ring = _FrameRing(min(max(1, len(image_paths)), ahead + 2 * big_pass), pin=on_gpu)
    files: List[Optional[Dict]] = [None] * len(image_paths)        # per file: dets, rows of finished hands, hands still out
    queue = deque()                                                   # (file index, detection index, frame tensor)
    det_jobs, batches = deque(), deque()
    hands_per_frame = 4.0                                             # running estimate, for how far ahead the detector runs

    def decode(index, path):
        """One file -> (HxWx3 BGR array or None, its page-locked slot as a tensor or None, rows bottom-up?)."""
        taken, raw = [], []

        def alloc(shape):
            got = ring.slot(index, shape)
            if got is None:
                return None
            taken.append(got[0])
            return got[1]
        im = _imread_bgr(path, alloc, raw)
        if trace is not None and index % 8 == 7:
            mark(f"decoded file {index}")
        if im is not None and not taken:                              # not decoded in place (PIL path): one copy into the slot
            got = ring.slot(index, im.shape)
            if got is not None:
                np.copyto(got[1], im)
                return got[1], got[0], False
        return im, (taken[0] if taken and im is not None else None), bool(raw and raw[0] and taken)

    # ------------------------------------------------------------------ detector passes
    def det_enqueue(first, items):
        """items: [(file index, array, slot tensor or None)] of one frame size -> upload + one batched pass on the detector's
        stream; returns the job to harvest."""
        shape = items[0][1].shape
        with stream_ctx(dstream):
            if not on_gpu:                                         # (host-only run: the "upload" is a copy out of the ring)
                frames = [torch.from_numpy(np.array(im[::-1] if rev else im, copy=True)) for _, im, _, rev in items]
            elif all(t is not None for _, _, t, _ in items):
                d = torch.empty((len(items),) + tuple(shape), dtype=torch.uint8, device=dev)
                s0 = items[0][0] % ring.n
                buf = ring.bufs[tuple(shape)][0]
                if all(fi % ring.n == s0 + j for j, (fi, _, _, _) in enumerate(items)):
                    d.copy_(buf[s0:s0 + len(items)], non_blocking=True)           # consecutive slots: ONE copy for the pass
                else:
                    for j, (_, _, t, _) in enumerate(items):
                        d[j].copy_(t, non_blocking=True)
                ev = new_event()
                ev.record(dstream)
                mark(f"det_upload_enqueued {len(items)}")
                for fi, _, _, _ in items:
                    ring.events[fi % ring.n] = ev
                # frames the decoders left bottom-up (raw BMP rows, one read per file) are turned on the device: one flip of the
                # pass's tensor when all of them are (the usual case), else frame by frame
                if all(rev for _, _, _, rev in items):
                    d = torch.flip(d, dims=(1,))
                elif any(rev for _, _, _, rev in items):
                    d = torch.stack([torch.flip(d[j], dims=(0,)) if items[j][3] else d[j] for j in range(len(items))])
                frames = [d[j] for j in range(len(items))]
            else:
                frames = [torch.from_numpy(np.ascontiguousarray(im[::-1] if rev else im)).to(dev) for _, im, _, rev in items]
            if hasattr(detector, "detect_frames_enqueue"):
                token = detector.detect_frames_enqueue(frames)
            else:
                token = None
            done = new_event()
            done.record(dstream)
        st["det_passes"] += 1
        st["det_pass_sizes"].append(len(items))
        mark(f"det_enqueued {len(items)}")
        return {"items": items, "frames": frames, "token": token, "done": done}

    def det_harvest(job):
        """The pass's one host sync -> box lists -> per-file state + the hand queue."""
        nonlocal hands_per_frame
        frames, items = job["frames"], job["items"]
        with stream_ctx(dstream):
            if job["token"] is not None:
                _, dets_lists = detector.detect_frames_finish(job["token"])
            elif hasattr(detector, "detect_frames"):
                _, dets_lists = detector.detect_frames(frames)
            else:                                                  # any object with the reference's detect(image) works too
                dets_lists = [detector.detect(np.ascontiguousarray(im[::-1]) if rev else im)[1] for _, im, _, rev in items]
        job["done"].synchronize()        # the pass's uploads and kernels are complete: its frames may be read from any stream
        mark(f"det_harvested {len(items)}")
        kept = []                                  # (all frames' lists first, then the commit: a box that raises here must not
        for fr, dl in zip(frames, dets_lists):     #  leave half a pass in the queue before the pass is redone file by file)
            dl = [d for d in _detection_list(dl) if box_has_area(d)]
            if depth_model is not None and dl:
                ok = depth_model.valid_boxes(dl, int(fr.shape[1]), int(fr.shape[0]))
                dl = [d for d, v in zip(dl, ok) if v]
            kept.append(dl)
        found = 0
        for (fi, _, _, _), fr, dl in zip(items, frames, kept):
            files[fi] = {"dets": dl, "rows": [None] * len(dl), "out": len(dl)}
            for j in range(len(dl)):
                queue.append((fi, j, fr))
            found += len(dl)
        hands_per_frame = 0.7 * hands_per_frame + 0.3 * (found / max(1, len(items)))

    def det_one_by_one(items):
        """A pass that raised: isolate the bad file (the reference's per-file try/except, infer.py:1314-1316)."""
        for it in items:
            try:
                job = det_enqueue(it[0], [it])
                det_harvest(job)
            except Exception as e1:
                print(f"Error processing file {image_paths[it[0]]}: {e1}")
                files[it[0]] = {"dets": [], "rows": [], "out": 0}

    # ------------------------------------------------------------------ HaMeR batches
    def batch_enqueue(hands, stream):
        """`hands`: [(file, detection, frame)] in queue order -> crops, (RootNet,) HaMeR forward and camera step on `stream`.
        (Their frames are on the device: a hand enters the queue only behind its pass's host sync, see det_harvest.)"""
        frames, dets_lists = [], []
        for fi, j, fr in hands:
            if not frames or frames[-1] is not fr:
                frames.append(fr); dets_lists.append([])
            dets_lists[-1].append(files[fi]["dets"][j])
        with stream_ctx(stream):
            depth = depth_model.estimate_root_depths_frames(frames, k_real, dets_lists) if depth_model is not None else None
            out, _ = hamer.estimate_from_frames(frames, dets_lists, k_real, depth_refine=depth)
            mp = out['pred_mano_params']
            dev_res = {'betas': mp['betas'], 'global_orient': mp['global_orient'], 'hand_pose': mp['hand_pose'],
                       'cam_t': out['pred_cam_t_full'], 'do_flip': out['do_flip']}
            done = new_event()
            done.record(stream)          # (its own event: a stream synchronise would also wait for the NEXT batch queued on that stream)
        st["forwards"] += 1
        st["forward_sizes"].append(len(hands))
        mark(f"batch_enqueued {len(hands)}")
        return {"done": done, "hands": [(fi, j) for fi, j, _ in hands], "dev": dev_res, "frames": frames}   # (frames: alive until finish)

    def batch_finish(job):
        job["done"].synchronize()
        mark(f"batch_done {len(job['hands'])}")
        res = {k: v.detach().cpu().numpy() for k, v in job["dev"].items()}
        n = res['betas'].shape[0]
        res['pose_global'] = rodrigues_log_batch(res['global_orient'].reshape(n, 3, 3))
        res['pose_hand'] = rodrigues_log_batch(res['hand_pose'].reshape(n * 15, 3, 3)).reshape(n, 45)
        for r, (fi, j) in enumerate(job["hands"]):
            files[fi]["rows"][j] = {k: v[r] for k, v in res.items()}
            files[fi]["out"] -= 1
        st["hands"] += n

    def batch_failed(hands, stream, err):
        """A batch that raised: redo it frame by frame so that one bad box costs its own frame's hands only."""
        print(f"Error processing a batch of {len(hands)} hands starting at {image_paths[hands[0][0]]}: {err}")
        groups = []
        for h in hands:
            if not groups or groups[-1][-1][0] != h[0]:
                groups.append([])
            groups[-1].append(h)
        for g in groups:
            try:
                batch_finish(batch_enqueue(g, stream))
            except Exception as e1:
                print(f"Error processing file {image_paths[g[0][0]]}: {e1}")
                for fi, j, _ in g:
                    files[fi]["out"] -= 1                          # (its row stays None: dropped at emission)

    emit_at = 0

    def emit_ready():
        """Files in path order whose hands have all come back."""
        nonlocal emit_at
        while emit_at < len(files) and files[emit_at] is not None and files[emit_at]["out"] == 0:
            f = files[emit_at]
            keep = [j for j, r in enumerate(f["rows"]) if r is not None]
            if keep:
                hands = {k: np.stack([f["rows"][j][k] for j in keep]) for k in f["rows"][keep[0]]}
                st["frames"] += 1
                yield image_paths[emit_at], [f["dets"][j] for j in keep], hands
            files[emit_at] = {"dets": [], "rows": [], "out": 0}        # (drop the arrays)
            emit_at += 1

    pool = _decode_pool(nthreads)
    # the interpreter hands the GIL over every 5 ms by default: with sixteen decoders and an enqueueing thread that is the time
    # a decoder may wait before it even starts its (GIL-free) read -- 0.5 ms for the duration of the pass: the first detector
    # pass goes out ~1 ms earlier (tools/gpu/r04_aa.sh)
    import sys as _sys
    _switch = _sys.getswitchinterval()
    _sys.setswitchinterval(min(_switch, 0.0005))
    try:
        futs, submitted, uploaded, taken = deque(), 0, 0, 0
        carry = None                                                  # a decoded frame that did not fit the previous pass (other size)

        def top_up():
            # a rolling window: `ahead` files beyond the ones already taken, and never into a slot whose occupant has not gone up
            nonlocal submitted
            while submitted < min(len(image_paths), taken + ahead, uploaded + ring.n):
                futs.append(pool.submit(decode, submitted, image_paths[submitted]))
                submitted += 1

        def next_items(limit):
            """Up to `limit` consecutive decoded frames of one size (unreadable files are skipped with an empty result)."""
            nonlocal carry, taken
            items, shape = [], None
            while len(items) < limit:
                if carry is not None:
                    it, carry = carry, None
                elif taken < len(image_paths):
                    top_up()
                    im, slot, rev = futs.popleft().result()
                    it = (taken, im, slot, rev)
                    taken += 1
                    if im is None:
                        files[it[0]] = {"dets": [], "rows": [], "out": 0}
                        continue
                else:
                    break
                if items and it[1].shape != shape:
                    carry = it
                    break
                shape = it[1].shape
                items.append(it)
            return items

        passes = 0
        tail_plan = None                                              # sizes of the last forwards, fixed once the folder's last hands are queued
        while True:
            frames_left = carry is not None or taken < len(image_paths)
            # 1. keep the detector ahead of HaMeR: a pass in flight whenever fewer than ~3 batches of hands are queued or expected
            expected = len(queue) + sum(len(j["items"]) for j in det_jobs) * hands_per_frame
            if frames_left and len(det_jobs) < 2 and expected < 3 * H:
                items = next_items(ramp[passes] if passes < len(ramp) else big_pass)
                if items:
                    mark(f"det_items_ready {len(items)}")
                    passes += 1
                    try:
                        det_jobs.append(det_enqueue(items[0][0], items))
                    except Exception as e:
                        print(f"Error processing a detector pass starting at {image_paths[items[0][0]]}: {e}")
                        det_one_by_one(items)
                    uploaded = items[-1][0] + 1
                    top_up()
                    continue
            # 2. full batches as long as the queue holds them; at the end of the folder the remainder (a last batch of up to
            #    H + H/4 hands rather than a full one and a sliver)
            draining = not frames_left and not det_jobs
            if draining and queue and tail_plan is None:
                # the end of the folder: what is queued now is all there will be.  Forwards of H hands until the rest fits two
                # forwards of at most 1.5 H, and that rest in EQUAL parts, one per stream -- the streams then finish together (a
                # last forward running alone costs 8 % more per hand than two in flight, and a full one followed by a sliver
                # twice that; forwards of 40 .. 96 hands cost within 6 % of the tuned 64 per hand)
                rest, tail_plan = len(queue), []
                while balance_tail and rest > n_streams * (H + H // 2):
                    tail_plan.append(H); rest -= H
                if balance_tail and n_streams > 1 and rest > H // 2:
                    parts = n_streams if rest <= n_streams * (H + H // 2) else 2 * n_streams
                    tail_plan += [rest // parts + (1 if i < rest % parts else 0) for i in range(parts)]
                else:
                    while rest > H + H // 4:
                        tail_plan.append(H); rest -= H
                    tail_plan.append(rest)
            while len(batches) < 2 * n_streams and (len(queue) >= H or (draining and queue)):
                take = tail_plan.pop(0) if (draining and tail_plan) else H
                take = min(take, len(queue))
                hands = [queue.popleft() for _ in range(take)]
                stream = hstreams[st["forwards"] % n_streams]
                try:
                    batches.append(batch_enqueue(hands, stream))
                except Exception as e:
                    batch_failed(hands, stream, e)
            # 3. wait where waiting costs least: for the detector when HaMeR has nothing to chew on, else for the oldest batch
            if det_jobs and (len(queue) < H or not batches):
                job = det_jobs.popleft()
                try:
                    det_harvest(job)
                except Exception as e:
                    print(f"Error processing a detector pass starting at {image_paths[job['items'][0][0]]}: {e}")
                    det_one_by_one(job["items"])
            elif batches:
                job = batches.popleft()
                try:
                    batch_finish(job)
                except Exception as e:
                    hs = [(fi, j, None) for fi, j in job["hands"]]
                    print(f"Error finishing a batch starting at {image_paths[hs[0][0]]}: {e}")
                    for fi, j, _ in hs:
                        files[fi]["out"] -= 1
                yield from emit_ready()
            elif not frames_left and not queue:
                break
        yield from emit_ready()
        mark("end")
    finally:
        _sys.setswitchinterval(_switch)


def _record_from(hands: Dict, i: int, is_right: bool) -> Dict:
    """hand i of an image as the dict of infer.py:1296-1303."""
    pg, ph = hands['pose_global'][i], hands['pose_hand'][i]
    return {'betas': hands['betas'][i].squeeze(), 'theta': np.concatenate((pg, ph)), 'pose_hand': ph, 'pose_global': pg,
            'cam_t': hands['cam_t'][i].squeeze(), 'is_right': is_right}


def _default_models(hamer, detector):
    if hamer is None:
        hamer = hamer_inference(hamer_opt)
    if detector is None:
        from .config.yolo_config import yolo_opt
        from .yolo.detector import Detector
        detector = Detector(yolo_opt)
    return hamer, detector


def _rank_world(rank, world):
    """(rank, world) of a folder job: explicit arguments win; otherwise the process group when one is initialised (every rank
    of a torchrun launch then takes its own share of the folder); otherwise the whole folder."""
    if rank is not None or world is not None:
        return int(rank or 0), int(world or 1)
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def gather_job_stats(stats: Dict) -> Dict:
    """Every rank's (images, frames, hands) of a sharded folder job -> ``stats['per_rank']`` (a list in rank order) and the
    ``global_*`` sums, on every rank: ONE all_gather of three integers, after the data path has finished (the job itself has
    no collective).  A single process gets its own numbers back."""
    import torch.distributed as dist
    from . import shard
    mine = [int(stats.get(k, 0)) for k in ("images", "frames", "hands")]
    if dist.is_available() and shard._multi():
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.tensor(mine, dtype=torch.int64, device=dev)
        full = torch.empty(dist.get_world_size() * 3, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(full, t)
        rows = full.cpu().reshape(-1, 3).tolist()
    else:
        rows = [mine]
    stats["per_rank"] = [{"images": a, "frames": b, "hands": c} for a, b, c in rows]
    for i, k in enumerate(("images", "frames", "hands")):
        stats["global_" + k] = sum(r[i] for r in rows)
    return stats


def process_batch_manopara(input_folder, output_folder, k_real=None, hamer=None, detector=None,
                           frames_per_step: int = FRAMES_PER_STEP, rank: Optional[int] = None, world: Optional[int] = None, **pipeline):
    """infer.py:1223-1318: per image, detect -> HaMeR -> save ``<stem>.npy`` holding ``{'left': None|hand, 'right':
    None|hand}`` (the last detection of a label wins, as in the reference's loop); images without detections write nothing.
    Runs on the pipeline above: same files, same numbers, the GPU kept busy.  Under ``torch.distributed`` (or with explicit
    ``rank`` / ``world``) every rank processes ``shard_paths(files, rank, world)`` on its own GPU and writes its own files into
    the same folder; the union is what one process writes.  Returns this rank's ``{'images', 'frames', 'hands', ...}`` counts
    (the reference returns None)."""
    os.makedirs(output_folder, exist_ok=True)
    hamer, detector = _default_models(hamer, detector)
    rank, world = _rank_world(rank, world)
    image_paths = _list_images(input_folder)
    print(f"{len(image_paths)} images" + (f" (rank {rank} of {world}: {len(shard_paths(image_paths, rank, world))})" if world > 1 else ""))
    stats: Dict = {}
    for img_path, detection_list, hands in iter_folder_results(image_paths, hamer, detector, k_real, frames_per_step,
                                                               rank=rank, world=world, stats=stats, **pipeline):
        file_name = os.path.splitext(os.path.basename(img_path))[0]
        image_results = {'left': None, 'right': None}
        for i, bbox in enumerate(detection_list):
            image_results[bbox[0]] = _record_from(hands, i, bbox[0] == 'right')
        np.save(os.path.join(output_folder, f"{file_name}.npy"), image_results)
    return stats


def process_batch(input_folder, output_folder, k_real=None, hamer=None, detector=None, frames_per_step: int = FRAMES_PER_STEP,
                  rank: Optional[int] = None, world: Optional[int] = None, **pipeline):
    """infer.py:908-1036: per detected hand one ``<stem>_<label>[_<n>].npz`` with the ROTATION-MATRIX form of the MANO
    parameters: ``betas (1,10)``, ``global_orient (1,1,3,3)``, ``hand_pose (1,15,3,3)``, ``cam_t (1,3)``, ``is_right``
    (= do_flip == 0).  A second hand with the same label gets the suffix ``_2``, ``_3`` ... (:996-998).  Same pipeline and
    rank sharding as process_batch_manopara; the saved arrays are the per-hand slices the reference's one-hand calls produce."""
    os.makedirs(output_folder, exist_ok=True)
    hamer, detector = _default_models(hamer, detector)
    rank, world = _rank_world(rank, world)
    stats: Dict = {}
    for img_path, detection_list, hands in iter_folder_results(_list_images(input_folder), hamer, detector, k_real, frames_per_step,
                                                               rank=rank, world=world, stats=stats, **pipeline):
        file_name = os.path.splitext(os.path.basename(img_path))[0]
        saved_counts = {'left': 0, 'right': 0}
        for i, bbox in enumerate(detection_list):
            hand_label = bbox[0]
            suffix = f"_{hand_label}"
            if saved_counts[hand_label] > 0:
                suffix += f"_{saved_counts[hand_label] + 1}"
            np.savez(os.path.join(output_folder, f"{file_name}{suffix}.npz"),
                     betas=hands['betas'][i:i + 1], global_orient=hands['global_orient'][i:i + 1],
                     hand_pose=hands['hand_pose'][i:i + 1], cam_t=hands['cam_t'][i:i + 1],
                     is_right=bool(hands['do_flip'][i] == 0))
            saved_counts[hand_label] += 1
    return stats


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
    ap.add_argument('--ckpt', type=str, default=None, help="hamer.ckpt path or synthetic:<seed> (default: config/hamer_config.py)")
    ap.add_argument('--yolo-weights', type=str, default=None, help="yolov7 .pt path or synthetic:<seed> (default: config/yolo_config.py)")
    args = ap.parse_args(argv)
    k_real = load_intrinsics(args.intrinsics) if args.intrinsics else None
    if args.ckpt:
        hamer_opt.ckpt_path = args.ckpt
    if args.yolo_weights:
        from .config.yolo_config import yolo_opt
        yolo_opt.weights = args.yolo_weights
    # under `python -m torch.distributed.run --nproc-per-node N -m hamer_yolo_amd.infer ...` every rank takes its share of
    # the folder on its own GPU (RANK / LOCAL_RANK / WORLD_SIZE from the environment); a plain launch is one process
    from . import shard
    rank, local, world = shard.init_distributed()
    if world > 1 and torch.cuda.is_available():
        torch.cuda.set_device(local)
    hamer = hamer_inference(hamer_opt)
    stats = process_batch_manopara(args.input, args.output, k_real, hamer=hamer, rank=rank, world=world)
    if world > 1:
        gather_job_stats(stats)
        if rank == 0:
            print(f"{stats['global_frames']} files, {stats['global_hands']} hands over {world} ranks: {stats['per_rank']}")
        import torch.distributed as dist
        dist.barrier()
    if args.obj and rank == 0:
        reconstruct_and_save_obj_with_wrapper(args.output, args.obj, hamer)


if __name__ == '__main__':
    main()

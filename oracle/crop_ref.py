"""ORACLE (test infrastructure): CPU restatement of the crop path
``hamer_inference.prepare_batch_bbox`` (hamer/infer.py:154-259) and its helpers in
hamer/hamer/datasets/utils.py (``expand_to_aspect_ratio`` :15-34, ``gen_trans_from_patch_cv``
:82-129, ``generate_image_patch_cv2`` :318-376, ``convert_cvimg_to_tensor`` :379-392) in numpy.

PARITY UNPINNED against OpenCV: the reference calls ``cv2.getAffineTransform`` /
``cv2.warpAffine`` / ``cv2.flip`` (opencv-python>=4.1.1, unpinned, absent from this image and
from /root/reference).  ``warp_affine_u8`` restates the published classic algorithm of
``cv::warpAffine`` for 8-bit INTER_LINEAR + BORDER_CONSTANT: the 2x3 matrix is inverted in
double, source coordinates are evaluated in 10-bit fixed point (AB_BITS) with
``round_delta = 16``, quantised to 1/32 pixel (INTER_BITS = 5) and blended with the integer
weights (32-fx)(32-fy)*32 (sum 2^15) and a rounding shift.  The pure-Python helpers
(`expand_to_aspect_ratio`, crop-size rule) are pinned by the closed-form known answers of
SURVEY.md section 8a (tests/test_crop_oracle.py).
"""
from __future__ import annotations

import numpy as np

AB_BITS, INTER_BITS, INTER_TAB = 10, 5, 32


def expand_to_aspect_ratio(input_shape, target_aspect_ratio=None):
    """datasets/utils.py:15-34."""
    if target_aspect_ratio is None:
        return input_shape
    w, h = input_shape
    w_t, h_t = target_aspect_ratio
    if h / w < h_t / w_t:
        h_new, w_new = max(w * h_t / w_t, h), w
    else:
        h_new, w_new = h, max(h * w_t / h_t, w)
    return np.array([w_new, h_new])


def bbox_to_center_size(x1, y1, x2, y2, bbox_shape=(192, 256), rescaling_factor=2.5):
    """infer.py:179-199: centre and the square crop side S."""
    cx, cy = (x1 + x2) / 2.0, (y1 + y2) / 2.0
    scale = np.array([rescaling_factor * (x2 - x1) / 200.0, rescaling_factor * (y2 - y1) / 200.0])
    size = expand_to_aspect_ratio(scale * 200, target_aspect_ratio=list(bbox_shape)).max()
    return cx, cy, float(size)


def gen_trans_from_patch(c_x, c_y, src_w, src_h, dst_w, dst_h):
    """datasets/utils.py:82-129 with scale 1, rot 0; cv2.getAffineTransform restated as the exact
    solve of the 3-point system (float32 control points, double arithmetic)."""
    src_center = np.array([c_x, c_y], dtype=np.float64)
    src_down = np.array([0.0, np.float32(src_h * 0.5)], dtype=np.float32)    # rotate_2d(...) with rot 0
    src_right = np.array([np.float32(src_w * 0.5), 0.0], dtype=np.float32)
    src = np.zeros((3, 2), dtype=np.float32)
    src[0] = src_center
    src[1] = src_center + src_down
    src[2] = src_center + src_right
    dst = np.array([[dst_w * 0.5, dst_h * 0.5], [dst_w * 0.5, dst_h], [dst_w, dst_h * 0.5]], dtype=np.float32)
    A = np.concatenate([src.astype(np.float64), np.ones((3, 1))], axis=1)
    return np.linalg.solve(A, dst.astype(np.float64)).T               # (2,3): dst = M . [src; 1]


def warp_affine_u8(img: np.ndarray, M: np.ndarray, out_w: int, out_h: int) -> np.ndarray:
    """cv::warpAffine, 8-bit, INTER_LINEAR, BORDER_CONSTANT(0), restated (see module docstring)."""
    M = np.array(M, dtype=np.float64)
    D = M[0, 0] * M[1, 1] - M[0, 1] * M[1, 0]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[1, 1] * D, M[0, 0] * D
    iM = np.array([[A11, -M[0, 1] * D, 0.0], [-M[1, 0] * D, A22, 0.0]])
    iM[0, 2] = -iM[0, 0] * M[0, 2] - iM[0, 1] * M[1, 2]
    iM[1, 2] = -iM[1, 0] * M[0, 2] - iM[1, 1] * M[1, 2]
    H, W = img.shape[:2]
    xs = np.arange(out_w, dtype=np.float64)
    ys = np.arange(out_h, dtype=np.float64)
    AB = float(1 << AB_BITS)
    adelta = np.rint(iM[0, 0] * xs * AB).astype(np.int64)
    bdelta = np.rint(iM[1, 0] * xs * AB).astype(np.int64)
    rd = (1 << AB_BITS) // INTER_TAB // 2
    X0 = np.rint((iM[0, 1] * ys + iM[0, 2]) * AB).astype(np.int64) + rd
    Y0 = np.rint((iM[1, 1] * ys + iM[1, 2]) * AB).astype(np.int64) + rd
    X = (X0[:, None] + adelta[None, :]) >> (AB_BITS - INTER_BITS)
    Y = (Y0[:, None] + bdelta[None, :]) >> (AB_BITS - INTER_BITS)
    sx, sy = X >> INTER_BITS, Y >> INTER_BITS
    fx, fy = X & (INTER_TAB - 1), Y & (INTER_TAB - 1)
    src = img.astype(np.int64)

    def tap(yy, xx):
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        v = src[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
        return v * ok[..., None]

    w00 = ((INTER_TAB - fx) * (INTER_TAB - fy))[..., None]
    w01 = (fx * (INTER_TAB - fy))[..., None]
    w10 = ((INTER_TAB - fx) * fy)[..., None]
    w11 = (fx * fy)[..., None]
    acc = w00 * tap(sy, sx) + w01 * tap(sy, sx + 1) + w10 * tap(sy + 1, sx) + w11 * tap(sy + 1, sx + 1)
    return ((acc + 512) >> 10).astype(np.uint8)


def prepare_batch_bbox(img_bgr: np.ndarray, bboxs, mean, std, bbox_shape=(192, 256), image_size=256):
    """infer.py:154-259.  bboxs: list of [label, [x1, y1, x2, y2]]; mean/std in 0..255 units.
    Returns the dict of numpy arrays the reference stacks at :250-258."""
    imgs, centers, sizes, img_sizes, transs, flips = [], [], [], [], [], []
    for label, (x1, y1, x2, y2) in bboxs:
        do_flip = 0.0 if label == "right" else 1.0
        cx, cy, S = bbox_to_center_size(x1, y1, x2, y2, bbox_shape)
        trans = gen_trans_from_patch(cx, cy, S, S, image_size, image_size)
        patch = warp_affine_u8(img_bgr, trans, image_size, image_size)
        patch = patch[:, :, ::-1]                     # BGR -> RGB (infer.py:228)
        if label != "right":
            patch = patch[:, ::-1]                    # cv2.flip(patch, 1) (infer.py:229-230)
        t = np.transpose(patch, (2, 0, 1)).astype(np.float32)
        for c in range(3):
            t[c] = (t[c] - np.float32(mean[c])) / np.float32(std[c])
        imgs.append(t); centers.append([cx, cy]); sizes.append(S)
        img_sizes.append([img_bgr.shape[1], img_bgr.shape[0]]); transs.append(trans); flips.append(do_flip)
    return {
        "img": np.stack(imgs).astype(np.float32), "box_center": np.array(centers, np.float32),
        "box_size": np.array(sizes, np.float32), "img_size": np.array(img_sizes, np.float32),
        "trans": np.stack(transs).astype(np.float32), "inv_trans": np.stack(transs).astype(np.float32),
        "do_flip": np.array(flips, np.float32),
    }

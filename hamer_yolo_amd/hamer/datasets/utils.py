"""Host-side crop helpers (reference: hamer/hamer/datasets/utils.py).  The pixel work
(cv2.warpAffine, flip, normalisation) runs in the HIP kernel hm_crop_batch; what stays on the
host is the per-box scalar arithmetic."""
import numpy as np


def expand_to_aspect_ratio(input_shape, target_aspect_ratio=None):
    """Smallest (w, h) >= input with w : h = target (reference behaviour: datasets/utils.py:15-34): each side is raised to
    what the other side demands at the target ratio -- exactly one of the two grows.  Anything that is not a (w, h) pair is
    returned unchanged, as is the input when no target is given."""
    if target_aspect_ratio is None:
        return input_shape
    wh = np.asarray(input_shape, dtype=np.float64)
    if wh.shape != (2,):
        return input_shape
    tw, th = target_aspect_ratio
    return np.maximum(wh, np.array([wh[1] * tw / th, wh[0] * th / tw]))


def gen_trans_from_patch_cv(c_x, c_y, src_width, src_height, dst_width, dst_height, scale=1.0, rot=0.0):
    """datasets/utils.py:82-129 for rot == 0: the 2x3 matrix cv2.getAffineTransform returns for the
    three float32 control points (centre, centre + down, centre + right), solved in closed form."""
    if rot != 0:
        raise NotImplementedError("the inference path always crops with rot = 0 (infer.py:221)")
    src_w, src_h = src_width * scale, src_height * scale
    p0x, p0y = np.float32(c_x), np.float32(c_y)
    p1y = np.float32(c_y + np.float32(src_h * 0.5))
    p2x = np.float32(c_x + np.float32(src_w * 0.5))
    ax = (dst_width * 0.5) / (float(p2x) - float(p0x))
    by = (dst_height * 0.5) / (float(p1y) - float(p0y))
    return np.array([[ax, 0.0, dst_width * 0.5 - ax * float(p0x)],
                     [0.0, by, dst_height * 0.5 - by * float(p0y)]], dtype=np.float64)


def convert_cvimg_to_tensor(cvimg: np.ndarray):
    """datasets/utils.py:379-392."""
    return np.transpose(cvimg.copy(), (2, 0, 1)).astype(np.float32)

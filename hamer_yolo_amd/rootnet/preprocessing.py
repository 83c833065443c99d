"""Box handling of the RootNet patch (reference behaviour: rootnet/preprocessing.py:152-188 ``sanitize_bbox`` /
``process_bbox``), written for a whole batch of boxes at once: the driver turns all hands of a frame into patch boxes in
one call.  The patch itself is the affine crop HaMeR uses (``src = c + (dst - 128) * S / 256``), done by hm_crop_batch."""
import numpy as np


def clip_boxes(boxes, img_width, img_height):
    """(N, 4) [x, y, w, h] -> (clipped (N, 4) float64, valid (N,) bool).  A box keeps its top-left corner inside the image
    (never below 0), its far corner is pulled back to the last pixel, and its extent is measured between those two pixels
    (so a 100-wide box becomes 99 wide); it is valid while it has area and the clipped extent is positive on both axes."""
    b = np.asarray(boxes, dtype=np.float64).reshape(-1, 4)
    lo = np.maximum(b[:, :2], 0.0)
    far = lo + np.maximum(b[:, 2:] - 1.0, 0.0)
    hi = np.minimum(far, np.array([img_width - 1.0, img_height - 1.0]))
    ext = hi - lo
    valid = (b[:, 2] * b[:, 3] > 0) & (ext > 0).all(axis=1)
    return np.concatenate([lo, ext], axis=1), valid


def patch_boxes(boxes, img_width, img_height, input_img_shape, ratio=1.25):
    """(N, 4) detector boxes -> (patch boxes (N, 4) float32, valid (N,)): clip, keep the centre, grow the short side to the
    aspect ratio of the network input (width / height = shape[1] / shape[0]) and scale both sides by ``ratio``."""
    clipped, valid = clip_boxes(boxes, img_width, img_height)
    centre = clipped[:, :2] + clipped[:, 2:] / 2.0
    aspect = input_img_shape[1] / input_img_shape[0]
    w, h = clipped[:, 2], clipped[:, 3]
    side = np.stack([np.maximum(w, h * aspect), np.maximum(h, w / aspect)], axis=1) * ratio
    return np.concatenate([centre - side / 2.0, side], axis=1).astype(np.float32), valid


def sanitize_bbox(bbox, img_width, img_height):
    """One box; None when nothing is left of it (the reference's calling convention)."""
    clipped, valid = clip_boxes(bbox, img_width, img_height)
    return clipped[0] if valid[0] else None


def process_bbox(bbox, img_width, img_height, input_img_shape, ratio=1.25):
    """One box -> float32 [x, y, w, h] of the patch, or None."""
    out, valid = patch_boxes(bbox, img_width, img_height, input_img_shape, ratio)
    return out[0] if valid[0] else None

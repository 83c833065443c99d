"""Box handling of the RootNet patch (reference: rootnet/preprocessing.py:152-188).  The patch itself is the affine crop
HaMeR uses (gen_trans_from_patch_cv + cv2.warpAffine, :39-150): ``src = c + (dst - 128) * S / 256``, done by hm_crop_batch."""
import numpy as np


def sanitize_bbox(bbox, img_width, img_height):
    """preprocessing.py:152-163: clip [x, y, w, h] to the image; None when nothing is left."""
    x, y, w, h = bbox
    x1 = np.max((0, x))
    y1 = np.max((0, y))
    x2 = np.min((img_width - 1, x1 + np.max((0, w - 1))))
    y2 = np.min((img_height - 1, y1 + np.max((0, h - 1))))
    if w * h > 0 and x2 > x1 and y2 > y1:
        bbox = np.array([x1, y1, x2 - x1, y2 - y1])
    else:
        bbox = None
    return bbox


def process_bbox(bbox, img_width, img_height, input_img_shape, ratio=1.25):
    """preprocessing.py:166-188: sanitize, grow to the aspect ratio of the network input, scale by ``ratio``."""
    bbox = sanitize_bbox(bbox, img_width, img_height)
    if bbox is None:
        return bbox
    w = bbox[2]
    h = bbox[3]
    c_x = bbox[0] + w / 2.
    c_y = bbox[1] + h / 2.
    aspect_ratio = input_img_shape[1] / input_img_shape[0]
    if w > aspect_ratio * h:
        h = w / aspect_ratio
    elif w < aspect_ratio * h:
        w = h * aspect_ratio
    bbox[2] = w * ratio
    bbox[3] = h * ratio
    bbox[0] = c_x - bbox[2] / 2.
    bbox[1] = c_y - bbox[3] / 2.
    bbox = bbox.astype(np.float32)
    return bbox

"""Camera maths of the reference's renderer module (hamer/hamer/utils/renderer.py:12-74).  The
pyrender/trimesh renderer classes themselves are visualisation and out of scope."""
import torch


def cam_crop_to_full(cam_bbox, box_center, box_size, img_size, focal_length=5000.):
    """renderer.py:12-22."""
    img_w, img_h = img_size[:, 0], img_size[:, 1]
    cx, cy, b = box_center[:, 0], box_center[:, 1], box_size
    w_2, h_2 = img_w / 2., img_h / 2.
    bs = b * cam_bbox[:, 0] + 1e-9
    tz = 2 * focal_length / bs
    tx = (2 * (cx - w_2) / bs) + cam_bbox[:, 1]
    ty = (2 * (cy - h_2) / bs) + cam_bbox[:, 2]
    return torch.stack([tx, ty, tz], dim=-1)


def custom_cam_crop_to_full(cam_bbox, box_center, box_size, img_size, fx, fy, cx, cy, depth_refine=None):
    """renderer.py:24-74 (the two progress prints of the reference are dropped)."""
    b = cam_bbox.shape[0]
    device = cam_bbox.device

    def to_tensor(val):
        if isinstance(val, (float, int)):
            return torch.full((b,), val, device=device).float()
        if isinstance(val, torch.Tensor):
            if val.dim() == 0:
                return val.unsqueeze(0).repeat(b).float()
            if val.dim() == 1 and val.shape[0] == 1:
                return val.repeat(b).float()
            return val.float()
        return torch.tensor(val, device=device).float()

    fx, fy, cx_real, cy_real = to_tensor(fx), to_tensor(fy), to_tensor(cx), to_tensor(cy)
    if depth_refine is not None:
        tz = to_tensor(depth_refine)
        bs = 2 * fx / (tz + 1e-9)
    else:
        bs = box_size * cam_bbox[:, 0] + 1e-9
        tz = 2 * fx / bs
    tx = (2 * (box_center[:, 0] - cx_real) / bs) + cam_bbox[:, 1]
    ty = (2 * (box_center[:, 1] - cy_real) / bs) + cam_bbox[:, 2]
    if not torch.allclose(fx, fy):
        ty = ty * (fx / fy)
    return torch.stack([tx, ty, tz], dim=-1)

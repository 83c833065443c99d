"""Crop camera -> full-image camera (reference behaviour: hamer/hamer/utils/renderer.py:12-74 ``cam_crop_to_full`` /
``custom_cam_crop_to_full``).  The weak-perspective camera (s, tx, ty) predicted on the crop is moved into the frame of
the full image: with ``bs = S * s`` the size of the hand box in focal-normalised units, depth is ``2 fx / bs`` and the
principal-point offset of the box centre adds ``2 (c_box - c) / bs``.  The pyrender/trimesh renderer classes of that module
are visualisation and out of scope."""
import torch


def _per_hand(value, n: int, like: torch.Tensor) -> torch.Tensor:
    """Scalar, 0-d / 1-element tensor, sequence or (n,) tensor -> float32 (n,) on ``like``'s device."""
    t = torch.as_tensor(value, device=like.device).to(torch.float32).reshape(-1)
    return t.expand(n) if t.numel() == 1 else t


def custom_cam_crop_to_full(cam_bbox, box_center, box_size, img_size, fx, fy, cx, cy, depth_refine=None):
    """cam_bbox (B,3) = (s, tx, ty) on the crop; box_center (B,2), box_size (B,) in frame pixels; intrinsics per hand or
    shared.  With ``depth_refine`` (metres, RootNet) the depth is taken from it and the box scale follows from the depth.
    ``ty`` is rescaled by fx / fy (a factor of exactly 1 when the focal lengths agree).  Returns (B,3) = (tx, ty, tz)."""
    n = cam_bbox.shape[0]
    f = torch.stack([_per_hand(fx, n, cam_bbox), _per_hand(fy, n, cam_bbox)], dim=1)            # (B, 2)
    pp = torch.stack([_per_hand(cx, n, cam_bbox), _per_hand(cy, n, cam_bbox)], dim=1)
    if depth_refine is None:
        bs = box_size * cam_bbox[:, 0] + 1e-9
        tz = 2 * f[:, 0] / bs
    else:
        tz = _per_hand(depth_refine, n, cam_bbox)
        bs = 2 * f[:, 0] / (tz + 1e-9)
    t_xy = 2 * (box_center[:, :2] - pp) / bs.unsqueeze(1) + cam_bbox[:, 1:3]
    # ty * fx / fy, unconditionally: with fx == fy the factor is exactly 1.0 and the product exact, and a data-dependent
    # `if` here would read a device value back -- a host sync in the middle of the folder drivers' pipeline (it was: 175 ms
    # of a 253 ms pass over 64 frames went to waiting in that test)
    t_xy = t_xy * torch.stack([torch.ones_like(tz), f[:, 0] / f[:, 1]], dim=1)
    return torch.cat([t_xy, tz.unsqueeze(1)], dim=1)


def cam_crop_to_full(cam_bbox, box_center, box_size, img_size, focal_length=5000.):
    """The same with one focal length and the principal point at the image centre (img_size (B,2) = (w, h))."""
    return custom_cam_crop_to_full(cam_bbox, box_center, box_size, img_size, focal_length, focal_length,
                                   img_size[:, 0] / 2., img_size[:, 1] / 2.)

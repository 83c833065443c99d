"""Geometry helpers kept for API parity (reference: hamer/hamer/utils/geometry.py).  On the hot
path rot6d_to_rotmat and the crop-space projection run inside the fused HIP MANO kernel; these
torch versions serve the full-image reprojection of estimate_from_rgb (21 points per hand)."""
from typing import Optional

import torch


def rot6d_to_rotmat(x: torch.Tensor) -> torch.Tensor:
    """6-D rotation representation -> rotation matrices (reference behaviour: geometry.py:47-70, the same Gram-Schmidt as
    the fused HIP MANO kernel): the first three numbers are the first COLUMN direction, the next three the second; columns
    (b1, b2, b1 x b2), norms floored at 1e-12."""
    a1, a2 = x.reshape(-1, 2, 3).unbind(dim=1)
    b1 = a1 / a1.norm(dim=1, keepdim=True).clamp_min(1e-12)
    u = a2 - (b1 * a2).sum(dim=1, keepdim=True) * b1
    b2 = u / u.norm(dim=1, keepdim=True).clamp_min(1e-12)
    return torch.stack((b1, b2, torch.linalg.cross(b1, b2, dim=1)), dim=-1)


def perspective_projection(points: torch.Tensor, translation: torch.Tensor, focal_length: torch.Tensor,
                           camera_center: Optional[torch.Tensor] = None,
                           rotation: Optional[torch.Tensor] = None) -> torch.Tensor:
    """geometry.py:72-110."""
    B = points.shape[0]
    if rotation is not None:
        points = torch.einsum("bij,bkj->bki", rotation, points)
    if camera_center is None:
        camera_center = torch.zeros(B, 2, device=points.device, dtype=points.dtype)
    p = points + translation.unsqueeze(1)
    p = p / p[:, :, -1].unsqueeze(-1)
    return p[:, :, :2] * focal_length.unsqueeze(1) + camera_center.unsqueeze(1) * p[:, :, 2:]

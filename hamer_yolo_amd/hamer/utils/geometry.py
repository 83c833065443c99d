"""Geometry helpers kept for API parity (reference: hamer/hamer/utils/geometry.py).  On the hot
path rot6d_to_rotmat and the crop-space projection run inside the fused HIP MANO kernel; these
torch versions serve the full-image reprojection of estimate_from_rgb (21 points per hand)."""
from typing import Optional

import torch


def rot6d_to_rotmat(x: torch.Tensor) -> torch.Tensor:
    """geometry.py:47-70."""
    x = x.reshape(-1, 2, 3).permute(0, 2, 1).contiguous()
    a1, a2 = x[:, :, 0], x[:, :, 1]
    b1 = torch.nn.functional.normalize(a1)
    b2 = torch.nn.functional.normalize(a2 - torch.einsum("bi,bi->b", b1, a2).unsqueeze(-1) * b1)
    b3 = torch.linalg.cross(b1, b2, dim=1)
    return torch.stack((b1, b2, b3), dim=-1)


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

"""The numeric half of the reference's reprojection check (reference: hamer/reconstruct.py:27-77): read the OBJ
written by ``reconstruct_and_save_obj_with_wrapper`` and project its camera-frame vertices with the intrinsics.
The overlay drawing itself (cv2.fillConvexPoly / addWeighted, :65-76) is visualisation and out of scope.

Formulated for whole meshes at once: the OBJ text is split into its record kinds in one pass and each kind is parsed
by one array conversion; the projection is one homogeneous product followed by one division."""
import os
import re

import numpy as np

_FACE_CORNER = re.compile(rb"(-?\d+)(?:/\S*)?")


def load_intrinsics(txt_path):
    """reconstruct.py:14-25: the 3x3 matrix of a whitespace-separated text file, None (with a message) when it cannot
    be had."""
    try:
        return np.loadtxt(txt_path)
    except OSError:
        print(f"[Error] intrinsics file not found: {txt_path}")
    except ValueError as e:
        print(f"[Error] cannot read intrinsics: {e}")
    return None


def load_obj(obj_path):
    """reconstruct.py:27-47: ``(vertices (V,3) float64, faces (F,3) int, 0-based)``; ``(None, None)`` when the file is
    missing.  Only ``v`` and ``f`` records count; of a face corner ``i/t/n`` the vertex index ``i`` is taken, of a
    polygon its first three corners (what the reference keeps)."""
    try:
        with open(obj_path, "rb") as fh:
            records = fh.read().splitlines()
    except OSError:
        return None, None
    vert_text = [r[2:] for r in records if r[:2] == b"v "]
    face_text = [r[2:] for r in records if r[:2] == b"f "]
    verts = np.array([t.split()[:3] for t in vert_text], dtype=np.float64).reshape(len(vert_text), 3) if vert_text else np.array([])
    corners = [[int(m) for m in _FACE_CORNER.findall(t)[:3]] for t in face_text]
    faces = np.asarray(corners, dtype=np.int64) - 1 if corners else np.array([])
    return verts, faces


def project_vertices(vertices, faces, K):
    """reconstruct.py:55-66: integer pixel coordinates (V,2) int32 of the pinhole projection ``K @ v`` and the painter's
    order of the faces, farthest first (by mean corner depth).  A vertex at depth exactly 0 is moved to 1e-5, as the
    reference does before dividing."""
    pts = np.asarray(vertices, dtype=np.float64)
    depth = np.where(pts[:, 2] == 0.0, 1e-5, pts[:, 2])
    cam = np.column_stack([pts[:, :2], depth])
    homo = cam @ np.asarray(K, dtype=np.float64).T                      # rows [x', y', w']
    pixels = (homo[:, :2] / homo[:, 2:3]).astype(np.int32)
    face_depth = depth[np.asarray(faces, dtype=np.int64)].mean(axis=1)
    far_to_near = np.argsort(face_depth)[::-1]
    return pixels, far_to_near

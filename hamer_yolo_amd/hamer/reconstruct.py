"""The numeric half of the reference's reprojection check (reference: hamer/reconstruct.py:27-77): read the OBJ
written by ``reconstruct_and_save_obj_with_wrapper`` and project its camera-frame vertices with the intrinsics.
The overlay drawing itself (cv2.fillConvexPoly / addWeighted, :65-76) is visualisation and out of scope."""
import os

import numpy as np


def load_intrinsics(txt_path):
    """reconstruct.py:14-25."""
    if not os.path.exists(txt_path):
        print(f"[Error] intrinsics file not found: {txt_path}")
        return None
    try:
        return np.loadtxt(txt_path)
    except Exception as e:
        print(f"[Error] cannot read intrinsics: {e}")
        return None


def load_obj(obj_path):
    """reconstruct.py:27-47: (vertices (V,3) float, faces (F,3) int, 0-based); (None, None) when the file is missing."""
    vertices, faces = [], []
    if not os.path.exists(obj_path):
        return None, None
    with open(obj_path, 'r') as f:
        for line in f:
            if line.startswith('v '):
                parts = line.strip().split()
                vertices.append([float(parts[1]), float(parts[2]), float(parts[3])])
            elif line.startswith('f '):
                parts = line.strip().split()
                idx = [int(p.split('/')[0]) - 1 for p in parts[1:]]
                faces.append(idx[:3])
    return np.array(vertices), np.array(faces)


def project_vertices(vertices, faces, K):
    """reconstruct.py:55-66: pixel coordinates (V,2) int32 of ``K @ v`` (zero depths nudged to 1e-5 as the reference
    does) and the far-to-near face order used for painting."""
    vertices = np.array(vertices, dtype=np.float64, copy=True)
    Z = vertices[:, 2]
    Z[Z == 0] = 1e-5
    projected_homo = (np.asarray(K, dtype=np.float64) @ vertices.T).T
    u = projected_homo[:, 0] / projected_homo[:, 2]
    v = projected_homo[:, 1] / projected_homo[:, 2]
    pixels = np.stack([u, v], axis=1).astype(np.int32)
    order = np.argsort(np.mean(Z[faces], axis=1))[::-1]
    return pixels, order

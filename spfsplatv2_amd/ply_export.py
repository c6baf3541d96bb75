"""Export a scene's Gaussians as a 3DGS-viewer ``.ply`` -- host-side mirror of the reference's ``export_ply``
(/root/reference/src/model/ply_export.py:76-142): same signature, same vertex layout (x y z nx ny nz f_dc_0..2
opacity scale_0..2 rot_0..3, float32, binary little-endian), same scene normalisation (median to origin, 95th
percentile to unit range), same viewer rotation (+Z up, -45 degrees about Z, composed with the camera's
world-to-camera rotation), log-scales, wxyz quaternions, DC band only.  No plyfile / scipy dependency."""
from __future__ import annotations

import math
from pathlib import Path

import numpy as np
import torch
from torch import Tensor

ATTRIBUTES = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2", "opacity",
              "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]


def _quat_xyzw_to_matrix(q: np.ndarray) -> np.ndarray:
    q = q / np.linalg.norm(q, axis=-1, keepdims=True)
    x, y, z, w = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    m = np.empty((q.shape[0], 3, 3), dtype=np.float64)
    m[:, 0, 0] = 1 - 2 * (y * y + z * z); m[:, 0, 1] = 2 * (x * y - w * z); m[:, 0, 2] = 2 * (x * z + w * y)
    m[:, 1, 0] = 2 * (x * y + w * z); m[:, 1, 1] = 1 - 2 * (x * x + z * z); m[:, 1, 2] = 2 * (y * z - w * x)
    m[:, 2, 0] = 2 * (x * z - w * y); m[:, 2, 1] = 2 * (y * z + w * x); m[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return m


def _matrix_to_quat_xyzw(m: np.ndarray) -> np.ndarray:
    """Shepperd's method, vectorised; returns unit quaternions (x, y, z, w)."""
    n = m.shape[0]
    q = np.empty((n, 4), dtype=np.float64)
    tr = m[:, 0, 0] + m[:, 1, 1] + m[:, 2, 2]
    choice = np.argmax(np.stack([m[:, 0, 0], m[:, 1, 1], m[:, 2, 2], tr], axis=1), axis=1)
    for i in range(3):
        sel = choice == i
        if not sel.any():
            continue
        j, k = (i + 1) % 3, (i + 2) % 3
        mm = m[sel]
        q[sel, i] = 1 - tr[sel] + 2 * mm[:, i, i]
        q[sel, j] = mm[:, j, i] + mm[:, i, j]
        q[sel, k] = mm[:, k, i] + mm[:, i, k]
        q[sel, 3] = mm[:, k, j] - mm[:, j, k]
    sel = choice == 3
    if sel.any():
        mm = m[sel]
        q[sel, 0] = mm[:, 2, 1] - mm[:, 1, 2]
        q[sel, 1] = mm[:, 0, 2] - mm[:, 2, 0]
        q[sel, 2] = mm[:, 1, 0] - mm[:, 0, 1]
        q[sel, 3] = 1 + tr[sel]
    return q / np.linalg.norm(q, axis=1, keepdims=True)


def ply_vertex_table(extrinsics: Tensor, means: Tensor, scales: Tensor, rotations: Tensor, harmonics: Tensor,
                     opacities: Tensor) -> np.ndarray:
    """[G,17] float32 table in ATTRIBUTES order (what ``export_ply`` writes)."""
    means = means - means.median(dim=0).values                       # median Gaussian to the origin
    scale_factor = means.abs().quantile(0.95, dim=0).max()           # most Gaussians within [-1, 1]
    means = means / scale_factor
    scales = scales / scale_factor
    rotation = torch.tensor([[0, 0, 1], [-1, 0, 0], [0, -1, 0]], dtype=torch.float32, device=means.device)
    a = math.radians(-45.0)                                          # viewer starts at 45 degrees: rotate about Z
    adjustment = torch.tensor([[math.cos(a), -math.sin(a), 0], [math.sin(a), math.cos(a), 0], [0, 0, 1]],
                              dtype=torch.float32, device=means.device)
    rotation = adjustment @ rotation
    rotation = rotation @ extrinsics[:3, :3].inverse()               # default view = camera space
    means = means @ rotation.T
    rot_np = rotation.detach().cpu().numpy().astype(np.float64)
    mats = rot_np @ _quat_xyzw_to_matrix(rotations.detach().cpu().numpy().astype(np.float64))
    x, y, z, w = _matrix_to_quat_xyzw(mats).T
    quat_wxyz = np.stack((w, x, y, z), axis=-1)
    cols = (means.detach().cpu().numpy(), np.zeros((means.shape[0], 3), np.float32),
            harmonics[..., 0].detach().cpu().contiguous().numpy(), opacities[..., None].detach().cpu().numpy(),
            scales.log().detach().cpu().numpy(), quat_wxyz)
    return np.concatenate(cols, axis=1).astype(np.float32)


def export_ply(extrinsics: Tensor, means: Tensor, scales: Tensor, rotations: Tensor, harmonics: Tensor,
               opacities: Tensor, path: Path) -> None:
    """extrinsics [4,4], means [G,3], scales [G,3], rotations [G,4] (xyzw), harmonics [G,3,d_sh], opacities [G]."""
    table = ply_vertex_table(extrinsics, means, scales, rotations, harmonics, opacities)
    path = Path(path)
    path.parent.mkdir(exist_ok=True, parents=True)
    header = ["ply", "format binary_little_endian 1.0", f"element vertex {table.shape[0]}"]
    header += [f"property float {a}" for a in ATTRIBUTES] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(header) + "\n").encode("ascii"))
        f.write(np.ascontiguousarray(table, dtype="<f4").tobytes())


def read_ply(path: Path) -> np.ndarray:
    """Minimal reader for files written by ``export_ply`` -> [G,17] float32."""
    raw = Path(path).read_bytes()
    end = raw.index(b"end_header\n") + len(b"end_header\n")
    head = raw[:end].decode("ascii").splitlines()
    n = int(next(h for h in head if h.startswith("element vertex")).split()[-1])
    props = [h.split()[-1] for h in head if h.startswith("property float")]
    assert props == ATTRIBUTES, props
    return np.frombuffer(raw[end:], dtype="<f4").reshape(n, len(props))

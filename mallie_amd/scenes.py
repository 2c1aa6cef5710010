"""Synthetic benchmark scenes of BASELINE.json / SURVEY.md 8(d): the suzanne grid (C4: 32x32 -> 991 232 triangles,
C5: 102x102 -> 10 071 072 triangles).  Pure numpy; the mesh arrays have the same meaning as the reference's Mesh."""
import numpy as np

SUZANNE_VERTS = slice(24, 531)   # vertices 25..531 of cornellbox_suzanne.obj = object `suzanne_tri`
SUZANNE_FACES = slice(12, 980)   # its 968 triangles (after the six 2-triangle planes)


def face_normals(verts, faces):
    """Facevarying normals as MeshLoader::LoadObj computes them for a file without `vn` (mesh_loader.cc:15-22,133-171):
    normalize(cross(v2 - v0, v1 - v0)) with the 1e-6 length guard, repeated for the three corners."""
    v0, v1, v2 = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    a, b = v1 - v0, v2 - v0
    n = np.stack([b[:, 1] * a[:, 2] - b[:, 2] * a[:, 1], b[:, 2] * a[:, 0] - b[:, 0] * a[:, 2],
                  b[:, 0] * a[:, 1] - b[:, 1] * a[:, 0]], axis=1)
    ln = np.sqrt(n[:, 0] * n[:, 0] + n[:, 1] * n[:, 1] + n[:, 2] * n[:, 2])
    inv = np.where(np.abs(ln) > 1.0e-6, 1.0 / np.where(ln == 0, 1.0, ln), 1.0)
    n = n * inv[:, None]
    return np.ascontiguousarray(np.tile(n, (1, 3)))


def suzanne_grid(cornell_verts, cornell_faces, n):
    """n x n copies of suzanne_tri on the XZ plane, pitch = 1.1 x the object's largest bbox extent, copy (gx, gz) moved
    by ((gx - n//2)*pitch, 0, (gz - n//2)*pitch); coordinates rounded to float32 like every mesh the reference loads.
    Returns (verts float64 [V,3], faces uint32 [F,3], matIDs uint32 [F], normals float64 [F,9])."""
    sv = np.asarray(cornell_verts, np.float64)[SUZANNE_VERTS]
    sf = np.asarray(cornell_faces, np.int64)[SUZANNE_FACES] - SUZANNE_VERTS.start
    assert sf.min() == 0 and sf.max() == len(sv) - 1
    pitch = 1.1 * float((sv.max(0) - sv.min(0)).max())
    g = np.arange(n) - n // 2
    gx, gz = np.meshgrid(g, g, indexing="ij")
    off = np.stack([gx.ravel() * pitch, np.zeros(n * n), gz.ravel() * pitch], axis=1)          # [n*n, 3]
    verts = (sv[None, :, :] + off[:, None, :]).astype(np.float32).astype(np.float64).reshape(-1, 3)
    faces = (sf[None, :, :] + (np.arange(n * n) * len(sv))[:, None, None]).reshape(-1, 3).astype(np.uint32)
    mats = np.zeros(len(faces), np.uint32)
    return verts, faces, mats, face_normals(verts, faces)

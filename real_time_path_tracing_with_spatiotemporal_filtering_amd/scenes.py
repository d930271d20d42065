"""Synthetic scenes for the divergent-traversal configuration (BASELINE.json configs[4]):
the reference builds one BLAS with one identity instance of a 32-triangle mesh (main.cpp:728-741), so
"instanced Cornell box x1000 (~1M tris)" is generated here: every quad of the OBJ is tessellated n x n
(the normal-keyed albedo, raytrace.comp.glsl:155-163, is tessellation-invariant) and the mesh is
instanced on a lattice of translations passed to rtpt_scene_upload as 3x4 transforms."""
from __future__ import annotations

import numpy as np


def tessellate_quads(xyz: np.ndarray, idx: np.ndarray, n: int):
    """Split each fan-triangulated quad (triangles 2q = (a,b,c), 2q+1 = (a,c,d)) into an n x n grid of
    cells, two triangles per cell, same winding.  n = 6 turns 32 triangles into 1,152."""
    xyz = np.asarray(xyz, np.float32)
    idx = np.asarray(idx, np.uint32).reshape(-1, 3)
    if n <= 1:
        return xyz.copy(), idx.copy()
    assert len(idx) % 2 == 0
    verts, tris = [], []
    t = np.linspace(0.0, 1.0, n + 1, dtype=np.float32)
    for q in range(len(idx) // 2):
        a, b, c = idx[2 * q]
        a2, c2, d = idx[2 * q + 1]
        assert a == a2 and c == c2, "mesh is not a fan-triangulated quad list"
        A, B, C, D = (xyz[i] for i in (a, b, c, d))
        base = sum(len(v) for v in verts)
        # bilinear grid: P(s,t) = (1-s)(1-t)A + s(1-t)B + s t C + (1-s) t D, evaluated in float32
        S, T = np.meshgrid(t, t, indexing="ij")
        S, T = S[..., None], T[..., None]
        P = ((1 - S) * (1 - T)) * A + (S * (1 - T)) * B + (S * T) * C + ((1 - S) * T) * D
        verts.append(P.reshape(-1, 3).astype(np.float32))
        for i in range(n):
            for j in range(n):
                p00 = base + i * (n + 1) + j
                p10 = base + (i + 1) * (n + 1) + j
                p11 = base + (i + 1) * (n + 1) + j + 1
                p01 = base + i * (n + 1) + j + 1
                tris.append((p00, p10, p11))
                tris.append((p00, p11, p01))
    return np.concatenate(verts).astype(np.float32), np.array(tris, np.uint32)


def lattice_xforms(nx: int, ny: int, nz: int, pitch: float = 2.5) -> np.ndarray:
    """nx*ny*nz translations (3x4 row-major), lattice centred on x and z, resting on y = 0, the
    nearest layer's front face staying near z = +1 like the single box."""
    out = []
    for iz in range(nz):
        for iy in range(ny):
            for ix in range(nx):
                m = np.zeros((3, 4), np.float32)
                m[0, 0] = m[1, 1] = m[2, 2] = 1.0
                m[0, 3] = np.float32((ix - (nx - 1) / 2.0) * pitch)
                m[1, 3] = np.float32(iy * pitch)
                m[2, 3] = np.float32(-iz * pitch)
                out.append(m)
    return np.stack(out).reshape(-1, 12)


def instanced_cornell(xyz, idx, lattice=(10, 10, 10), tess=6, pitch=2.5):
    """(xyz, idx, xforms, camera, z_far) of the configs[4] scene; 10^3 x 32 x 2*6^2/2 = 1,152,000 triangles"""
    vx, ti = tessellate_quads(xyz, idx, tess)
    xf = lattice_xforms(*lattice, pitch=pitch)
    nx, ny, nz = lattice
    height = (ny - 1) * pitch + 2.0
    width = (nx - 1) * pitch + 2.0
    # K2 camera: vertical slope tan(0.2); frame the front layer with a margin
    dist = max(height, width * 9.0 / 16.0) / 2.0 / 0.2027 * 1.15
    camera = (np.float32(-0.001), np.float32(height / 2.0), np.float32(1.0 + dist))
    z_far = float(dist + nz * pitch + 10.0)
    return vx, ti, xf, camera, z_far

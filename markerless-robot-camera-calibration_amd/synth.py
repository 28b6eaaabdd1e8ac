"""Deterministic synthetic inputs: point clouds, colours, labels, EE crops, key points.

No reference data ships (sample frames are missing blobs), so every parity test and the
benchmark use these generators.  Pure numpy; identical on the build container and on the GPU box.
"""
import numpy as np


def gen_room(n: int, L: float = 2.4, seed: int = 0, sigma: float = 0.002):
    """n points on the six faces of an obliquely rotated cube of side L, pushed to z ~ L.

    Returns (points float32[n,3], rgb float32[n,3] in [-0.5,0.5), labels int64[n] in {0,1,2}).
    Labels are the face-pair id (the axis the face is normal to) -> a 3-class ground truth.
    """
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(Q) < 0:
        Q[:, 0] = -Q[:, 0]
    face = rng.integers(0, 6, size=n)
    uv = rng.uniform(-L / 2, L / 2, size=(n, 2))
    normal = np.where(face % 2 == 0, L / 2, -L / 2) + rng.normal(0.0, sigma, size=n)
    axis = face // 2
    pts = np.empty((n, 3), dtype=np.float64)
    for a in range(3):
        m = axis == a
        other = [i for i in range(3) if i != a]
        pts[m, a] = normal[m]
        pts[m, other[0]] = uv[m, 0]
        pts[m, other[1]] = uv[m, 1]
    pts = pts @ Q.T + np.array([0.0, 0.0, L])
    rgb = rng.uniform(0.0, 1.0, size=(n, 3)) - 0.5
    return pts.astype(np.float32), rgb.astype(np.float32), axis.astype(np.int64)

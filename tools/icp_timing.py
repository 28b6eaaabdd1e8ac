"""Timing of sv_icp_point2point (8192 CAD points against an end-effector crop, <= 30 iterations)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mrcc_amd  # noqa: E402,F401
from mrcc_amd.utils import icp as I  # noqa: E402

rng = np.random.default_rng(0)
src = rng.uniform(-0.1, 0.1, size=(8192, 3)).astype(np.float32)
t = np.array([0.01, -0.005, 0.004], dtype=np.float32)
for nt in (2000, 8000):
    tgt = (src[rng.choice(8192, nt)] + t + rng.normal(size=(nt, 3)).astype(np.float32) * 1e-3).astype(np.float32)
    s_d, t_d = torch.from_numpy(src).cuda(), torch.from_numpy(tgt).cuda()
    for _ in range(3):
        out = I.icp_point2point(s_d, t_d, np.eye(4))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = I.icp_point2point(s_d, t_d, np.eye(4))
    torch.cuda.synchronize()
    print(f"{nt} target points: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms per ICP call, {out[3]} updates, "
          f"fitness {out[1]:.3f}, rmse {out[2] * 1e3:.2f} mm")

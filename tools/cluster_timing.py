"""Largest single-linkage cluster (ClusterUtil.get_largest_cluster): device path vs the host k-d-tree path on EE-like
point sets (a 10 x 22 x 13 cm box of points + 10 % scattered false positives, 6 cm threshold).
    python tools/cluster_timing.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mrcc_amd  # noqa: E402,F401
from mrcc_amd.utils.output import ClusterUtil  # noqa: E402

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
cu = ClusterUtil()
for n in (2000, 4096, 10000, 30000, 100000):
    pts = np.concatenate([rng.uniform([-0.05, -0.11, -0.065], [0.05, 0.11, 0.065], (int(n * 0.9), 3)),
                          rng.uniform(-1, 1, (n - int(n * 0.9), 3))]).astype(np.float32)
    pts = pts[rng.permutation(n)]
    t = torch.from_numpy(pts).to(dev)
    for _ in range(2):
        got = cu.get_largest_cluster(t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        got = cu.get_largest_cluster(t)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / reps * 1e3
    host = ""
    if n <= 10000:
        t0 = time.perf_counter()
        want = cu.get_largest_cluster(pts)
        host = f"host path {(time.perf_counter() - t0) * 1e3:8.1f} ms, equal: {np.array_equal(got.cpu().numpy(), want)}"
    print(f"n = {n:6d}: device {gpu_ms:7.3f} ms per call (members {got.numel()})  {host}", flush=True)

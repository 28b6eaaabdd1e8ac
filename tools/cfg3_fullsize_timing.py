"""Cfg-3 at full size: 64 Cfg-2 frames (200k points each) in ONE sparse tensor through RobotNetSegmentation(MinkUNet18D)
(+ slice/argmax) - which kernel instances run (sv_conv_last_instance) and frames/s.  Tensors beyond the 2 GB extent of the
buffer-addressed instances run as batch ranges (ConvPlan.chunks)."""
import collections
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench  # noqa: E402
import mrcc_amd  # noqa: E402
from mrcc_amd import MinkowskiEngine as ME  # noqa: E402
from mrcc_amd import profiling  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
with torch.no_grad():
    model = bench.build_model(dev)
    coords, feats, *_ = bench.make_frame(0, dev, batch=B)

    def step():
        field = ME.TensorField(feats, coords, device=dev)
        x = field.sparse()
        return model(x).slice_argmax(field)[0], x.F.shape[0]

    profiling.INSTANCE_LOG = log = []
    _, V = step()
    profiling.INSTANCE_LOG = None
    torch.cuda.synchronize()
    cnt = collections.Counter((e[0], e[1]["fast"]) for e in log)
    for (name, fast), n in sorted(cnt.items()):
        print(f"  {n:4d} x {name} fast={fast}")
    for reps in (1, 3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"Cfg-3 full size: {B} frames x 200k pts in one tensor, {V} voxels: {dt * 1e3:.1f} ms/batch = {B / dt:.1f} frames/s "
              f"(max memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB)")

"""Cfg-5 (500k points, 1 cm voxels) through the headline pipeline, for a kernel trace:
    rocprofv3 --kernel-trace --stats -- python3 tools/cfg5_profile.py [frames]
prints ms per frame, active voxels, algorithmic GFLOP per frame and the end-to-end fraction of the fp32 matrix peak."""
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
from mrcc_amd.app.pipeline import FramePipeline  # noqa: E402

dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
with torch.no_grad():
    model = bench.build_model(dev)
    frames = []
    for s in range(2):
        pts, rgb, _ = mrcc_amd.synth.gen_room(500_000, 2.4, s)
        c4 = np.concatenate([np.zeros((len(pts), 1), np.float32), pts * np.float32(100)], axis=1)
        frames.append((torch.from_numpy(c4).to(dev), torch.from_numpy(rgb).to(dev)))
    pipe = FramePipeline(dev, levels=4, compute_streams=3)
    bench.run_frames(model, pipe, frames, 4)
    pipe.drain()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vox = bench.run_frames(model, pipe, frames, n)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    x = ME.TensorField(frames[0][1], frames[0][0], device=dev).sparse()
    gf = profiling.pass_gflop(lambda: model(x))
    print(f"cfg5 seg pipeline: {dt * 1e3:.2f} ms/frame = {1 / dt:.2f} frames/s, {int(vox) // n} voxels/frame, {gf:.0f} GFLOP/frame "
          f"-> {gf / dt / 1e3:.1f} TFLOP/s = {gf / dt / 1e3 / 157.3:.3f} of the fp32 matrix peak end to end")

"""The prep stream's work for one Cfg-2 frame (voxelise, coordinate maps, kernel maps, plans) ALONE on the GPU, N times -
run under `rocprofv3 --kernel-trace --stats` for its kernel time per frame; prints host wall time per prepare()."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from mrcc_amd.app.pipeline import FramePipeline  # noqa: E402

dev = torch.device("cuda:0")
frames = [bench.make_frame(i, dev) for i in range(4)]
pipe = FramePipeline(dev, levels=4)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for i in range(3):
    pipe.prepare(*frames[i % 4][:2])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    pipe.prepare(*frames[i % 4][:2])
torch.cuda.synchronize()
print(f"prepare alone: {(time.perf_counter() - t0) / n * 1e3:.2f} ms host wall per frame ({n} frames + 3 warm-up)")

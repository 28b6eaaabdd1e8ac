"""GPU memory of the frame pipeline over 600 frames (no growth expected: frames are retired once their stream is done and a
plan holds only a weak reference to its coordinate manager).  python tools/pipeline_memory.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench  # noqa: E402
from mrcc_amd.app.pipeline import FramePipeline  # noqa: E402

dev = torch.device("cuda:0")
model = bench.build_model(dev)
frames = [bench.make_frame(i, dev) for i in range(4)]
pipe = FramePipeline(dev, levels=4, compute_streams=3)
with torch.no_grad():
    for rep in range(4):
        bench.run_frames(model, pipe, frames, 150)
        pipe.drain()
        torch.cuda.synchronize()
        print("after %3d frames: allocated %.2f GiB, reserved %.2f GiB, max allocated %.2f GiB" % (
            (rep + 1) * 150, torch.cuda.memory_allocated() / 2 ** 30, torch.cuda.memory_reserved() / 2 ** 30,
            torch.cuda.max_memory_allocated() / 2 ** 30))

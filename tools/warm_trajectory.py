"""How long does a fresh process take to reach steady frame times?  Prints ms/frame per block of 10 frames."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mrcc_amd.app.pipeline import FramePipeline
streams = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
model = bench.build_model(dev)
frames = [bench.make_frame(i, dev) for i in range(4)]
pipe = FramePipeline(dev, levels=4, compute_streams=streams)
t_start = time.perf_counter()
with torch.no_grad():
    for blk in range(int(sys.argv[2]) if len(sys.argv) > 2 else 20):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        bench.run_frames(model, pipe, frames, 10)
        pipe.drain(); torch.cuda.synchronize()
        print(f"t={time.perf_counter() - t_start:6.2f}s block {blk:2d}: {(time.perf_counter() - t0) * 100:.2f} ms/frame", flush=True)

"""Where does the host spend a frame?  Times prepare() and run() (enqueue only) of the two-stream pipeline."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mrcc_amd.app.pipeline import FramePipeline
dev = torch.device("cuda:0")
model = bench.build_model(dev)
frames = [bench.make_frame(i, dev) for i in range(4)]
for prio in (0, -1):
    pipe = FramePipeline(dev, levels=4, compute_streams=2)
    if prio:
        pipe.prep_stream = torch.cuda.Stream(device=dev, priority=prio)
    def unet(x, field):
        out = model(x); return out.slice_argmax(field)[0]
    with torch.no_grad():
        nxt = pipe.prepare(*frames[0][:2])
        for i in range(3):
            cur = nxt; pipe.run(cur, unet); nxt = pipe.prepare(*frames[(i + 1) % 4][:2])
        torch.cuda.synchronize()
        tp = tr = 0.0
        t0 = time.perf_counter()
        n = 10
        for i in range(n):
            cur = nxt
            a = time.perf_counter(); pipe.run(cur, unet); b = time.perf_counter()
            nxt = pipe.prepare(*frames[(i + 1) % 4][:2]); c = time.perf_counter()
            tr += b - a; tp += c - b
        torch.cuda.synchronize()
        tot = time.perf_counter() - t0
    print(f"prio={prio}: per frame wall {tot / n * 1e3:.2f} ms; host run-enqueue {tr / n * 1e3:.2f} ms; host prepare {tp / n * 1e3:.2f} ms")
# prepare alone on an idle GPU
pipe = FramePipeline(dev, levels=4, compute_streams=2)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(5):
    pipe.prepare(*frames[i % 4][:2]); torch.cuda.synchronize()
print(f"prepare alone (idle GPU): {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")

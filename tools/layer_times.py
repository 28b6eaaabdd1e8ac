"""Per-layer table of one cfg2 frame (single stream, every conv launch event-timed): where do the milliseconds go?"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from mrcc_amd import profiling
from mrcc_amd.app.pipeline import FramePipeline

dev = torch.device("cuda:0")
model = bench.build_model(dev)
frames = [bench.make_frame(i, dev) for i in range(2)]
pipe = FramePipeline(dev, levels=4)


def unet(x, field):
    return model(x).slice_argmax(field)[0]


with torch.no_grad():
    for i in range(3):
        pipe.run(pipe.prepare(*frames[i % 2][:2]), unet)
    torch.cuda.synchronize()
    acc = {}
    REP = 5
    for rep in range(REP):
        cur = pipe.prepare(*frames[0][:2])
        torch.cuda.synchronize()
        t = profiling.KernelTimer(capacity=600)
        profiling.TIMER = t
        pipe.run(cur, unet)
        torch.cuda.synchronize()
        profiling.TIMER = None
        for idx, (kernel, K, Cin, Cout, V, pairs, s, e, _first) in enumerate(t.records):
            P = int(pairs.item()) if pairs is not None else V
            a = acc.setdefault(idx, [kernel, K, Cin, Cout, V, P, 0.0])
            a[6] += s.elapsed_time(e) / REP
tot = 0.0
print(f"{'#':>3} {'kernel':28} {'K':>3} {'Cin':>4} {'Cout':>4} {'V_out':>7} {'pairs':>8} {'ms':>7} {'TF':>6} {'gatherGB/s':>10}")
for idx in sorted(acc):
    kernel, K, Cin, Cout, V, P, ms = acc[idx]
    tot += ms
    print(f"{idx:3d} {kernel:28} {K:3d} {Cin:4d} {Cout:4d} {V:7d} {P:8d} {ms:7.3f} {2.0 * P * Cin * Cout / ms / 1e9:6.1f} "
          f"{(P * (4.0 * Cin + 8) + 4.0 * V * Cout) / ms / 1e6:10.0f}")
print(f"sum of timed conv launches: {tot:.3f} ms")

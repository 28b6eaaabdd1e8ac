"""Does HIP stream priority protect a chip-filling conv from latency-bound small launches on another stream?
Stream A runs level-0 384->384 convs back to back; stream B runs a chain of small-level convs (levels 2, 3, 4) in a
loop.  Reports A's time per launch and B's chain time for every (priority A, priority B) pair, and each alone."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import mrcc_amd  # noqa: E402
from mrcc_amd import MinkowskiEngine as ME  # noqa: E402
from mrcc_amd import nn as svnn  # noqa: E402

dev = torch.device("cuda:0")
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
pts, rgb, _ = mrcc_amd.synth.gen_room(200000, 2.4, 0)
coords4 = np.concatenate([np.zeros((len(pts), 1), np.float32), pts * np.float32(50)], axis=1)
x = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(coords4), device=dev).sparse()
cm = x.coordinate_manager
plans = {l: cm.plan_k3(1 << l) for l in range(5)}
V = {l: cm.stride_map(1 << l).V for l in range(5)}
torch.manual_seed(0)
f0 = torch.randn(V[0], 384, device=dev)
W = torch.randn(27, 384, 384, device=dev) * 0.05
small = [(l, torch.randn(V[l], c, device=dev), torch.randn(27, c, c, device=dev) * 0.05)
         for l, c in ((2, 384), (3, 384), (4, 256), (3, 128), (2, 64))]
torch.cuda.synchronize()


def run(pa, pb, big=True, little=True, nbig=40):
    sa = torch.cuda.Stream(device=dev, priority=pa)
    sb = torch.cuda.Stream(device=dev, priority=pb)
    ea, eb = [torch.cuda.Event(enable_timing=True) for _ in range(2)], [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize()
    nchain = 0
    if big:
        with torch.cuda.stream(sa):
            ea[0].record()
            for _ in range(nbig):
                svnn.conv_forward(f0, W, plans[0], V[0])
            ea[1].record()
    if little:
        with torch.cuda.stream(sb):
            eb[0].record()
            for _ in range(nbig * 2 if big else 40):
                for l, f, w in small:
                    svnn.conv_forward(f, w, plans[l], V[l])
                nchain += 1
            eb[1].record()
    torch.cuda.synchronize()
    ta = ea[0].elapsed_time(ea[1]) / nbig if big else float("nan")
    tb = eb[0].elapsed_time(eb[1]) / nchain if little else float("nan")
    return ta, tb


run(0, 0)
print("big alone      : %.3f ms/launch" % run(0, 0, True, False)[0])
print("small chain alone: %.3f ms/chain (5 launches)" % run(0, 0, False, True)[1])
for pa, pb in ((0, 0), (-1, 0), (0, -1), (-1, -1)):
    ta, tb = run(pa, pb)
    print(f"prio big={pa:2d} small={pb:2d}: big {ta:.3f} ms/launch, small chain {tb:.3f} ms/chain (while the big stream is busy + after)")

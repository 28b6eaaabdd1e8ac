"""Fixed cost of a launch in a back-to-back series on one stream (HIP events around 200 launches): a 1-row sv_affine_act,
and the gather layers at 1/16 of the frame - what the `hbm_bound_layers` figures contain besides moving bytes."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mrcc_amd  # noqa: E402
from mrcc_amd import MinkowskiEngine as ME  # noqa: E402
from mrcc_amd import nn as svnn  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, n=200):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


tiny = torch.randn(1, 32, device=dev)
print(f"sv_affine_act on one row: {timed(lambda: svnn.affine_act(tiny, act=1)):.2f} us per launch (incl. the output allocation)")
for npts, L in ((200_000, 2.4), (50_000, 1.2), (12_500, 0.6), (800_000, 4.8)):
    pts, rgb, _ = mrcc_amd.synth.gen_room(npts, L, 0)
    coords4 = np.concatenate([np.zeros((len(pts), 1), np.float32), pts * np.float32(50)], axis=1)
    x = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(coords4), device=dev).sparse()
    cm = x.coordinate_manager
    for name, level, cin in (("conv0 3->32", 0, 3), ("32->32 level 1", 1, 32), ("32->32 level 0", 0, 32)):
        plan = cm.plan_k3(1 << level)
        V = cm.stride_map(1 << level).V
        feats = x.F if cin == 3 else torch.randn(V, cin, device=dev)
        W = torch.randn(27, cin, 32, device=dev) * 0.1
        out = torch.empty(V, 32, device=dev)
        us = timed(lambda: svnn.conv_forward(feats, W, plan, V, None, None, None, 1, out=out))
        P = plan.num_pairs()
        gb = (P * (4.0 * cin + 8) + 4.0 * V * 32 + 4.0 * 27 * cin * 32) / 1e9
        print(f"{npts:7d} pts  {name:15s} V={V:7d}: {us:7.2f} us  {gb / us * 1e6:7.1f} GB/s = {gb / us * 1e6 / 8000:.3f} of peak")

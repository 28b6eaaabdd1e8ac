"""phase stamps of the thin 32->32 kernel (SV_THIN_TRACE=1): python tools/thin_trace.py"""
import os, sys
os.environ["SV_THIN_TRACE"] = "1"
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import mrcc_amd
from mrcc_amd import MinkowskiEngine as ME
from mrcc_amd import nn as svnn
dev = torch.device("cuda:0")
mrcc_amd._lib.call("sv_conv_set_dispatch", __import__("ctypes").c_double(1.0), __import__("ctypes").c_double(-1.0))
pts, rgb, _ = mrcc_amd.synth.gen_room(200_000, 2.4, 0)
c4 = np.concatenate([np.zeros((len(pts), 1), np.float32), pts * np.float32(50)], axis=1)
x = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(c4), device=dev).sparse()
cm = x.coordinate_manager
for level in (0, 1, 2):
    plan = cm.plan_k3(1 << level); V = cm.stride_map(1 << level).V
    f = torch.randn(V, 32, device=dev); W = torch.randn(27, 32, 32, device=dev) * 0.1
    for _ in range(3):
        svnn.conv_forward(f, W, plan, V)
    torch.cuda.synchronize()

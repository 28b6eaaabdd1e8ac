"""Single-layer microbenchmark of sv_conv_fwd on the Cfg-2 cloud (200k pts, 2 cm): k27 Cin->Cout at a chosen level.
    python tools/conv_microbench.py [--cin 384 --cout 384 --level 0 --iters 10]
Prints algorithmic TFLOP/s (2 P Cin Cout / time) and the plan's MFMA row-slot efficiency."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mrcc_amd  # noqa: E402
from mrcc_amd import MinkowskiEngine as ME  # noqa: E402
from mrcc_amd import nn as svnn  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cin", type=int, default=384)
ap.add_argument("--cout", type=int, default=384)
ap.add_argument("--level", type=int, default=0)
ap.add_argument("--iters", type=int, default=200)  # long enough for the clocks to settle
ap.add_argument("--points", type=int, default=200000)
ap.add_argument("--scale", type=float, default=50.0, help="voxels per metre (50 = 2 cm: Cfg-2; 100 with --points 500000 = Cfg-5)")
ap.add_argument("--kind", default="k3")
ap.add_argument("--cube", type=int, default=0, help="solid cube of this edge length (voxels) instead of the room cloud")
ap.add_argument("--split", default="0", help="run the 3x3x3 layer as passes over offset ranges: 14 -> [0,14) [14,27); 9,18 -> three")
ap.add_argument("--frames", type=int, default=1, help="this many room clouds (seeds 0, 1, ...) in one sparse tensor: the headline's launch size is 4")
args = ap.parse_args()

dev = torch.device("cuda:0")
if args.cin == 32 and args.cout == 32:  # the thin layers: one layer alone on the GPU = the one-frame dispatch (LDS-weights kernel)
    import ctypes

    mrcc_amd._lib.call("sv_conv_set_dispatch", ctypes.c_double(1.0), ctypes.c_double(-1.0))
if args.cube:
    g = np.arange(args.cube, dtype=np.float32) + 0.5
    xyz = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
    coords4 = np.concatenate([np.zeros((len(xyz), 1), np.float32), xyz], axis=1)
    rgb = np.zeros((len(xyz), 3), np.float32)
else:
    clouds = [mrcc_amd.synth.gen_room(args.points, 2.4, sd) for sd in range(args.frames)]
    coords4 = np.concatenate([np.concatenate([np.full((len(c[0]), 1), b, np.float32), c[0] * np.float32(args.scale)], axis=1)
                              for b, c in enumerate(clouds)])
    rgb = np.concatenate([c[1] for c in clouds])
x = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(coords4), device=dev).sparse()
cm = x.coordinate_manager
ts = 2 ** args.level
if args.kind == "k3":
    plan = cm.plan_k3(ts)
    V_in = V = cm.stride_map(ts).V
    K = 27
elif args.kind == "dense":
    plan = None
    V_in = V = cm.stride_map(ts).V
    K = 1
torch.manual_seed(0)
feats = torch.randn(V_in, args.cin, device=dev)
W = torch.randn(K, args.cin, args.cout, device=dev) * 0.05
P = plan.num_pairs() if plan is not None else V
subs = [plan.submask.cpu().numpy()] if plan is not None else []
cuts = tuple(int(v) for v in args.split.split(",") if int(v))
if cuts and args.kind == "k3":
    plan = cm.plan_k3_split(ts, cuts[0] if len(cuts) == 1 else cuts)
    subs = [pl.submask.cpu().numpy() for _, _, pl in plan.parts]
if subs:
    slots = sum(bin(int(v)).count("1") for sub in subs for v in sub.reshape(-1)) * 16
    print(f"V={V} pairs={P} row-slots={slots} slot-efficiency={P / slots:.3f} tiles={subs[0].shape[0]}"
          + (f" ({len(subs)} passes)" if len(subs) > 1 else ""))
for _ in range(2):
    out = svnn.conv_forward(feats, W, plan, V)
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True)
e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(args.iters):
    out = svnn.conv_forward(feats, W, plan, V)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / args.iters
fl = 2.0 * P * args.cin * args.cout
gb = P * (4.0 * args.cin + 8) + 4.0 * V * args.cout + 4.0 * K * args.cin * args.cout  # SURVEY.md 8(d) gather-bytes
print(f"{args.kind} level{args.level} {args.cin}->{args.cout}: {ms:.3f} ms/launch, {fl / ms / 1e9:.1f} TFLOP/s algorithmic "
      f"({fl / 1e9:.1f} GFLOP), gather {gb / ms / 1e6:.0f} GB/s algorithmic ({gb / 1e6:.1f} MB)")

"""What in the SHAPE of the room cloud's plan costs matrix-op issue rate?  TIMING ONLY (results are wrong by construction).

A solid cube gives tiles that are all alike (27 offsets x 8 active sub-tiles, 324 steps per 64-row workgroup at 384
channels) and runs at ~93 % of the SIMDs' issue rate while the chip is full; the room cloud reaches ~75 %.  This script
doctors the cube's plan (`submask`, `tile_order`) one property at a time - unequal tile lengths, different offsets per
tile, sparse sub-tiles - and prints the ISSUED matrix-op rate (bits set in submask x 16 rows x Cin x Cout x 2 / time)
as a fraction of the 157.3 TFLOP/s peak.  Edge 64 => 2048 plan tiles => 8192 64-row workgroups = 8 full rounds of 1024.
    SV_CONV_TAIL=0 python tools/tile_shape_experiment.py [edge] [iters]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mrcc_amd  # noqa: E402,F401
from mrcc_amd import MinkowskiEngine as ME  # noqa: E402
from mrcc_amd import nn as svnn  # noqa: E402

edge = int(sys.argv[1]) if len(sys.argv) > 1 else 64
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
CIN = COUT = 384
dev = torch.device("cuda:0")
g = np.arange(edge, dtype=np.float32) + 0.5
xyz = np.stack(np.meshgrid(g, g, g, indexing="ij"), axis=-1).reshape(-1, 3)
coords4 = np.concatenate([np.zeros((len(xyz), 1), np.float32), xyz], axis=1)
x = ME.TensorField(torch.zeros(len(xyz), 3), torch.from_numpy(coords4), device=dev).sparse()
cm = x.coordinate_manager
plan = cm.plan_k3(1)
V = cm.stride_map(1).V
T, K = plan.submask.shape
torch.manual_seed(0)
feats = torch.randn(V, CIN, device=dev)
W = torch.randn(K, CIN, COUT, device=dev) * 0.05
orig_sub = plan.submask.clone()
orig_order = plan.tile_order.clone()
rng = np.random.default_rng(0)
print(f"cube {edge}^3: V={V}, plan tiles={T}, SV_CONV_TAIL={os.environ.get('SV_CONV_TAIL', 'default')}", flush=True)


def popcount(a):
    a = a.astype(np.uint32)
    return sum(((a >> b) & 1) for b in range(8))


ONLY = os.environ.get("TILE_EXPS", "").split()   # e.g. TILE_EXPS="E1 E6"


def run(name, sub_np):
    if ONLY and name.split()[0] not in ONLY:
        return
    sub_np = sub_np.astype(np.int32)
    bits = popcount(sub_np)                              # [T, K] active sub-tiles per (tile, offset)
    cost = bits.sum(axis=1)
    order = np.argsort(-cost, kind="stable").astype(np.int32)   # longest first, as sv_plan_build does
    plan.submask.copy_(torch.from_numpy(sub_np).to(dev))
    plan.tile_order.copy_(torch.from_numpy(order).to(dev))
    for _ in range(2):
        svnn.conv_forward(feats, W, plan, V)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        svnn.conv_forward(feats, W, plan, V)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    issued = float(bits.sum()) * 16 * CIN * COUT * 2
    visited = float((bits > 0).sum()) * 128 * CIN * COUT * 2   # what a skip-free body would issue
    steps = (sub_np != 0).sum(axis=1) * (CIN // 32)
    print(f"{name:66s} {ms:7.3f} ms  issued {issued / ms / 1e9:6.1f} TF = {issued / ms / 1e9 / 157.3:.3f}"
          f"  (steps/tile {steps.min()}..{steps.max()}, mean {steps.mean():.0f}; sub-tile fill {issued / visited:.2f})", flush=True)


full = orig_sub.cpu().numpy().astype(np.int32) & 0xFF
run("E0 cube as is (interior tiles: 27 offsets x 8 sub-tiles)", full)
allon = np.full((T, K), 0xFF, np.int32)
run("E1 every tile: all 27 offsets, all sub-tiles (equal tiles)", allon)

# E2 unequal lengths: tile keeps a random n ~ U[3, 27] offsets, all sub-tiles
sub = np.zeros((T, K), np.int32)
for t in range(T):
    n = rng.integers(3, 28)
    sub[t, rng.choice(K, n, replace=False)] = 0xFF
run("E2 unequal: n ~ U[3,27] random offsets per tile, all sub-tiles", sub)

# E3 equal lengths, different offsets per tile (14 of 27)
sub = np.zeros((T, K), np.int32)
for t in range(T):
    sub[t, rng.choice(K, 14, replace=False)] = 0xFF
run("E3 equal length, 14 random offsets per tile, all sub-tiles", sub)

# E4 equal lengths, same 14 offsets in every tile
sub = np.zeros((T, K), np.int32)
sub[:, rng.choice(K, 14, replace=False)] = 0xFF
run("E4 equal length, the SAME 14 offsets in every tile, all sub-tiles", sub)

# E5 all offsets, sparse sub-tiles (each bit kept with p = 0.75; at least one per offset)
sub = np.zeros((T, K), np.int32)
keep = rng.random((T, K, 8)) < 0.75
keep[:, :, 0] |= ~keep.any(axis=2)
for b in range(8):
    sub |= keep[:, :, b].astype(np.int32) << b
run("E5 all 27 offsets, each sub-tile kept with p = 0.75", sub)

# E6 the room cloud's own distribution of (offsets per tile, sub-tile fill), transplanted onto the cube's rows
pts, rgb, _ = mrcc_amd.synth.gen_room(200000, 2.4, 0)
c4 = np.concatenate([np.zeros((len(pts), 1), np.float32), pts * np.float32(50)], axis=1)
room = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(c4), device=dev).sparse().coordinate_manager.plan_k3(1)
rs = room.submask.cpu().numpy().astype(np.int32) & 0xFF
sub = rs[rng.integers(0, rs.shape[0], T)]
run("E6 room cloud's submask rows, sampled onto the cube", sub)

# E7 as E6 but every visited offset multiplies all 8 sub-tiles
run("E7 as E6, visited offsets with all sub-tiles", np.where(sub != 0, 0xFF, 0))

# E8 as E6 but every tile padded with extra offsets to the same count (27)
run("E8 as E6 plus the missing offsets with ONE sub-tile (equal step counts)", np.where(sub != 0, sub, 1))

plan.submask.copy_(orig_sub)
plan.tile_order.copy_(orig_order)

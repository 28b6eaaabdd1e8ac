"""Timings of the BASELINE.json configurations that are parity cases rather than the bench line:
  Cfg-5: 500k-point / 1 cm frame through RobotNetSegmentation(MinkUNet18D) (the bench workload at 3.5x the voxels),
  Cfg-3: 64 frames in one batched sparse tensor: seg + vote heads (MinkUNet14A-sized, as in the parity test) + 64 Kabsch,
  dense solves: 512 / 65 536 Kabsch problems per launch, FPS 8192 -> 2048.
Synthetic data, random-init weights; every number is steady state after warm-up."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import bench  # noqa: E402
import mrcc_amd  # noqa: E402
from mrcc_amd import MinkowskiEngine as ME  # noqa: E402
from mrcc_amd.app.pipeline import FramePipeline  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, warm=3, reps=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


with torch.no_grad():
    # ---- Cfg-5
    model = bench.build_model(dev)
    frames = []
    for s in range(3):
        pts, rgb, _ = mrcc_amd.synth.gen_room(500_000, 2.4, s)
        c4 = np.concatenate([np.zeros((len(pts), 1), np.float32), pts * np.float32(100)], axis=1)
        frames.append((torch.from_numpy(c4).to(dev), torch.from_numpy(rgb).to(dev)))
    pipe = FramePipeline(dev, levels=4, compute_streams=4)
    V = [0]

    def unet(x, field):
        V[0] = x.F.shape[0]
        return model(x).slice_argmax(field)[0]

    def run5(n=6):
        nxt = pipe.prepare(*frames[0])
        for i in range(n):
            cur = nxt
            pipe.run(cur, unet)
            if i + 1 < n:
                nxt = pipe.prepare(*frames[(i + 1) % 3])
        pipe.drain()

    run5(4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run5(12)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 12
    print(f"Cfg-5  500k pts / 1 cm, {V[0]} voxels, seg U-Net forward (voxelise + maps + U-Net + slice/argmax): "
          f"{dt * 1e3:.1f} ms/frame = {1 / dt:.1f} frames/s")
    del model, frames, pipe
    torch.cuda.empty_cache()

    # ---- Cfg-3
    from mrcc_amd.model.backbone.minkunet import MinkUNet14A
    from mrcc_amd.model.robotnet_segmentation import _classification_head
    from mrcc_amd.utils import transformation as T

    B = 64
    torch.manual_seed(0)
    seg = _classification_head(MinkUNet14A, lambda: 3, "SegSmall")(3, num_classes=3).to(dev).eval()
    vote = _classification_head(MinkUNet14A, lambda: 2, "VoteSmall")(3, num_classes=2).to(dev).eval()
    crops = [mrcc_amd.synth.gen_ee_crop(s, n=1500) for s in range(B)]
    coords = ME.utils.batched_coordinates([torch.from_numpy(c[0] * np.float32(100)) for c in crops],
                                          dtype=torch.float32).to(dev)
    feats = torch.from_numpy(np.concatenate([c[1] for c in crops])).to(dev)
    ref = np.repeat(mrcc_amd.synth.REFERENCE_KEY_POINTS[None], B, axis=0)
    tgt = np.stack([c[3] for c in crops])

    def cfg3():
        field = ME.TensorField(feats, coords, device=dev)
        x = field.sparse()
        labels, _ = seg(x).slice_argmax(field)
        v = vote(x).slice(field).F
        T.get_rigid_transform_3D_batched(ref, tgt, device=dev)
        return labels, v

    dt = timed(cfg3)
    print(f"Cfg-3  batch of 64 frames x 1500 pts (1 cm), seg + vote (MinkUNet14A heads) + 64 Kabsch: {dt * 1e3:.1f} ms/batch "
          f"= {B / dt:.0f} frames/s")

    # ---- dense solves
    for n in (512, 65536):
        rng = np.random.default_rng(0)
        r = rng.normal(size=(n, 6, 3))
        t = r + rng.normal(size=(n, 1, 3))
        rd, td = torch.from_numpy(r).to(dev), torch.from_numpy(t).to(dev)
        from mrcc_amd import _lib
        from mrcc_amd._lib import c_int, ptr, stream_ptr
        R = torch.empty(n, 9, dtype=torch.float64, device=dev)
        tt = torch.empty(n, 3, dtype=torch.float64, device=dev)
        q = torch.empty(n, 4, dtype=torch.float64, device=dev)
        f = lambda: _lib.call("sv_kabsch_batched", ptr(rd), ptr(td), None, c_int(6), c_int(n), ptr(R), ptr(tt), ptr(q), stream_ptr())
        dt = timed(f, reps=50)
        print(f"Kabsch {n} problems (6 points, f64 Jacobi SVD, one wave per problem): {dt * 1e6:.1f} us/launch = {n / dt / 1e6:.2f} M problems/s")
    from mrcc_amd.model import pointnet2_utils as P2
    xyz = torch.rand(8, 8192, 3, device=dev)
    st = torch.zeros(8, dtype=torch.long, device=dev)
    dt = timed(lambda: P2.farthest_point_sample(xyz, 2048, start=st), reps=10)
    print(f"FPS 8 clouds x 8192 -> 2048 samples: {dt * 1e3:.2f} ms")

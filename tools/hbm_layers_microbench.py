"""The two gather-bound layers of the bench's `hbm_bound_layers` block, alone, under experiment switches:
    conv0 3->32 k27 at level 0 (88k voxels)   SV_CONV_FIRST_VALU=1 = the thread-per-voxel kernel
    32->32 k27 at level 1 (26k) and level 0 (88k)   SV_THIN_VARIANT=n
and a bit-exact comparison of every variant with the default path.  python tools/hbm_layers_microbench.py"""
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    import mrcc_amd
    from mrcc_amd import MinkowskiEngine as ME
    from mrcc_amd import nn as svnn

    dev = torch.device("cuda:0")
    mrcc_amd._lib.call("sv_conv_set_dispatch", __import__("ctypes").c_double(1.0), __import__("ctypes").c_double(-1.0))  # one layer alone
    pts, rgb, _ = mrcc_amd.synth.gen_room(200_000, 2.4, 0)
    coords4 = np.concatenate([np.zeros((len(pts), 1), np.float32), pts * np.float32(50)], axis=1)
    x = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(coords4), device=dev).sparse()
    cm = x.coordinate_manager
    torch.manual_seed(0)
    res = {}
    for name, level, cin in (("conv0", 0, 3), ("thin_l1", 1, 32), ("thin_l0", 0, 32)):
        plan = cm.plan_k3(1 << level)
        V = cm.stride_map(1 << level).V
        feats = x.F if cin == 3 else torch.randn(V, cin, device=dev)
        W = torch.randn(27, cin, 32, device=dev) * 0.1
        sc, sh = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev)
        for _ in range(3):
            out = svnn.conv_forward(feats, W, plan, V, sc, sh, None, 1)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(100):
            out = svnn.conv_forward(feats, W, plan, V, sc, sh, None, 1)
        e.record()
        torch.cuda.synchronize()
        us = s.elapsed_time(e) * 10
        P = plan.num_pairs()
        gb = (P * (4.0 * cin + 8) + 4.0 * V * 32 + 4.0 * 27 * cin * 32) / 1e9
        res[name] = out.cpu().numpy()
        print(f"  {name:8s} V={V:6d}: {us:7.2f} us = {gb / us * 1e6:7.1f} GB/s = {gb / us * 1e6 / 8000:.3f} of the HBM peak "
              f"[{mrcc_amd._lib.conv_last_instance()[0]}]")
    np.savez(sys.argv[2], **res)
    sys.exit(0)

import tempfile

tmp = tempfile.mkdtemp()
base = None
for label, env in (("default", {}), ("conv0 VALU kernel", {"SV_CONV_FIRST_VALU": "1"}),
                   ("thin variant 0 (round 3: weights through L1, one wave per sub-tile, D=4)", {"SV_THIN_VARIANT": "0"}),
                   ("thin variant 10 (LDS weights, D=4, 16 waves)", {"SV_THIN_VARIANT": "10"}),
                   ("thin variant 20 (LDS weights, D=2, 16 waves, next table prefetched)", {"SV_THIN_VARIANT": "20"})):
    if len(sys.argv) > 1 and label != "default" and env.get("SV_THIN_VARIANT") not in sys.argv[1:]:
        continue  # `python tools/hbm_layers_microbench.py 5 6`: the default and those thin-kernel variants only
    print(label, flush=True)
    path = os.path.join(tmp, label.split()[0] + str(len(env)) + "".join(env.values()) + ".npz")
    r = subprocess.run([sys.executable, __file__, "child", path], env=dict(os.environ, **env), capture_output=True, text=True)
    sys.stdout.write(r.stdout)
    if r.returncode:
        sys.stdout.write(r.stderr[-2000:])
        continue
    got = np.load(path)
    if base is None:
        base = {k: got[k] for k in got.files}
    else:
        print("  bit-exact vs default:", {k: bool(np.array_equal(base[k], got[k])) for k in got.files}, flush=True)

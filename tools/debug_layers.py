"""Per-layer isolation on the GPU box: run each fused layer of MinkUNet18D-seg on the GPU and feed the SAME input to
the oracle; report the first layers whose outputs differ bitwise."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import mrcc_amd  # noqa: E402
import sv_oracle as O  # noqa: E402
from mrcc_amd import MinkowskiEngine as ME  # noqa: E402
from mrcc_amd.model.robotnet_segmentation import RobotNetSegmentation  # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(1)
model = RobotNetSegmentation(in_channels=3, num_classes=3)
g = torch.Generator().manual_seed(2)
for m in model.modules():
    if isinstance(m, torch.nn.BatchNorm1d):
        with torch.no_grad():
            m.weight.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
            m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.1)
            m.running_var.copy_(torch.rand(m.num_features, generator=g) * 0.5 + 0.75)
model = model.to(dev).eval()
sd = {k: v.cpu() for k, v in model.state_dict().items()}
pts, rgb, lab = mrcc_amd.synth.gen_room(6000, 0.5, 3)
coords4 = np.concatenate([np.zeros((len(pts), 1), np.float32), pts * np.float32(50)], axis=1)
field = ME.TensorField(torch.from_numpy(rgb), torch.from_numpy(coords4), device=dev)
x = field.sparse()
vox = O.voxelize(coords4)
frame = O.Frame(vox["coords"])
for ts in (1, 2, 4, 8):
    frame.down(ts)


def report(name, got, want):
    got = got.cpu().numpy()
    eq = np.array_equal(got, want)
    print(f"{name:28s} {'OK ' if eq else 'DIFF'} shape={got.shape} maxdiff={np.abs(got - want).max():.3e}")
    return eq


def check_conv(name, conv, bn, xin, kind, ts, act, residual=None):
    out = conv.forward_fused(xin, bn=bn, residual=residual, act=act)
    xi = xin.F.cpu().numpy()
    prefix = name
    W = O._kernel3(sd, prefix + ".kernel")
    if kind == "k3":
        nbr, V = frame.k3(ts), len(frame.maps[ts])
    elif kind == "down":
        nbr, V = frame.kdown(ts), len(frame.maps[2 * ts])
    elif kind == "up":
        nbr, V = frame.kup(ts), len(frame.maps[ts // 2])
    else:
        nbr, V = None, xi.shape[0]
    if bn is not None:
        bn_name = [k for k, m in model.named_modules() if m is bn][0]
        s, b = O._bn(sd, bn_name)
        gs, gb = bn.folded()
        if not (np.array_equal(gs.cpu().numpy(), s) and np.array_equal(gb.cpu().numpy(), b)):
            print(f"   fold_bn differs for {bn_name}: scale {np.abs(gs.cpu().numpy() - s).max():.3e} "
                  f"shift {np.abs(gb.cpu().numpy() - b).max():.3e}")
    else:
        s = None
        b = sd[prefix + ".bias"].numpy().reshape(-1) if prefix + ".bias" in sd else None
    res = residual.F.cpu().numpy() if residual is not None else None
    want = O.conv(xi, W, nbr, V, s, b, res, act)
    report(name, out.F, want)
    return out


def check_block(name, block, xin, ts):
    out = xin
    for i, blk in enumerate(block):
        p = f"{name}.{i}"
        o1 = check_conv(p + ".conv1", blk.conv1, blk.norm1, out, "k3", ts, 1)
        if blk.downsample is not None:
            res = check_conv(p + ".downsample.0", blk.downsample[0], blk.downsample[1], out, "k1", ts, 0)
        else:
            res = out
        out = check_conv(p + ".conv2", blk.conv2, blk.norm2, o1, "k3", ts, 1, residual=res)
    return out


with torch.no_grad():
    out = check_conv("conv0p1s1", model.conv0p1s1, model.bn0, x, "k3", 1, 1)
    skips = [out]
    for i in range(1, 5):
        ts = 2 ** (i - 1)
        out = check_conv(f"conv{i}p{ts}s2", getattr(model, f"conv{i}p{ts}s2"), getattr(model, f"bn{i}"), out, "down",
                         ts, 1)
        out = check_block(f"block{i}", getattr(model, f"block{i}"), out, 2 * ts)
        skips.append(out)
    out = skips.pop()
    for j in range(4, 8):
        ts = 2 ** (8 - j)
        out = check_conv(f"convtr{j}p{ts}s2", getattr(model, f"convtr{j}p{ts}s2"), getattr(model, f"bntr{j}"), out,
                         "up", ts, 1)
        out = ME.cat(out, skips.pop())
        out = check_block(f"block{j + 1}", getattr(model, f"block{j + 1}"), out, ts // 2)
    out = check_conv("final", model.final, None, out, "k1", 1, 2)

"""PointNet++ sampling / grouping with the reference's names (model/pointnet2_utils.py) on libsvhip.

farthest_point_sample(xyz [B,N,3], npoint) -> int64 [B,npoint]     reference :65-86 (python loop of npoint launches)
query_ball_point(radius, nsample, xyz, new_xyz) -> int64 [B,S,nsample]   reference :89-109 ([B,S,N] matrix + sort)
index_points, square_distance: thin torch helpers with the reference's semantics (:21-62).
"""
from ctypes import c_double, c_int

import torch

from .._lib import call, ptr, require_cuda, stream_ptr


def farthest_point_sample(xyz, npoint, start=None):
    require_cuda(xyz, "xyz")
    B, N, C = xyz.shape
    x = xyz[..., :3].to(torch.float32).contiguous()
    if start is None:  # the reference: torch.randint(0, N, (B,))
        start = torch.randint(0, N, (B,), dtype=torch.long, device=xyz.device)
    start = start.to(device=xyz.device, dtype=torch.int64).contiguous()
    out = torch.empty((B, npoint), dtype=torch.int64, device=xyz.device)
    call("sv_fps", ptr(x), c_int(B), c_int(N), c_int(npoint), ptr(start), ptr(out), stream_ptr())
    return out


def query_ball_point(radius, nsample, xyz, new_xyz):
    require_cuda(xyz, "xyz")
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    x = xyz.to(torch.float32).contiguous()
    q = new_xyz.to(torch.float32).contiguous()
    out = torch.empty((B, S, nsample), dtype=torch.int64, device=xyz.device)
    call("sv_ball_query", ptr(x), ptr(q), c_int(B), c_int(N), c_int(S), c_double(float(radius)), c_int(nsample),
         ptr(out), stream_ptr())
    return out


def index_points(points, idx):
    B = points.shape[0]
    view = [B] + [1] * (idx.dim() - 1)
    batch = torch.arange(B, dtype=torch.long, device=points.device).view(view).expand_as(idx)
    return points[batch, idx, :]


def square_distance(src, dst):
    dist = -2 * torch.matmul(src, dst.permute(0, 2, 1))
    dist += torch.sum(src ** 2, -1).unsqueeze(-1)
    dist += torch.sum(dst ** 2, -1).unsqueeze(1)
    return dist


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False):
    """reference :112-140."""
    B, N, C = xyz.shape
    fps_idx = farthest_point_sample(xyz, npoint)
    new_xyz = index_points(xyz, fps_idx)
    idx = query_ball_point(radius, nsample, xyz, new_xyz)
    grouped_xyz = index_points(xyz, idx)
    grouped_xyz_norm = grouped_xyz - new_xyz.view(B, npoint, 1, C)
    new_points = grouped_xyz_norm if points is None else torch.cat([grouped_xyz_norm, index_points(points, idx)], -1)
    if returnfps:
        return new_xyz, new_points, grouped_xyz, fps_idx
    return new_xyz, new_points


# ----------------------------------------------------------------------------------------------------------------------
# set abstraction / feature propagation modules (reference :163-317).  Parameter names are the reference's
# (mlp_convs.i = nn.Conv2d / nn.Conv1d 1x1, mlp_bns.i = BatchNorm) so its checkpoints load by key; in eval mode the
# shared MLPs run as dense rows through the fp32-MFMA kernel with the BatchNorm folded into the epilogue.
# ----------------------------------------------------------------------------------------------------------------------
import numpy as np  # noqa: E402
import torch.nn as nn  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from .. import nn as svnn  # noqa: E402
from .._lib import SV_ACT_RELU  # noqa: E402


def _fold_conv_bn(conv, bn):
    """1x1 conv (+bias) followed by BatchNorm(eval) -> W[1,Cin,Cout], scale, shift with the conv bias folded in."""
    w = conv.weight.detach().reshape(conv.out_channels, conv.in_channels).t().contiguous().unsqueeze(0)
    g = bn.weight.detach().float().cpu().numpy()
    b = bn.bias.detach().float().cpu().numpy()
    mean = bn.running_mean.detach().float().cpu().numpy()
    var = bn.running_var.detach().float().cpu().numpy()
    scale = (g / np.sqrt(var + np.float32(bn.eps))).astype(np.float32)
    cb = conv.bias.detach().float().cpu().numpy() if conv.bias is not None else np.zeros_like(mean)
    shift = (b + (cb - mean) * scale).astype(np.float32)
    dev = conv.weight.device
    return w, torch.from_numpy(scale).to(dev), torch.from_numpy(shift).to(dev)


def _mlp_rows(rows, convs, bns):
    """rows [R, Cin] -> relu(bn(conv(.))) stack, one fused launch per layer."""
    for conv, bn in zip(convs, bns):
        w, scale, shift = _fold_conv_bn(conv, bn)
        rows = svnn.conv_forward(rows, w, None, rows.shape[0], scale, shift, None, SV_ACT_RELU)
    return rows


def sample_and_group_all(xyz, points):
    B, N, C = xyz.shape
    new_xyz = torch.zeros(B, 1, C, device=xyz.device)
    grouped = xyz.view(B, 1, N, C)
    new_points = grouped if points is None else torch.cat([grouped, points.view(B, 1, N, -1)], dim=-1)
    return new_xyz, new_points


class PointNetSetAbstraction(nn.Module):
    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all):
        super().__init__()
        self.npoint, self.radius, self.nsample, self.group_all = npoint, radius, nsample, group_all
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for out in mlp:
            self.mlp_convs.append(nn.Conv2d(last, out, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out))
            last = out

    def forward(self, xyz, points):
        """xyz [B,3,N], points [B,D,N] -> new_xyz [B,3,S], new_points [B,D',S]."""
        xyz = xyz.permute(0, 2, 1)
        if points is not None:
            points = points.permute(0, 2, 1)
        if self.group_all:
            new_xyz, new_points = sample_and_group_all(xyz, points)
        else:
            new_xyz, new_points = sample_and_group(self.npoint, self.radius, self.nsample, xyz, points)
        B, S, Kn, C = new_points.shape  # [B, npoint, nsample, C+D]
        if self.training:
            t = new_points.permute(0, 3, 2, 1)
            for conv, bn in zip(self.mlp_convs, self.mlp_bns):
                t = F.relu(bn(conv(t)))
            pooled = torch.max(t, 2)[0]
        else:
            rows = _mlp_rows(new_points.reshape(B * S * Kn, C).contiguous(), self.mlp_convs, self.mlp_bns)
            pooled = rows.view(B, S, Kn, -1).max(dim=2)[0].permute(0, 2, 1)  # [B, D', S]
        return new_xyz.permute(0, 2, 1), pooled


def three_nn_interpolate(xyz1, xyz2, points2):
    """[B,N,3], [B,S,3], [B,S,C] -> [B,N,C]: 3-NN inverse-distance interpolation on libsvhip (sv_three_nn_interpolate),
    replacing the reference's full [B,N,S] distance matrix + sort (model/pointnet2_utils.py:298-305)."""
    from ctypes import c_int

    from .._lib import call, ptr, stream_ptr

    B, N, _ = xyz1.shape
    S, C = points2.shape[1], points2.shape[2]
    x1, x2, p2 = (t.to(torch.float32).contiguous() for t in (xyz1, xyz2, points2))
    out = torch.empty((B, N, C), dtype=torch.float32, device=x1.device)
    call("sv_three_nn_interpolate", ptr(x1), ptr(x2), ptr(p2), c_int(B), c_int(N), c_int(S), c_int(C), ptr(out),
         stream_ptr())
    return out


class PointNetFeaturePropagation(nn.Module):
    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs = nn.ModuleList()
        self.mlp_bns = nn.ModuleList()
        last = in_channel
        for out in mlp:
            self.mlp_convs.append(nn.Conv1d(last, out, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out))
            last = out

    def forward(self, xyz1, xyz2, points1, points2):
        """3-NN inverse-distance interpolation of points2 (at xyz2) onto xyz1, concat points1, shared MLP."""
        xyz1 = xyz1.permute(0, 2, 1)
        xyz2 = xyz2.permute(0, 2, 1)
        points2 = points2.permute(0, 2, 1)
        B, N, _ = xyz1.shape
        S = xyz2.shape[1]
        if S == 1:
            interpolated = points2.repeat(1, N, 1)
        elif not self.training and xyz1.is_cuda and S >= 3:
            interpolated = three_nn_interpolate(xyz1, xyz2, points2)
        else:
            dists, idx = square_distance(xyz1, xyz2).topk(3, dim=-1, largest=False, sorted=True)
            recip = 1.0 / (dists + 1e-8)
            weight = recip / recip.sum(dim=2, keepdim=True)
            interpolated = torch.sum(index_points(points2, idx) * weight.view(B, N, 3, 1), dim=2)
        new_points = interpolated if points1 is None else torch.cat([points1.permute(0, 2, 1), interpolated], dim=-1)
        if self.training:
            t = new_points.permute(0, 2, 1)
            for conv, bn in zip(self.mlp_convs, self.mlp_bns):
                t = F.relu(bn(conv(t)))
            return t
        rows = _mlp_rows(new_points.reshape(B * N, -1).contiguous(), self.mlp_convs, self.mlp_bns)
        return rows.view(B, N, -1).permute(0, 2, 1)

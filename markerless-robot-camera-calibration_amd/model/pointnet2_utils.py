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

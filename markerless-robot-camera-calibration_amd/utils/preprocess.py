"""Pre-voxelisation transforms of the reference (utils/preprocess.py:8-56).

Same names, arguments and return values.  numpy inputs (what app/inference_engine.py hands over, DTOs hold host
arrays) are transformed on the host exactly as the reference does; torch tensors that already live on the GPU are
transformed there by libsvhip (`sv_col_stats` + `sv_center_scale`: a two-stage deterministic column reduction and one
elementwise pass), so a frame that arrives in HBM is never copied back for its preprocessing.  The data-dependent branches
of normalize_colors need the column extrema on the host: one 32-byte read-back.
"""
from ctypes import c_int, c_int64, c_size_t

import numpy as np
import torch


def _stats(x, sub=None, want_norm=False):
    """(min[C], max[C], sum[C] float64, max row norm or None) of a float32 CUDA tensor [N, C], C <= 4."""
    from .. import _lib
    from .._lib import call, ptr, stream_ptr

    N, C = x.shape
    dev = x.device
    ws_bytes = _lib.load().sv_col_stats_workspace_bytes(c_int64(N))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    mn = torch.empty(C, dtype=torch.float32, device=dev)
    mx = torch.empty(C, dtype=torch.float32, device=dev)
    sm = torch.empty(C, dtype=torch.float64, device=dev)
    nm = torch.empty(1, dtype=torch.float32, device=dev) if want_norm else None
    call("sv_col_stats", ptr(x), c_int64(x.stride(0)), c_int64(N), c_int(C), ptr(sub), ptr(ws), c_size_t(ws_bytes), ptr(mn),
         ptr(mx), ptr(sm), ptr(nm), stream_ptr())
    return mn, mx, sm, nm


def _apply(x, sub=None, div=None, add=None):
    from .._lib import call, ptr, stream_ptr

    N, C = x.shape
    out = torch.empty((N, C), dtype=torch.float32, device=x.device)
    call("sv_center_scale", ptr(x), c_int64(x.stride(0)), c_int64(N), c_int(C), ptr(sub), ptr(div), ptr(add), ptr(out),
         c_int64(C), stream_ptr())
    return out


def _device_rows(t):
    if t.dim() != 2 or t.shape[1] > 4 or t.shape[0] < 1:
        raise ValueError("device preprocessing expects [N >= 1, C <= 4] rows")
    t = t.to(torch.float32)
    return t if t.stride(1) == 1 else t.contiguous()


def center_at_origin(points):
    if torch.is_tensor(points) and points.is_cuda:
        x = _device_rows(points)
        mn, mx, _, _ = _stats(x)
        origin_offset = (mx + mn) / 2
        return _apply(x, sub=origin_offset), origin_offset
    origin_offset = (points.max(axis=0) + points.min(axis=0)) / 2
    return points - origin_offset, origin_offset


def base_at_origin(points):
    if torch.is_tensor(points) and points.is_cuda:
        x = _device_rows(points)
        mn, _, _, _ = _stats(x)
        return _apply(x, sub=mn), mn
    origin_base_offset = points.min(axis=0)
    return points - origin_base_offset, origin_base_offset


def _minmax01(col):
    lo, hi = col.min(), col.max()
    rng = hi - lo
    return (col - lo) / (rng if rng != 0 else 1.0)


def normalize_colors(rgb_input, is_color_in_range_0_255=False):
    """/255 when the data looks like 0..255, per-channel min-max when negative values are present, then shift
    [0,1] -> [-0.5,0.5] (the data-dependent branches of utils/preprocess.py:20-37)."""
    if torch.is_tensor(rgb_input) and rgb_input.is_cuda:
        x = _device_rows(rgb_input)
        dev = x.device
        mn, mx, _, _ = _stats(x)
        lo, hi = mn.cpu().numpy(), mx.cpu().numpy()  # the branch decisions below are host decisions in the reference too
        if is_color_in_range_0_255 or hi.max() > 2:
            x = _apply(x, div=torch.full((x.shape[1],), 255.0, device=dev))
            lo, hi = lo / np.float32(255.0), hi / np.float32(255.0)
        if lo.min() < 0:
            # utils/preprocess.py:28-30 rescales columns 0..2 only; a fourth column passes through
            rng = hi - lo
            rng[rng == 0] = 1.0
            sub, div = lo.copy(), rng.astype(np.float32)
            sub[3:], div[3:] = 0.0, 1.0
            x = _apply(x, sub=torch.from_numpy(sub).to(dev), div=torch.from_numpy(div).to(dev))
            lo[:3], hi[:3] = 0.0, 1.0
        if lo.min() > (-1e-6) and hi.max() < (1 + 1e-6):
            x = _apply(x, sub=torch.full((x.shape[1],), 0.5, device=dev))
        return x
    rgb = np.array(rgb_input, copy=True)
    if is_color_in_range_0_255 or rgb.max() > 2:
        rgb /= 255.0
    if rgb.min() < 0:
        for c in range(3):
            rgb[:, c] = _minmax01(rgb[:, c])
    if rgb.min() > (-1e-6) and rgb.max() < (1 + 1e-6):
        rgb -= 0.5
    return rgb


def normalize_points(pc, ver=2):
    if ver == 1 or not 1 < len(pc.shape) < 4:
        return pc
    if torch.is_tensor(pc) and pc.is_cuda and pc.dim() == 2:
        x = _device_rows(pc)
        _, _, sm, _ = _stats(x)
        mean = (sm / x.shape[0]).to(torch.float32)
        _, _, _, nmax = _stats(x, sub=mean, want_norm=True)
        return _apply(x, sub=mean, div=nmax.expand(x.shape[1]).contiguous())
    if len(pc.shape) == 2:
        pc = np.array(pc, copy=True)
        pc = pc - pc.mean(0)
        pc /= np.max(np.linalg.norm(pc, axis=-1))
        return pc
    if torch.is_tensor(pc):
        pc = pc - pc.mean(dim=1).view(-1, 1, 3)
        return pc / torch.max(torch.linalg.norm(pc, dim=-1), dim=-1).values.view(-1, 1, 1)
    pc = np.asarray(pc)
    pc = pc - pc.mean(1).reshape(-1, 1, 3)
    return pc / np.max(np.linalg.norm(pc, axis=-1), axis=-1).reshape(-1, 1, 1)

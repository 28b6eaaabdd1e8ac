"""ICP refinement with the reference's interface (utils/icp.py:13-83) on libsvhip.

    match = get_point2point_matcher(cad_points)       # the reference samples 8192 points from app/hand_files/*.obj
    pose = match(ee_points, pose_initial)             # (x, y, z, qw, qx, qy, qz) -> refined pose

Same registration as the reference's open3d call: point-to-point, max correspondence distance 0.1 m, at most 30
updates, relative fitness / rmse tolerance 1e-6, source = CAD points, target = end-effector crop, initial transform
from the predicted pose.  (The reference also estimates normals on the crop, which point-to-point ICP never reads.)
"""
from ctypes import c_double, c_int, c_int64, c_size_t

import numpy as np
import torch

from .. import _lib
from .._lib import call, ptr, stream_ptr
from .transformation import get_pose_from_matrix, get_transformation_matrix


def icp_point2point(src, tgt, init_T=None, max_distance=0.1, max_iterations=30, rel_fitness=1e-6, rel_rmse=1e-6,
                    device="cuda"):
    """src [S,3], tgt [T,3] (numpy or tensors) -> (T 4x4 float64 numpy, fitness, inlier rmse, updates)."""
    dev = torch.device(device)
    s = torch.as_tensor(np.ascontiguousarray(src, dtype=np.float32) if not torch.is_tensor(src) else src).to(
        device=dev, dtype=torch.float32).contiguous()
    t = torch.as_tensor(np.ascontiguousarray(tgt, dtype=np.float32) if not torch.is_tensor(tgt) else tgt).to(
        device=dev, dtype=torch.float32).contiguous()
    S, T = s.shape[0], t.shape[0]
    init = None if init_T is None else torch.as_tensor(np.ascontiguousarray(init_T, dtype=np.float64)).to(dev)
    ws_bytes = _lib.load().sv_icp_workspace_bytes(c_int64(S))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    out_T = torch.empty(16, dtype=torch.float64, device=dev)
    stats = torch.empty(3, dtype=torch.float64, device=dev)
    call("sv_icp_point2point", ptr(s), c_int64(S), ptr(t), c_int64(T), ptr(init), c_double(max_distance),
         c_int(max_iterations), c_double(rel_fitness), c_double(rel_rmse), ptr(ws), c_size_t(ws_bytes), ptr(out_T),
         ptr(stats), stream_ptr())
    st = stats.cpu().numpy()
    return out_T.cpu().numpy().reshape(4, 4), float(st[0]), float(st[1]), int(st[2])


def get_point2point_matcher(cad_points, icp_threshold=0.1, max_iterations=30, device="cuda"):
    cad = torch.as_tensor(np.ascontiguousarray(cad_points, dtype=np.float32)).to(device)

    def match(ee_points, pose_initial):
        if ee_points is None or pose_initial is None:
            return pose_initial
        T0 = get_transformation_matrix(np.asarray(pose_initial, dtype=np.float64), switch_w=False)
        T, _, _, _ = icp_point2point(cad, ee_points, T0, icp_threshold, max_iterations, device=device)
        return get_pose_from_matrix(T)

    return match

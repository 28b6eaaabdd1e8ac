"""Pose averaging with the reference's names (utils/calibration.py:69-139), eigen-solve on libsvhip.

compute_quaternions_weighted_average: principal eigenvector of sum w_i q_i q_i^T / sum w_i — the reference uses
np.linalg.eig and leaves the sign arbitrary; here the sign makes the largest-magnitude component positive.
"""
from ctypes import c_int

import numpy as np
import torch

from .._lib import call, ptr, stream_ptr


def compute_quaternions_weighted_average_batched(Q, w=None, M=None, device=None):
    """Q [B, Mmax, 4] (w,x,y,z); w [B, Mmax]; M int[B].  Returns [B,4] float64."""
    dev = torch.device("cuda" if device is None else device)
    Qt = torch.as_tensor(np.ascontiguousarray(Q, dtype=np.float64)).to(dev)
    B, Mmax, _ = Qt.shape
    wt = None if w is None else torch.as_tensor(np.ascontiguousarray(w, dtype=np.float64)).to(dev)
    Mt = None if M is None else torch.as_tensor(np.asarray(M, dtype=np.int32)).to(dev)
    out = torch.empty((B, 4), dtype=torch.float64, device=dev)
    call("sv_quat_avg_batched", ptr(Qt), ptr(wt), ptr(Mt), c_int(Mmax), c_int(B), ptr(out), stream_ptr())
    return out.cpu().numpy()


def compute_quaternions_weighted_average(Q, w):
    Q = np.asarray(Q, dtype=np.float64)
    w = np.asarray(w, dtype=np.float64)
    return compute_quaternions_weighted_average_batched(Q[None], w[None])[0]


def compute_quaternions_average(Q):
    return compute_quaternions_weighted_average(Q, np.ones(np.asarray(Q).shape[0]))


def compute_translations_average(t, weights=None):
    t = np.asarray(t, dtype=np.float64)
    if weights is None:
        weights = np.ones(len(t))
    weights = np.asarray(weights, dtype=np.float64)
    return np.sum(t * weights.reshape(-1, 1), axis=0) / np.sum(weights)


def compute_poses_average(poses, weights=None):
    """poses Nx7 (x, y, z, qw, qx, qy, qz) -> 7-vector (utils/calibration.py:117-139)."""
    if poses is None or len(poses) == 0:
        return poses
    poses = np.asarray(poses, dtype=np.float64)
    if len(poses.shape) != 2:
        poses = np.array(poses.reshape(-1, 7), copy=True)
    if len(poses) == 1:
        return poses[0]
    if weights is None or len(weights) != len(poses):
        weights = np.ones(len(poses))
    out = np.zeros(7)
    out[:3] = compute_translations_average(poses[:, :3], weights=weights)
    out[3:] = compute_quaternions_weighted_average(poses[:, 3:], weights)
    return out


def remove_pose_outliers(poses):
    # the reference computes the outlier mask and then returns the input unchanged (utils/calibration.py:55-61)
    return poses

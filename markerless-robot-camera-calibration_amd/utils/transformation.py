"""Rigid-transform helpers with the reference's names (utils/transformation.py), the solves running on libsvhip.

  get_rigid_transform_3D(reference, target) -> (R, t)        reference: utils/transformation.py:178-222
  get_q_from_matrix(R) -> (w, x, y, z)                        reference: :80-84 (scipy Rotation.from_matrix)
  get_rigid_transform_3D_batched(...)                         B problems in one launch (one wavefront each)
  get_quaternion_rotation_matrix / get_transformation_matrix(_inverse) / get_pose_from_matrix / get_base2cam_pose /
  transform_pose2pose                                          reference: :16-77, :87-93, :225-266 (small host algebra)
Arguments, return types (numpy float64) and error behaviour follow the reference.
"""
from ctypes import c_int

import numpy as np
import torch

from .._lib import call, ptr, stream_ptr


def _dev(device):
    return torch.device("cuda" if device is None else device)


def get_rigid_transform_3D_batched(reference, target, K=None, device=None, as_tensors=False):
    """reference, target: [B, Kmax, 3] (numpy, or float64 tensors already on the device); K: int[B] points used per problem
    (default Kmax).  Returns (R [B,3,3], t [B,3], q_wxyz [B,4]) as float64 numpy arrays - or, as_tensors=True, as CUDA
    tensors on the current stream with nothing downloaded and no host synchronisation (a caller inside a frame pipeline
    collects them later)."""
    dev = _dev(device)
    if torch.is_tensor(reference) and torch.is_tensor(target) and reference.is_cuda and target.is_cuda:
        ref = reference.to(dtype=torch.float64).contiguous()
        tgt = target.to(dtype=torch.float64).contiguous()
    else:
        ref = torch.as_tensor(np.ascontiguousarray(reference, dtype=np.float64)).to(dev)
        tgt = torch.as_tensor(np.ascontiguousarray(target, dtype=np.float64)).to(dev)
    if ref.shape != tgt.shape or ref.dim() != 3 or ref.shape[2] != 3:
        raise Exception(f"reference/target must both be Bx{'K'}x3, got {tuple(ref.shape)} and {tuple(tgt.shape)}")
    B, Kmax, _ = ref.shape
    Kt = None
    if K is not None:
        K = np.asarray(K, dtype=np.int32)
        if (K < 1).any() or (K > Kmax).any():
            raise Exception("K out of range")
        Kt = torch.as_tensor(K).to(dev)
    R = torch.empty((B, 3, 3), dtype=torch.float64, device=dev)
    t = torch.empty((B, 3), dtype=torch.float64, device=dev)
    q = torch.empty((B, 4), dtype=torch.float64, device=dev)
    call("sv_kabsch_batched", ptr(ref), ptr(tgt), ptr(Kt), c_int(Kmax), c_int(B), ptr(R), ptr(t), ptr(q),
         stream_ptr())
    if as_tensors:
        return R, t, q
    return R.cpu().numpy(), t.cpu().numpy(), q.cpu().numpy()


def get_rigid_transform_3D(reference, target):
    A = np.asarray(reference)
    B = np.asarray(target)
    assert A.shape == B.shape
    if A.ndim != 2 or A.shape[1] != 3:
        raise Exception(f"matrix A is not 3xN, it is {A.shape[1] if A.ndim == 2 else '?'}x{A.shape[0]}")
    R, t, _ = get_rigid_transform_3D_batched(A[None], B[None])
    return R[0], t[0].reshape(-1)


def get_q_from_matrix(rot_mat):
    """(w, x, y, z) of a rotation matrix, scipy's branch rule; runs the same device code as the Kabsch epilogue by
    solving the trivial problem identity-points -> rows of R."""
    R = np.array(rot_mat, dtype=np.float64, copy=True)
    eye = np.concatenate([np.eye(3), np.zeros((1, 3))])  # 4 points: e1, e2, e3, origin
    tgt = eye @ R.T
    _, _, q = get_rigid_transform_3D_batched(eye[None], tgt[None])
    return q[0]


def switch_w(pose):
    """x, y, z, qx, qy, qz, qw -> x, y, z, qw, qx, qy, qz (utils/transformation.py:7-13)."""
    return np.insert(np.array(pose[:-1], copy=True), len(pose) - 4, pose[-1])


def get_quaternion_rotation_matrix(Q_init, switch_w=True):
    Q = np.insert(Q_init[:3], 0, Q_init[-1]) if switch_w else Q_init
    q0, q1, q2, q3 = Q[0], Q[1], Q[2], Q[3]
    return np.array([
        [2 * (q0 * q0 + q1 * q1) - 1, 2 * (q1 * q2 - q0 * q3), 2 * (q1 * q3 + q0 * q2)],
        [2 * (q1 * q2 + q0 * q3), 2 * (q0 * q0 + q2 * q2) - 1, 2 * (q2 * q3 - q0 * q1)],
        [2 * (q1 * q3 - q0 * q2), 2 * (q2 * q3 + q0 * q1), 2 * (q0 * q0 + q3 * q3) - 1],
    ])


def get_quaternion_rotation_matrix_torch(quaternions):
    """(..., 4) quaternions, real part first and not necessarily unit -> (..., 3, 3) (utils/transformation.py:104-131)."""
    r, i, j, k = torch.unbind(quaternions, -1)
    two_s = 2.0 / (quaternions * quaternions).sum(-1)
    o = torch.stack((1 - two_s * (j * j + k * k), two_s * (i * j - k * r), two_s * (i * k + j * r),
                     two_s * (i * j + k * r), 1 - two_s * (i * i + k * k), two_s * (j * k - i * r),
                     two_s * (i * k - j * r), two_s * (j * k + i * r), 1 - two_s * (i * i + j * j)), -1)
    return o.reshape(quaternions.shape[:-1] + (3, 3))


def get_transformation_matrix(pose, switch_w=False):
    pose = np.asarray(pose, dtype=np.float64)
    out = np.eye(4)
    out[:3, :3] = get_quaternion_rotation_matrix(pose[3:], switch_w=switch_w)
    out[:3, 3] = pose[:3]
    return out


def get_transformation_matrix_inverse(trans_mat):
    out = np.array(trans_mat, copy=True)
    out[:3, :3] = trans_mat[:3, :3].T
    out[:3, 3] = (-out[:3, :3]) @ trans_mat[:3, 3]
    return out


def get_pose_from_matrix(trans_mat):
    return np.concatenate((trans_mat[:3, 3], get_q_from_matrix(np.array(trans_mat[:3, :3], copy=True))))


def get_pose_inverse(pose):
    return get_pose_from_matrix(get_transformation_matrix_inverse(get_transformation_matrix(pose)))


def get_base2cam_matrix(ee2cam_pose, ee2robot_pose):
    ee2cam = get_transformation_matrix(ee2cam_pose, switch_w=False)
    robot2ee = get_transformation_matrix_inverse(get_transformation_matrix(ee2robot_pose, switch_w=False))
    return ee2cam @ robot2ee


def get_base2cam_pose(ee2cam_pose, ee2robot_pose):
    return get_pose_from_matrix(get_base2cam_matrix(ee2cam_pose, ee2robot_pose))


def transform_pose2pose_matrix(pose1, pose2):
    return get_transformation_matrix(pose1, switch_w=False) @ get_transformation_matrix(pose2, switch_w=False)


def transform_pose2pose(pose1, pose2):
    return get_pose_from_matrix(transform_pose2pose_matrix(pose1, pose2))

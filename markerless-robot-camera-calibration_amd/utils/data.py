"""utils/data.py:13-34 get_farthest_point_sample_idx on libsvhip (one workgroup per cloud)."""
import numpy as np
import torch

from ..model.pointnet2_utils import farthest_point_sample


def get_farthest_point_sample_idx(point, npoint, start=None):
    """point [N, D] (xyz in the first 3 columns) -> int32[npoint].  The reference draws the first index with
    np.random.randint(0, N); pass `start` to pin it."""
    point = np.asarray(point)
    N = point.shape[0]
    if start is None:
        start = np.random.randint(0, N)
    xyz = torch.from_numpy(np.ascontiguousarray(point[:, :3], dtype=np.float32)).cuda().unsqueeze(0)
    st = torch.tensor([int(start)], dtype=torch.int64, device=xyz.device)
    return farthest_point_sample(xyz, npoint, start=st)[0].cpu().numpy().astype(np.int32)


def get_farthest_point_sample(point, npoint):
    return point[get_farthest_point_sample_idx(point, npoint)]

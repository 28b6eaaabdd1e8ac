"""Pre-voxelisation transforms of the reference (utils/preprocess.py:8-56): numpy host code, two passes over N."""
import numpy as np


def center_at_origin(points):
    origin_offset = (points.max(axis=0) + points.min(axis=0)) / 2
    return points - origin_offset, origin_offset


def base_at_origin(points):
    origin_base_offset = points.min(axis=0)
    return points - origin_base_offset, origin_base_offset


def _minmax01(col):
    lo, hi = col.min(), col.max()
    rng = hi - lo
    return (col - lo) / (rng if rng != 0 else 1.0)


def normalize_colors(rgb_input, is_color_in_range_0_255=False):
    """/255 when the data looks like 0..255, per-channel min-max when negative values are present, then shift
    [0,1] -> [-0.5,0.5] (the data-dependent branches of utils/preprocess.py:20-37)."""
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
    if len(pc.shape) == 2:
        pc = np.array(pc, copy=True)
        pc = pc - pc.mean(0)
        pc /= np.max(np.linalg.norm(pc, axis=-1))
        return pc
    pc = np.asarray(pc)
    pc = pc - pc.mean(1).reshape(-1, 1, 3)
    return pc / np.max(np.linalg.norm(pc, axis=-1), axis=-1).reshape(-1, 1, 1)

"""Frame sources in the reference's on-disk format: pickle files holding
{"points", "rgb", "labels", "instance_labels", "pose" (x,y,z,qx,qy,qz,qw), "joint_angles"[, "robot2ee_pose"]}
(README.md:55-62), listed per split in a JSON file with a "position" tag (app/data_engine.py:53-158,
utils/file_utils.py:4-16).  `write_frame_pickle` produces that format from a synthetic scene so the loaders can be
exercised without the authors' data (their sample frames are missing blobs)."""
import json
import os
import pickle
from datetime import datetime
from itertools import cycle

import numpy as np

from ..utils.transformation import get_quaternion_rotation_matrix, switch_w
from .dto import PointCloudDTO, RawDTO


def load_alive_file(filename):
    with open(filename, "rb") as fp:
        return pickle.load(fp, encoding="bytes")


def get_roi_mask(points, min_x=-500, max_x=500, min_y=-500, max_y=500, min_z=-500, max_z=500, offset=0.0):
    """Axis-aligned box test with open bounds (utils/data.py:58-75)."""
    p = np.asarray(points)
    lo = np.array([min_x, min_y, min_z]) - offset
    hi = np.array([max_x, max_y, max_z]) + offset
    return np.all((p < hi) & (p > lo), axis=1)


def get_ee_idx(points, pose, switch_w=True, ee_dim=None, arm_idx=None):
    """Indices of the points inside the end-effector box expressed in the EE frame (utils/data.py:78-103)."""
    dim = {"min_z": -0.006, "max_z": 0.12, "min_x": -0.05, "max_x": 0.05, "min_y": -0.11, "max_y": 0.11}
    if isinstance(ee_dim, dict):
        dim.update(ee_dim)
    rot = get_quaternion_rotation_matrix(np.asarray(pose[3:], dtype=np.float64), switch_w=switch_w)
    local = (np.asarray(points, dtype=np.float64) - np.asarray(pose[:3], dtype=np.float64)) @ rot  # R^T (p - t)
    ee_idx = np.where(get_roi_mask(local, **dim))[0]
    if arm_idx is not None:
        ee_idx = ee_idx[np.isin(ee_idx, arm_idx, assume_unique=True)]
    return ee_idx


def write_frame_pickle(path, scene):
    """Dump a synthetic scene (mrcc_amd.synth.gen_scene) in the reference's frame format (pose stored x,y,z,qx,qy,qz,qw;
    the EE points carry the ARM label 1 — the loader re-derives label 2 from the pose box, as the reference does)."""
    pose = np.asarray(scene["pose"], dtype=np.float32)
    labels = np.where(scene["segmentation"] == 2, 1, scene["segmentation"]).astype(np.float32)
    xyzw = lambda p: np.concatenate([p[:3], p[4:7], p[3:4]]).astype(np.float32)
    data = {"points": scene["points"].astype(np.float32), "rgb": scene["rgb"].astype(np.float32), "labels": labels,
            "instance_labels": labels.copy(), "pose": xyzw(pose), "joint_angles": np.zeros(9, np.float32)}
    if scene.get("ee2base_pose") is not None:
        data["robot2ee_pose"] = xyzw(np.asarray(scene["ee2base_pose"], dtype=np.float32))
    with open(path, "wb") as fp:
        pickle.dump(data, fp)


class PickleDataEngine:
    """app/data_engine.py:53-158: split JSON -> frames sorted by (position, numeric file name)."""

    EE_DIM = {"min_z": -0.0095, "max_z": 0.13, "min_x": -0.05, "max_x": 0.05, "min_y": -0.13, "max_y": 0.13}

    def __init__(self, data_path, split="test", cyclic=True):
        with open(data_path, "r") as fp:
            self.data = {split: []}
            self.data.update(json.load(fp))
        base = os.path.dirname(os.path.abspath(data_path))
        for item in self.data[split]:
            if not os.path.isabs(item["filepath"]):
                item["filepath"] = os.path.join(base, item["filepath"])
        self.data[split].sort(key=lambda x: (x["position"], int(os.path.basename(x["filepath"]).split(".")[0])))
        self.items = self.data[split]
        self.data_pool = cycle(self.items) if cyclic else iter(self.items)

    def __len__(self):
        return len(self.items)

    def _next(self):
        try:
            return next(self.data_pool)
        except StopIteration:
            return None

    @staticmethod
    def _unpack(data):
        if isinstance(data, dict):
            return (data["points"], data["rgb"], data.get("labels"), data["pose"], data.get("robot2ee_pose"))
        points, rgb, labels, _, pose = data  # the older tuple format
        return points, rgb, labels, pose, None

    def get(self) -> PointCloudDTO:
        item = self._next()
        if item is None:
            return None
        points, rgb, _, pose, ee2base = self._unpack(load_alive_file(item["filepath"]))
        return PointCloudDTO(points=points, rgb=rgb, timestamp=datetime.utcnow(),
                             ee2base_pose=None if ee2base is None else switch_w(ee2base),
                             gt_pose=None if pose is None else switch_w(pose))

    def get_raw(self) -> RawDTO:
        item = self._next()
        if item is None:
            return None
        points, rgb, labels, pose, ee2base = self._unpack(load_alive_file(item["filepath"]))
        points = points.astype(np.float32)
        rgb = rgb.astype(np.float32)
        labels = labels.astype(np.int64)
        pose = switch_w(pose)  # -> (x, y, z, qw, qx, qy, qz)
        if ee2base is not None:
            ee2base = switch_w(ee2base)
        ee_idx = get_ee_idx(points, pose, ee_dim=self.EE_DIM, arm_idx=np.where(labels == 1)[0], switch_w=False)
        labels[ee_idx] = 2
        return RawDTO(points, rgb, pose, labels, ee2base_pose=ee2base,
                      other={"filepath": item["filepath"], "position": item["position"]})

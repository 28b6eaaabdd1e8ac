"""Minimal stand-in for the reference's utils/config.py:15-102 Config singleton.

The reference parses sys.argv and mkdirs exp_path at import time; a library cannot do that, so this
Config is an explicit namespace with the defaults of config/default.yaml for the keys the hot path reads
(DATA.*, STRUCTURE.*, MODE, INFERENCE.*).  `Config()` returns the process-wide instance; `Config().update({...})`
overrides keys recursively (the reference's --override semantics).
"""
import copy
from types import SimpleNamespace

_DEFAULTS = {
    "MODE": "inference",
    "PARAM": {"ee_r": 0.02},
    "DATA": {
        "scale": 100, "input_channel": 3, "classes": 3, "ignore_label": -100, "pose_dim": 7, "data_type": "ee_seg",
        "voxelize_position": False, "center_at_origin": True, "num_of_keypoints": 6,
        "num_of_dense_input_points": 2048, "use_coordinates_as_features": False, "use_point_normals": False,
        "pointcloud_sampling_method": "farthest", "max_npoint": 250000,
    },
    "STRUCTURE": {
        "m": 32, "block_reps": 2, "use_joint_angles": False, "bottleneck": True, "backbone": "minkunet",
        "encode_only": False, "compute_confidence": False,
    },
    "INFERENCE": {
        "ee_point_counts_threshold": 512, "icp_enabled": False, "num_of_dense_input_points": 2048,
        "camera_link_transformation_pose": None,
        "SANITY": {"min_num_of_ee_points": 2048},
        "SEGMENTATION": {"backbone": "robotnet_segmentation", "scale": 200, "center_at_origin": True,
                         "checkpoint": None},
        "TRANSLATION": {"backbone": "minkunet", "scale": 200, "center_at_origin": True, "move_ee_to_origin": True,
                        "magic_enabled": True, "checkpoint": None},
        "ROTATION": {"backbone": "minkunet", "encode_only": True, "scale": 200, "center_at_origin": True,
                     "checkpoint": None},
        "KEY_POINTS": {"backbone": "minkunet", "scale": 200, "center_at_origin": True, "conf_threshold": 0.75,
                       "use_coordinates_as_features": False, "num_of_keypoints": 6, "error_margin": 0.05,
                       "pointcloud_sampling_method": "farthest", "checkpoint": None},
    },
}


def _to_ns(d):
    return SimpleNamespace(**{k: _to_ns(v) if isinstance(v, dict) else v for k, v in d.items()})


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v


class Config:
    _instance = None

    def __new__(cls):
        if cls._instance is None:
            inst = super().__new__(cls)
            inst._dict = copy.deepcopy(_DEFAULTS)
            inst._refresh()
            cls._instance = inst
        return cls._instance

    def _refresh(self):
        for k, v in self._dict.items():
            setattr(self, k, _to_ns(v) if isinstance(v, dict) else v)

    def __call__(self):
        return self._dict

    def update(self, overrides):
        _merge(self._dict, overrides)
        self._refresh()
        return self

    @classmethod
    def reset(cls):
        cls._instance = None

"""Evaluation harness: the per-frame metric recipe and aggregation of the reference's app/test.py:73-329 (TestApp),
without its xlsx writer.

Per frame (same order as run_tests): segmentation metrics -> EE crop -> rotation + translation -> pose metrics + ADD ->
key points -> Kabsch pose from key points -> pose metrics + ADD + key-point error -> sanity -> base poses; after the
frames, per-position and overall calibration.  Aggregates are mean / min / max / median / stdev, the five rows the
reference writes per column (app/test.py:295-329).
"""
import statistics
from collections import defaultdict

import numpy as np
import torch

from ..utils import metrics, preprocess
from ..utils.transformation import get_base2cam_pose
from .dto import TestResultDTO


def aggregate(values):
    """mean / min / max / median / stdev of a list, "N/A" where the reference writes it."""
    values = [float(v) for v in values]
    if not values:
        return {k: "N/A" for k in ("mean", "min", "max", "median", "stdev")}
    return {"mean": statistics.mean(values), "min": min(values), "max": max(values),
            "median": statistics.median(values), "stdev": statistics.stdev(values) if len(values) > 1 else "N/A"}


class TestApp:
    __test__ = False  # not a pytest class

    def __init__(self, inference_engine, ee_point_counts_threshold=None, evaluate_segmentation=True):
        self._engine = inference_engine
        cfg = inference_engine._config
        self.ee_threshold = (cfg.INFERENCE.ee_point_counts_threshold if ee_point_counts_threshold is None
                             else ee_point_counts_threshold)
        self.evaluate_segmentation = evaluate_segmentation
        self.clear_results()

    def clear_results(self):
        self.instance_results = defaultdict(dict)
        self.predictions = defaultdict(list)
        self.calibration = None

    def run_tests(self, frames):
        """frames: iterable of dicts with points, rgb, segmentation, pose, key_points, ee2base_pose, position
        (mrcc_amd.synth.gen_scene).  Returns {"instances", "positions", "overall", "calibration"}."""
        self.clear_results()
        eng = self._engine
        for i, d in enumerate(frames):
            key = f"{d['position']}/{i}"
            inst = self.instance_results[key]
            inst["position"] = d["position"]
            rgb = preprocess.normalize_colors(d["rgb"])
            seg = d["segmentation"]
            if self.evaluate_segmentation:
                seg = eng.predict_segmentation(d["points"], rgb)
                inst["segmentation"] = metrics.compute_segmentation_metrics(d["segmentation"], seg)
            result = TestResultDTO(segmentation=seg)
            ee_idx = np.where(seg == 2)[0]
            if len(ee_idx) < self.ee_threshold:
                self.instance_results.pop(key)  # "fail min # points" (app/test.py:110-113)
                continue
            ee_pts = d["points"][ee_idx]
            ee_rgb = torch.from_numpy(rgb[ee_idx]).to(dtype=torch.float32)
            gt_pose = np.asarray(d["pose"], dtype=np.float64)
            # rotation + translation network pose
            q = eng.predict_rotation(ee_pts, ee_rgb)
            pos, _ = eng.predict_translation(ee_pts, ee_rgb, q=q)
            result.ee_pose = np.concatenate((pos, q))
            inst.update({f"nn_{k}": v for k, v in metrics.compute_pose_metrics(gt_pose, result.ee_pose).items()})
            ee_centered, _ = preprocess.center_at_origin(ee_pts)
            inst["nn_ADD"] = metrics.compute_ADD_np(ee_centered, gt_pose, result.ee_pose)
            # key-point pose
            kp_coords, kp_classes, _ = eng.predict_key_points(ee_pts, ee_rgb)
            result.key_points = list(zip(kp_classes, kp_coords))
            result.key_points_pose = eng.predict_pose_from_kp(kp_coords, kp_classes)
            if result.key_points_pose is not None:
                inst.update({f"kp_{k}": v
                             for k, v in metrics.compute_pose_metrics(gt_pose, result.key_points_pose).items()})
                inst["kp_ADD"] = metrics.compute_ADD_np(ee_centered, gt_pose, result.key_points_pose)
                inst["kp_error"] = metrics.compute_kp_error(d["key_points"], kp_coords, kp_classes)
            result.is_confident = bool(eng.check_sanity(_Cloud(d["points"]), result))
            inst["is_confident"] = result.is_confident
            if d.get("ee2base_pose") is not None:
                result.base_pose = get_base2cam_pose(result.ee_pose, d["ee2base_pose"])
                gt_base = get_base2cam_pose(gt_pose, d["ee2base_pose"])
                inst.update({f"base_{k}": v for k, v in metrics.compute_pose_metrics(gt_base, result.base_pose).items()})
                if result.key_points_pose is not None:
                    result.key_points_base_pose = get_base2cam_pose(result.key_points_pose, d["ee2base_pose"])
            self.predictions[d["position"]].append(result)
        if self.predictions:
            self.calibration = eng.calibrate(self.predictions)
        return self.summary()

    def summary(self):
        numeric = defaultdict(lambda: defaultdict(list))
        for inst in self.instance_results.values():
            for scope in (inst["position"], "overall"):
                for k, v in inst.items():
                    if k == "segmentation":
                        for mk in ("accuracy", "precision", "recall", "miou"):
                            numeric[scope][f"seg_{mk}"].append(v[mk])
                    elif isinstance(v, (int, float, np.floating)) and not isinstance(v, bool):
                        numeric[scope][k].append(v)
        agg = {scope: {k: aggregate(vals) for k, vals in cols.items()} for scope, cols in numeric.items()}
        overall = agg.pop("overall", {})
        return {"instances": dict(self.instance_results), "positions": agg, "overall": overall,
                "calibration": self.calibration}


class _Cloud:
    def __init__(self, points):
        self.points = points

"""Evaluation harness plumbing with a stub engine (no GPU): metrics of known perturbations come out as expected and
the aggregation has the reference's five rows."""
import numpy as np


class _StubEngine:
    """Predicts the ground truth with controlled errors."""

    def __init__(self, frames):
        from mrcc_amd.app.inference_engine import InferenceEngine

        self._real = InferenceEngine(calibration_only=True)
        self._config = self._real._config
        self.frames = {id(f["points"]): f for f in frames}
        self.cur = None

    def predict_segmentation(self, points, rgb):
        self.cur = self.frames[id(points)]
        seg = self.cur["segmentation"].copy()
        seg[:200] = 0  # a known amount of label noise
        return seg

    def predict_rotation(self, pts, rgb):
        return self.cur["pose"][3:].copy()

    def predict_translation(self, pts, rgb, q=None):
        return self.cur["pose"][:3] + np.array([0.01, 0.0, 0.0]), np.zeros(3)

    def predict_key_points(self, pts, rgb):
        return self.cur["key_points"].copy(), np.arange(6), np.ones(6)

    def predict_pose_from_kp(self, kp_coords, kp_classes):
        import sys, os
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
        import sv_oracle as O

        R, t = O.get_rigid_transform_3D(self._real.reference_key_points[kp_classes], kp_coords)
        return np.concatenate((t, O.get_q_from_matrix(R)))

    def check_sanity(self, data, result):
        return True

    def calibrate(self, predictions):
        return {k: len(v) for k, v in predictions.items()}


def test_harness_reports_known_errors():
    import mrcc_amd
    from mrcc_amd.app.evaluate import TestApp, aggregate

    frames = [mrcc_amd.synth.gen_scene(s, n_bg=3000, n_arm=500, n_ee=800) for s in range(4)]
    for f in frames:
        f["ee2base_pose"] = None  # the base-pose step converts matrices to quaternions on the GPU (no CPU fallback)
    app = TestApp(_StubEngine(frames), ee_point_counts_threshold=100)
    # compute_ADD_np goes through the HIP library: substitute the oracle's restatement for this CPU-only test
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import sv_oracle as O
    from mrcc_amd.utils import metrics as M

    orig = M.compute_ADD_np
    M.compute_ADD_np = O.compute_ADD_np
    try:
        out = app.run_tests(frames)
    finally:
        M.compute_ADD_np = orig
    assert len(out["instances"]) == 4
    ov = out["overall"]
    assert abs(ov["nn_dist_position"]["mean"] - 0.01) < 1e-9 and ov["nn_angle_diff"]["max"] < 1e-6
    assert abs(ov["nn_ADD"]["mean"] - 0.01) < 1e-9  # pure 1 cm translation error -> ADD = 1 cm
    assert ov["kp_dist_position"]["mean"] < 5e-3 and ov["kp_error"]["mean"] < 5e-3  # 1 mm key-point noise
    assert 0.9 < ov["seg_miou"]["mean"] < 1.0 and ov["seg_accuracy"]["min"] > 0.9
    assert set(ov["nn_ADD"]) == {"mean", "min", "max", "median", "stdev"}
    assert sum(out["calibration"].values()) == 4 and set(out["positions"]) <= {"p1", "p2", "p3"}
    assert aggregate([])["mean"] == "N/A" and aggregate([1.0])["stdev"] == "N/A"

"""Wall time of every stage of InferenceEngine.predict (the reference's production entry point,
app/inference_engine.py:281-382) on synthetic inputs with random-init weights: which host-side step is next.
    python tools/engine_stages.py [points in the frame]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mrcc_amd  # noqa: E402
from mrcc_amd.app.dto import PointCloudDTO  # noqa: E402
from mrcc_amd.app.inference_engine import InferenceEngine  # noqa: E402
from mrcc_amd.utils import preprocess  # noqa: E402
from mrcc_amd.utils.config import Config  # noqa: E402

n_frame = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
Config.reset()
Config().update({"INFERENCE": {"SEGMENTATION": {"scale": 50}, "ROTATION": {"scale": 100},
                               "KEY_POINTS": {"scale": 100, "conf_threshold": 0.0},
                               "ee_point_counts_threshold": 64, "SANITY": {"min_num_of_ee_points": 64}}})
eng = InferenceEngine(allow_random_init=True, seed=7)
scene = mrcc_amd.synth.gen_scene(0, n_bg=n_frame - 8192, n_arm=4096, n_ee=4096)
ee_pts, ee_rgb, pose, kps = mrcc_amd.synth.gen_ee_crop(0, n=4096)
ee_rgb_t = torch.from_numpy(ee_rgb)


def timed(name, fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = fn()
    torch.cuda.synchronize()
    print(f"{name:58s} {(time.perf_counter() - t0) / reps * 1e3:9.2f} ms", flush=True)
    return out


rgb = timed("preprocess.normalize_colors (host numpy)", lambda: preprocess.normalize_colors(scene["rgb"]))
seg = timed(f"predict_segmentation ({len(scene['points'])} points, incl. cluster filter)",
            lambda: eng.predict_segmentation(scene["points"], rgb))
print(f"   EE predictions kept: {(seg == 2).sum()}")
q = timed("predict_rotation (4096-point EE crop)", lambda: eng.predict_rotation(ee_pts, ee_rgb_t))
timed("predict_translation", lambda: eng.predict_translation(ee_pts, ee_rgb_t, q=q))
kp = timed("predict_key_points", lambda: eng.predict_key_points(ee_pts, ee_rgb_t))
timed("predict_pose_from_kp (6 ground-truth key points)", lambda: eng.predict_pose_from_kp(kps, np.arange(6)))
if eng.match_icp is not None:
    timed("match_icp", lambda: eng.match_icp(ee_pts, pose))
data = PointCloudDTO(points=scene["points"], rgb=scene["rgb"], ee2base_pose=scene["ee2base_pose"])
timed("predict() end to end", lambda: eng.predict(data), reps=3)

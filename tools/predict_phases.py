"""Host time per phase of the per-frame InferenceEngine.predict on labelled 200k-point scenes:  python tools/predict_phases.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import mrcc_amd  # noqa: E402
from mrcc_amd.app.dto import PointCloudDTO  # noqa: E402
from mrcc_amd.app.inference_engine import InferenceEngine  # noqa: E402
from mrcc_amd.utils import preprocess  # noqa: E402
from mrcc_amd.utils.config import Config  # noqa: E402
import mrcc_amd.app.inference_engine as IE  # noqa: E402
import mrcc_amd.app.pipeline as PL  # noqa: E402

Config.reset()
Config().update({"INFERENCE": {"SEGMENTATION": {"scale": 50}, "ROTATION": {"scale": 100},
                               "KEY_POINTS": {"scale": 100, "conf_threshold": 0.0}}})
eng = InferenceEngine(allow_random_init=True, seed=1)
mrcc_amd.synth.wire_color_keyed_labels(eng._segmentation_model)
scenes = [mrcc_amd.synth.gen_scene(sd, n_bg=200_000 - 4000 - 4096, n_arm=4000, n_ee=4096, room=2.4, keyed_colors=True) for sd in range(4)]
dtos = [PointCloudDTO(points=sc["points"], rgb=sc["rgb"], ee2base_pose=sc["ee2base_pose"]) for sc in scenes]
acc = {}


def timed(name, fn):
    def wrap(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return wrap


for name in ("predict_segmentation", "_pose_enqueue", "_pose_collect", "_pose_nets_enqueue", "_solve_rigid", "predict_translation",
             "check_sanity"):
    setattr(eng, name, timed(name, getattr(eng, name)))
IE.preprocess.normalize_colors = timed("normalize_colors", preprocess.normalize_colors)
runner = eng._crop_runner()
runner.run = timed("crop_runner.run", runner.run)
runner.download = timed("crop_runner.download", runner.download)
orig_sparse = PL.ME.TensorField.sparse
PL.ME.TensorField.sparse = timed("TensorField.sparse", orig_sparse)
for _ in range(3):
    for d in dtos:
        eng.predict(d)
acc.clear()
n = 16
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(n):
    eng.predict(dtos[i % 4])
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
print(f"predict(): {ms:.2f} ms/frame; host ms/frame: " + ", ".join(f"{k} {v / n * 1e3:.2f}" for k, v in acc.items()))

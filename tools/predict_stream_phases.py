"""Host time per phase of InferenceEngine.predict_stream on labelled 200k-point scenes:  python tools/predict_stream_phases.py [group]"""
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
from mrcc_amd.utils.config import Config  # noqa: E402

Config.reset()
Config().update({"INFERENCE": {"SEGMENTATION": {"scale": 50}, "ROTATION": {"scale": 100},
                               "KEY_POINTS": {"scale": 100, "conf_threshold": 0.0}}})
eng = InferenceEngine(allow_random_init=True, seed=1)
mrcc_amd.synth.wire_color_keyed_labels(eng._segmentation_model)
scenes = [mrcc_amd.synth.gen_scene(sd, n_bg=200_000 - 4000 - 4096, n_arm=4000, n_ee=4096, room=2.4, keyed_colors=True) for sd in range(4)]
dtos = [PointCloudDTO(points=sc["points"], rgb=sc["rgb"], ee2base_pose=sc["ee2base_pose"]) for sc in scenes]
seq = [dtos[i % 4] for i in range(32)]
acc = {}


def timed(name, fn):
    def wrap(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
    return wrap


for name in ("_pose_enqueue", "_pose_collect", "_rotation_enqueue", "_key_points_enqueue", "_solve_rigid", "predict_translation",
             "check_sanity"):
    setattr(eng, name, timed(name, getattr(eng, name)))
from mrcc_amd.utils import preprocess  # noqa: E402
preprocess.normalize_colors = timed("normalize_colors", preprocess.normalize_colors)
import mrcc_amd.app.inference_engine as IE  # noqa: E402
IE.preprocess.normalize_colors = preprocess.normalize_colors

for group in [int(a) for a in sys.argv[1:]] or [4]:
    list(eng.predict_stream(iter(seq[:8]), group=group))
    st = eng._seg_streams[(3, 50)]
    for rep in range(2):
        acc.clear()
        st.host_s = {k: 0 for k in st.host_s}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = list(eng.predict_stream(iter(seq), group=group))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / len(seq) * 1e3
        n = len(seq)
        ph = ", ".join(f"{k} {v / n * 1e3:.2f}" for k, v in st.host_s.items() if k != "frames")
        po = ", ".join(f"{k} {v / n * 1e3:.2f}" for k, v in acc.items())
        print(f"group {group}: {ms:.2f} ms/frame; seg host ms/frame: {ph}; pose host ms/frame: {po}")

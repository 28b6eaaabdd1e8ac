"""GPU / host memory of the engine's paths over many frames (no growth expected): per-frame predict(), predict_stream with the
pose thread.  python tools/engine_memory.py"""
import os
import resource
import sys

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


def report(tag):
    torch.cuda.synchronize()
    print("%-28s GPU allocated %.2f GiB, reserved %.2f GiB, max allocated %.2f GiB; host max RSS %.2f GiB" % (
        tag, torch.cuda.memory_allocated() / 2 ** 30, torch.cuda.memory_reserved() / 2 ** 30,
        torch.cuda.max_memory_allocated() / 2 ** 30, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 2 ** 20), flush=True)


for rep in range(3):
    for i in range(60):
        eng.predict(dtos[i % 4])
    report(f"predict() x {60 * (rep + 1)}")
for rep in range(3):
    n = sum(1 for _ in eng.predict_stream(iter(dtos[i % 4] for i in range(200))))
    report(f"predict_stream x {200 * (rep + 1)} ({n})")

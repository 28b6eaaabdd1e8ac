"""Engine-path segmentation at 200k points per frame: the synchronous per-frame call (the reference's loop,
app/main.py:432-456) against InferenceEngine.predict_segmentation_stream, with the stream's host time per phase.
    python tools/engine_stream_phases.py [compute streams ...]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import mrcc_amd  # noqa: E402
from mrcc_amd.app.inference_engine import InferenceEngine  # noqa: E402
from mrcc_amd.utils.config import Config  # noqa: E402

Config.reset()
Config().update({"INFERENCE": {"SEGMENTATION": {"scale": 50}}})
eng = InferenceEngine(allow_random_init=True, seed=1)
pool = [mrcc_amd.synth.gen_room(200_000, 2.4, s)[:2] for s in range(4)]
frames = [pool[i % 4] for i in range(32)]
for _ in range(2):
    eng.predict_segmentation(*pool[0])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(8):
    ref = eng.predict_segmentation(*frames[i])
torch.cuda.synchronize()
print(f"predict_segmentation, one frame at a time: {(time.perf_counter() - t0) / 8 * 1e3:.2f} ms/frame "
      f"(EE labels kept in the last frame: {(ref == 2).sum()})")
one = getattr(eng, "_one_frame_stream", None)
if one is not None:
    one.host_s = {k: 0 for k in one.host_s}
    lat = []
    for i in range(8, 24):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.predict_segmentation(*frames[i])
        lat.append((time.perf_counter() - t0) * 1e3)
    n = one.host_s["frames"]
    ph = ", ".join(f"{k} {v / n * 1e3:.2f}" for k, v in one.host_s.items() if k != "frames")
    print(f"   per call: median {np.median(lat):.2f} ms, min {min(lat):.2f}, max {max(lat):.2f}; host ms/frame: {ph}")
for streams in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]:
    list(eng.predict_segmentation_stream(iter(frames[:8]), compute_streams=streams))
    st = eng._seg_streams[(streams, 50)]
    for rep in range(2):
        st.host_s = {k: 0 for k in st.host_s}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = list(eng.predict_segmentation_stream(iter(frames), compute_streams=streams))
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / len(frames) * 1e3
        n = st.host_s["frames"]
        ph = ", ".join(f"{k} {v / n * 1e3:.2f}" for k, v in st.host_s.items() if k != "frames")
        print(f"predict_segmentation_stream, {streams} compute stream(s): {ms:.2f} ms/frame; host ms/frame: {ph}")
    assert np.array_equal(got[7], ref)

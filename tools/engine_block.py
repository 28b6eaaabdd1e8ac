"""bench.py's `engine` block alone (per-frame and streamed InferenceEngine timings):  python tools/engine_block.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

print(json.dumps(bench.engine_block(torch.device("cuda:0")), indent=1))

"""Frames/s of the strong-scaling block's synthetic frame source alone (app/sharding.ordered_prefetch over synth.gen_room):
    python tools/source_rate.py [threads ...]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mrcc_amd  # noqa: E402
from mrcc_amd.app.sharding import ordered_prefetch  # noqa: E402

mrcc_amd.synth.gen_room(200_000, 2.4, 0)
print("cores", len(os.sched_getaffinity(0)))
for th in [int(a) for a in sys.argv[1:]] or [2, 4, 6, 8, 12, 16]:
    t = time.perf_counter()
    n = sum(1 for _ in ordered_prefetch(lambda sd: mrcc_amd.synth.gen_room(200_000, 2.4, sd)[:2], list(range(96)), threads=th,
                                        lookahead=2 * th + 2))
    print(f"{th} threads: {n / (time.perf_counter() - t):.1f} frames/s")

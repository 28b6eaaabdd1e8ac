"""N > 1 path on CPU: two gloo ranks shard the frame sequence and exchange ONE metrics record."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["REPO_ROOT"])
from mrcc_amd.app.sharding import frame_seeds_for_rank, gather_metrics
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
seeds = frame_seeds_for_rank(10, rank, world)
rec = {"frames": len(seeds), "elapsed": 1.0 + rank, "confusion": np.eye(3, dtype=np.int64) * (rank + 1),
       "seed_sum": int(sum(seeds))}
out = gather_metrics(rec, device="cpu")
if rank == 0:
    print(json.dumps({"frames": out["frames"], "elapsed_max": out["elapsed_max"],
                      "confusion_trace": int(np.trace(out["confusion"])), "seed_sum": out["seed_sum"],
                      "per_rank_frames": out["per_rank_frames"]}))
dist.destroy_process_group()
'''


def test_two_rank_sharding_and_single_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, REPO_ROOT=ROOT, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", "29731", str(script)]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-2000:]
    import json

    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["frames"] == 10 and out["per_rank_frames"] == [5, 5]
    assert out["elapsed_max"] == 2.0
    assert out["confusion_trace"] == 3 * (1 + 2)
    assert out["seed_sum"] == sum(range(10))  # every frame processed exactly once across ranks


def test_shard_rule_is_a_partition():
    from mrcc_amd.app.sharding import frame_seeds_for_rank

    for world in (1, 2, 4, 8):
        allf = sorted(s for r in range(world) for s in frame_seeds_for_rank(512, r, world))
        assert allf == list(range(512))
        assert frame_seeds_for_rank(512, 0, world)[:2] == [0, world] if world > 1 else True

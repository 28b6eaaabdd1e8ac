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


def _run_bench(args, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                          text=True, timeout=timeout)


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` (no torch.distributed environment) must start 2 ranks itself and relay rank 0's
    line with n_gpus = 2 (--gpus was parsed and ignored).  CPU rehearsal of everything around the GPU work."""
    import json

    res = _run_bench(["--gpus", "2", "--steps", "6", "--launcher-selftest"])
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["per_rank_frames"] == [6, 6] and out["frames"] == 12
    assert out["seed_sum"] == sum(range(12)) and abs(out["elapsed_max"] - 0.002) < 1e-9
    # N = 1 needs no launcher
    res = _run_bench(["--gpus", "1", "--steps", "3", "--launcher-selftest"])
    assert res.returncode == 0, res.stderr[-2000:]
    assert json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])["n_gpus"] == 1


def test_bench_rejects_world_size_mismatch_and_failing_ranks():
    # a torch.distributed environment that disagrees with --gpus is an error, not a silent 1-rank run
    res = _run_bench(["--gpus", "4", "--launcher-selftest"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert res.returncode != 0 and "WORLD_SIZE=1" in res.stderr
    # a rank that dies makes the launcher exit non-zero (no GPU here: the real bench fails loudly in every rank)
    res = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"], timeout=600)
    assert res.returncode != 0

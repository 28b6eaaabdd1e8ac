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
    # a rank that dies makes the launcher exit non-zero.  Deterministic and CPU-only: the selftest makes rank 1 exit 3
    # before the rendezvous (this test must never start the real GPU bench: on a multi-GPU host that would succeed)
    res = _run_bench(["--gpus", "2", "--steps", "2", "--launcher-selftest", "--selftest-fail-rank", "1"])
    assert res.returncode != 0 and "fails on purpose" in res.stderr
    assert not any(l.startswith("{") for l in res.stdout.splitlines())  # no JSON line from a failed job


def test_strong_scaling_mode_partitions_a_fixed_job():
    """Cfg-4 (`--total-frames T`): the SAME T frames at every N - rank r takes seeds r, r + N, ... - frame counts, one
    gather, `scaling: strong`; 2 and 4 gloo ranks, T not a multiple of N."""
    import json

    for world, total in ((2, 11), (4, 18), (1, 5)):
        res = _run_bench(["--gpus", str(world), "--total-frames", str(total), "--launcher-selftest"])
        assert res.returncode == 0, res.stderr[-2000:]
        out = json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][-1])
        assert out["scaling"] == "strong" and out["n_gpus"] == world and out["frames"] == total
        want = [len(range(r, total, world)) for r in range(world)]
        assert out["per_rank_frames"] == want and out["seed_sum"] == sum(range(total))  # each frame exactly once
        assert len(out["elapsed_repeats_max"]) == 5  # per repeat: the slowest rank's time
        assert abs(out["elapsed_repeats_max"][0] - 0.001 * world) < 1e-12
        assert out["pinned_cores"] >= 1
        # rank-by-rank figures travel in the same gather: every rank's own elapsed time, host time and core count, so a
        # slow or host-bound rank is identifiable from rank 0's line
        assert len(out["per_rank_elapsed"]) == world
        assert all(abs(out["per_rank_elapsed"][r] - 0.001 * (r + 1)) < 1e-12 for r in range(world))
        assert out["per_rank"]["host_prepare_ms"] == [1.5 + r for r in range(world)]
        assert len(out["per_rank"]["pinned_cores"]) == world and min(out["per_rank"]["pinned_cores"]) >= 1


def test_rank_pinning_from_sysfs_topology(tmp_path):
    """Ranks are pinned to their GPU's NUMA node before any GPU call (app/sharding.py pin_rank): fake sysfs with 4 GPUs on
    2 nodes (render nodes in minor order = device order), a non-AMD render node in between, and the no-information case."""
    from mrcc_amd.app import sharding

    def mk(minor, vendor, numa):
        d = tmp_path / "class" / "drm" / f"renderD{minor}" / "device"
        d.mkdir(parents=True)
        (d / "vendor").write_text(vendor + "\n")
        (d / "numa_node").write_text(f"{numa}\n")

    mk(128, "0x1002", 0)
    mk(129, "0x10de", 0)  # not an AMD GPU: skipped
    mk(130, "0x1002", 0)
    mk(131, "0x1002", 1)
    mk(132, "0x1002", 1)
    for node, cpus in ((0, "0-7,32-39"), (1, "8-15,40-47")):
        d = tmp_path / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus + "\n")
    allowed = set(range(48))
    got = [sharding.pin_rank(r, 4, sysfs=str(tmp_path), allowed=allowed, apply=False) for r in range(4)]
    assert [g["numa_node"] for g in got] == [0, 0, 1, 1]
    assert all(g["cores"] == 8 and g["how"] == "gpu numa node" for g in got)  # two ranks share each 16-core node
    assert (got[0]["first_core"], got[0]["last_core"]) == (0, 7) and (got[1]["first_core"], got[1]["last_core"]) == (32, 39)
    assert (got[2]["first_core"], got[3]["last_core"]) == (8, 47)
    cpus, numa = sharding.gpu_numa_cpus(1, sysfs=str(tmp_path), visible=[3, 0])  # HIP_VISIBLE_DEVICES=3,0: rank 1 = GPU 0
    assert numa == 0 and 0 in cpus
    # no topology information: even split of the allowed cores, still disjoint
    none = [sharding.pin_rank(r, 4, sysfs=str(tmp_path / "missing"), allowed=allowed, apply=False) for r in range(4)]
    assert [(n["first_core"], n["last_core"]) for n in none] == [(0, 11), (12, 23), (24, 35), (36, 47)]
    assert all(n["numa_node"] is None for n in none)


def test_ordered_prefetch_keeps_order_and_bounds_lookahead():
    import threading
    import time

    from mrcc_amd.app.sharding import ordered_prefetch

    started, lock = [], threading.Lock()

    def work(i):
        with lock:
            started.append(i)
        time.sleep(0.002 * ((7 * i) % 5))  # uneven durations: completion order != submission order
        return i * i

    gen = ordered_prefetch(work, range(40), threads=4, lookahead=6)
    out = []
    for v in gen:
        out.append(v)
        assert len(started) <= len(out) + 6  # never more than `lookahead` frames ahead of the consumer
    assert out == [i * i for i in range(40)]

    def boom(i):
        if i == 3:
            raise ValueError("frame 3 is broken")
        return i

    got = []
    try:
        for v in ordered_prefetch(boom, range(10), threads=2):
            got.append(v)
        raise AssertionError("the source's exception must reach the consumer")
    except ValueError:
        assert got == [0, 1, 2]

"""Multi-GPU decomposition of the path: frames are independent, so ranks shard the frame sequence with
no data-path collective; the only collective of a run is one all_gather of a small fixed-size metrics record
(RCCL over xGMI on GPUs — backend "nccl" — or gloo on CPU in the tests)."""
import numpy as np
import torch


def frame_seeds_for_rank(total_frames, rank, world):
    """Cfg-4 rule: rank r takes frames r, r + world, r + 2 world, ..."""
    return list(range(rank, total_frames, world))


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        cpus.update(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_cpus(local_rank, sysfs="/sys", visible=None):
    """CPUs of the NUMA node the rank's GPU hangs off, read from sysfs WITHOUT touching HIP: render nodes in minor order
    are the GPUs in the runtime's enumeration order (`visible`: the HIP_/ROCR_VISIBLE_DEVICES list, applied first).
    Returns (cpu set, numa node) or (None, None) when the topology does not say (numa_node -1, no sysfs)."""
    import glob
    import os
    import re

    nodes = []
    for path in glob.glob(os.path.join(sysfs, "class", "drm", "renderD*")):
        m = re.search(r"renderD(\d+)$", path)
        try:
            with open(os.path.join(path, "device", "vendor")) as f:
                if f.read().strip().lower() != "0x1002":
                    continue
            with open(os.path.join(path, "device", "numa_node")) as f:
                nodes.append((int(m.group(1)), int(f.read().strip())))
        except (OSError, ValueError):
            continue
    nodes.sort()
    if visible:
        nodes = [nodes[i] for i in visible if i < len(nodes)]
    if not nodes:
        return None, None
    numa = nodes[local_rank % len(nodes)][1]
    if numa < 0:
        return None, None
    try:
        with open(os.path.join(sysfs, "devices", "system", "node", f"node{numa}", "cpulist")) as f:
            return _parse_cpulist(f.read()), numa
    except OSError:
        return None, None


def pin_rank(local_rank, world_local, sysfs="/sys", allowed=None, apply=True):
    """Restrict the calling process to the cores of its GPU's NUMA node (call BEFORE the first GPU call; nothing is
    re-exec'ed).  Ranks that share a node split its cores evenly; when sysfs does not name the node, the allowed set is
    split evenly over the local ranks instead (still: no two ranks fight for the same cores, no cross-socket migration).
    Returns a small record for the bench line."""
    import os

    allowed = set(os.sched_getaffinity(0)) if allowed is None else set(allowed)
    visible = None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v and all(t.strip().isdigit() for t in v.split(",")):
            visible = [int(t) for t in v.split(",")]
            break
    cpus, numa = gpu_numa_cpus(local_rank, sysfs, visible)
    how = "gpu numa node"
    if cpus is not None:
        # ranks whose GPUs share this node take turns over its cores
        peers = [r for r in range(world_local) if gpu_numa_cpus(r, sysfs, visible)[1] == numa]
        mine = sorted(cpus & allowed)
        if len(peers) > 1 and len(mine) >= len(peers):
            k = peers.index(local_rank)
            mine = mine[k * len(mine) // len(peers):(k + 1) * len(mine) // len(peers)]
    else:
        how, numa = "even split of the allowed cores (no NUMA information)", None
        ordered = sorted(allowed)
        mine = ordered[local_rank * len(ordered) // world_local:(local_rank + 1) * len(ordered) // world_local]
    if not mine:
        return {"pinned": False, "reason": "empty core set", "cores": len(allowed)}
    if apply:
        os.sched_setaffinity(0, mine)
    return {"pinned": bool(apply), "how": how, "numa_node": numa, "cores": len(mine), "first_core": mine[0],
            "last_core": mine[-1]}


_FIELDS = 4  # frames, elapsed, seed_sum, ncls


def gather_metrics(record, device="cuda", num_classes=3):
    """record: {"frames": int, "elapsed": float, "confusion": int64[num_classes, num_classes], "seed_sum": int,
    "elapsed_repeats": [float, ...] (optional: the timed region repeated R times; same R on every rank),
    "per_rank": {name: float} (optional: figures reported rank by rank - elapsed, host milliseconds per phase, pinned
    cores ...; the same names on every rank)}.
    ONE all_gather of a float64 vector per rank; returns the aggregate on every rank, `per_rank` as {name: [value of
    rank 0, rank 1, ...]} so that a slow or host-bound rank can be told from the line."""
    import torch.distributed as dist

    reps = [float(x) for x in record.get("elapsed_repeats", [])]
    names = sorted(record.get("per_rank", {}))
    extra = [float(record["per_rank"][k]) for k in names]
    vec = torch.zeros(_FIELDS + num_classes * num_classes + len(reps) + len(extra), dtype=torch.float64, device=device)
    vec[0] = record["frames"]
    vec[1] = record["elapsed"]
    vec[2] = record.get("seed_sum", 0)
    vec[3] = num_classes
    nc2 = num_classes * num_classes
    vec[_FIELDS:_FIELDS + nc2] = torch.as_tensor(np.asarray(record["confusion"], dtype=np.float64).reshape(-1))
    if reps:
        vec[_FIELDS + nc2:_FIELDS + nc2 + len(reps)] = torch.as_tensor(reps, dtype=torch.float64)
    if extra:
        vec[_FIELDS + nc2 + len(reps):] = torch.as_tensor(extra, dtype=torch.float64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        parts = [torch.zeros_like(vec) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, vec)
        allv = torch.stack(parts).cpu().numpy()
    else:
        allv = vec.cpu().numpy()[None]
    return {
        "frames": int(allv[:, 0].sum()),
        "per_rank_frames": [int(x) for x in allv[:, 0]],
        "elapsed_max": float(allv[:, 1].max()),
        "seed_sum": int(allv[:, 2].sum()),
        "confusion": allv[:, _FIELDS:_FIELDS + nc2].sum(axis=0).reshape(num_classes, num_classes).astype(np.int64),
        # per repeat: the slowest rank's time (the job's time for that repeat)
        "elapsed_repeats_max": [float(x) for x in allv[:, _FIELDS + nc2:_FIELDS + nc2 + len(reps)].max(axis=0)],
        "per_rank_elapsed": [float(x) for x in allv[:, 1]],
        "per_rank": {k: [float(x) for x in allv[:, _FIELDS + nc2 + len(reps) + i]] for i, k in enumerate(names)},
    }


def ordered_prefetch(fn, items, threads=2, lookahead=None):
    """Generator over fn(item) for item in items, IN ORDER, computed by `threads` background host threads at most
    `lookahead` items ahead of the consumer (the frame source of a rank: load / generate frame i + 1 .. i + lookahead
    while frame i is on the GPU; numpy releases the GIL in its heavy loops).  An exception in fn surfaces at the
    consumer when it reaches that item."""
    import collections
    import concurrent.futures as cf

    lookahead = lookahead or 2 * threads
    it = iter(items)
    pending = collections.deque()
    with cf.ThreadPoolExecutor(max_workers=threads) as pool:
        try:
            for item in it:
                pending.append(pool.submit(fn, item))
                if len(pending) >= lookahead:
                    yield pending.popleft().result()
            while pending:
                yield pending.popleft().result()
        finally:
            for f in pending:
                f.cancel()

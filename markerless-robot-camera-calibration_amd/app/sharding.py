"""Multi-GPU decomposition of the path: frames are independent, so ranks shard the frame sequence with
no data-path collective; the only collective of a run is one all_gather of a small fixed-size metrics record
(RCCL over xGMI on GPUs — backend "nccl" — or gloo on CPU in the tests)."""
import numpy as np
import torch


def frame_seeds_for_rank(total_frames, rank, world):
    """Cfg-4 rule: rank r takes frames r, r + world, r + 2 world, ..."""
    return list(range(rank, total_frames, world))


_FIELDS = 4  # frames, elapsed, seed_sum, ncls


def gather_metrics(record, device="cuda", num_classes=3):
    """record: {"frames": int, "elapsed": float, "confusion": int64[num_classes, num_classes], "seed_sum": int}.
    ONE all_gather of a float64 vector per rank; returns the aggregate on every rank."""
    import torch.distributed as dist

    vec = torch.zeros(_FIELDS + num_classes * num_classes, dtype=torch.float64, device=device)
    vec[0] = record["frames"]
    vec[1] = record["elapsed"]
    vec[2] = record.get("seed_sum", 0)
    vec[3] = num_classes
    vec[_FIELDS:] = torch.as_tensor(np.asarray(record["confusion"], dtype=np.float64).reshape(-1))
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
        "confusion": allv[:, _FIELDS:].sum(axis=0).reshape(num_classes, num_classes).astype(np.int64),
    }

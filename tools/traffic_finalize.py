"""gpurun_out/traffic/traffic_raw.json (tools/pmc_traffic.sh) -> profiles/<round>_traffic.json.
    python tools/traffic_finalize.py [raw.json] [round tag, e.g. r02] [commit the counters were collected at]

FETCH_SIZE / WRITE_SIZE are in KB; FETCH_SIZE under-reports this kernel's row gathers, so both are scaled by the factors
measured on the known-traffic launch (tools/traffic_calib.py), as the MI355X guide's HBM/rocprofv3 section prescribes.
Kernel names are folded to bench.py's naming: conv_fwd_kernel<TM, WAVES_N, NT[, fused Cin]>."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = json.load(open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out/traffic/traffic_raw.json")))


def fold(name):
    m = re.search(r"linear_narrow_kernel<(\d+)", name)
    if m:
        return f"linear_narrow_kernel<{m.group(1)}>"
    m = re.search(r"conv_fwd_dual_kernel<(\d+), (\d+), (\d+), (\d+)", name)
    if m:
        return f"conv_fwd_dual_kernel<{m.group(1)}, {m.group(2)}, {m.group(3)}, {m.group(4)}>"
    m = re.search(r"conv_fwd_kernel<([^>]*)>", name)
    a = [x.strip() for x in m.group(1).split(",")]
    cpo = int(a[4]) if len(a) > 4 else 0
    return f"conv_fwd_kernel<{a[0]}, {a[1]}, {a[2]}" + (f", fused {cpo}>" if cpo else ">")


known = raw["known"]
(cal_name, cal), = [(k, v) for k, v in raw["calibration"].items() if "FETCH_SIZE" in v][:1]
ff = known["read_bytes"] / (cal["FETCH_SIZE"] * 1024)
wf = known["write_bytes"] / (cal["WRITE_SIZE"] * 1024)
out = {
    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 4 --streams 1` (4 frames per sparse tensor: the default --group)",
    "calibration": {"kernel": fold(cal_name), "known_read_bytes": known["read_bytes"],
                    "known_write_bytes": known["write_bytes"], "FETCH_SIZE_KB": cal["FETCH_SIZE"],
                    "WRITE_SIZE_KB": cal["WRITE_SIZE"], "fetch_factor": round(ff, 4), "write_factor": round(wf, 4),
                    "note": "dense 1x1 launch reading 2.0 GB / writing 1.0 GB exactly once (tools/traffic_calib.py); "
                            "FETCH_SIZE * 1024 under-reports the 16 B/lane row gathers by fetch_factor, WRITE_SIZE is exact"},
    "kernels": {},
}
acc = {}
for name, b in raw["bench"].items():
    k = fold(name)
    n = b["FETCH_SIZE"]["launches"]
    d = acc.setdefault(k, [0, 0.0, 0.0])
    d[0] += n
    d[1] += b["FETCH_SIZE"]["per_launch"] * n * 1024 * ff
    d[2] += b["WRITE_SIZE"]["per_launch"] * n * 1024 * wf
for k, (n, r, w) in sorted(acc.items()):
    out["kernels"][k] = {"launches": n, "read_GB_per_launch": round(r / n / 1e9, 4),
                         "write_GB_per_launch": round(w / n / 1e9, 4), "traffic_GB_per_launch": round((r + w) / n / 1e9, 4)}
# the same per (kernel instance, grid size): one instance serves several layer shapes (the dual-body kernel runs pyramid
# levels 0 and 1); grids within 5 % of each other (the frames of the pool differ by < 1 % in voxel count) are one group
out["kernels_by_grid"] = {}
for name, grids in raw.get("bench_by_grid", {}).items():
    k = fold(name)
    groups = out["kernels_by_grid"].setdefault(k, [])
    for wgs, b in sorted(grids.items(), key=lambda kv: int(kv[0])):
        if "FETCH_SIZE" not in b or "WRITE_SIZE" not in b:
            continue
        n = b["FETCH_SIZE"]["launches"]
        r = b["FETCH_SIZE"]["per_launch"] * n * 1024 * ff
        w = b["WRITE_SIZE"]["per_launch"] * n * 1024 * wf
        for g in groups:
            if abs(int(wgs) - g["_wgs"] / g["launches"]) <= 0.05 * g["_wgs"] / g["launches"]:
                g["launches"] += n; g["_wgs"] += int(wgs) * n; g["_r"] += r; g["_w"] += w
                break
        else:
            groups.append({"launches": n, "_wgs": int(wgs) * n, "_r": r, "_w": w})
for k, groups in out["kernels_by_grid"].items():
    for g in groups:
        n = g["launches"]
        g["grid_workgroups"] = round(g.pop("_wgs") / n)
        r, w = g.pop("_r"), g.pop("_w")
        g["read_GB_per_launch"] = round(r / n / 1e9, 4)
        g["write_GB_per_launch"] = round(w / n / 1e9, 4)
        g["traffic_GB_per_launch"] = round((r + w) / n / 1e9, 4)
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
out["commit"] = sys.argv[3] if len(sys.argv) > 3 else "unknown"
out["source"] = out["source"].replace("bench.py --steps", "bench.py --no-cpu-baseline --no-extras --steps")
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

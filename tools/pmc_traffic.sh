#!/bin/bash
# HBM traffic of the dominant conv kernel via PMC (separate passes, no tracing flags), calibrated on a known launch.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
OUT=gpurun_out/traffic; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE TCC_EA0_RDREQ_sum; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/calib_$c -- python tools/traffic_calib.py > $OUT/calib_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $OUT/bench_$c -- python bench.py --no-cpu-baseline --no-extras --steps 8 --warmup 4 --streams 1 > $OUT/bench_$c.log 2>&1
done
python - <<PY
import csv, glob, json, collections
def per_kernel(prefix, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"$OUT/{prefix}_{counter}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and ("conv_fwd_kernel" in r["Kernel_Name"] or "conv_fwd_dual_kernel" in r["Kernel_Name"] or "linear_narrow_kernel" in r["Kernel_Name"]):
                acc[r["Kernel_Name"].split("(")[0].replace("void sv::", "")].append(float(r["Counter_Value"]))
    return acc
V = 1_300_000
known_read = V * 384 * 4 + 384 * 192 * 4
known_write = V * 192 * 4
def per_kernel_grid(prefix, counter):
    """as per_kernel, keyed by (kernel, workgroups of the launch): one instance serves several layer shapes"""
    acc = collections.defaultdict(list)
    for f in glob.glob(f"$OUT/{prefix}_{counter}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and ("conv_fwd_kernel" in r["Kernel_Name"] or "conv_fwd_dual_kernel" in r["Kernel_Name"]):
                wgs = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
                acc[(r["Kernel_Name"].split("(")[0].replace("void sv::", ""), wgs)].append(float(r["Counter_Value"]))
    return acc
res = {"calibration": {}, "bench": {}, "bench_by_grid": {}}
for c in ("FETCH_SIZE", "WRITE_SIZE", "TCC_EA0_RDREQ_sum"):
    cal = per_kernel("calib", c)
    for k, v in cal.items():
        res["calibration"].setdefault(k, {})[c] = sum(v) / len(v)
    b = per_kernel("bench", c)
    for k, v in b.items():
        res["bench"].setdefault(k, {})[c] = {"per_launch": sum(v) / len(v), "launches": len(v)}
    for (k, wgs), v in per_kernel_grid("bench", c).items():
        res["bench_by_grid"].setdefault(k, {}).setdefault(str(wgs), {})[c] = {"per_launch": sum(v) / len(v), "launches": len(v)}
res["known"] = {"read_bytes": known_read, "write_bytes": known_write}
print(json.dumps(res, indent=1))
json.dump(res, open("$OUT/traffic_raw.json", "w"), indent=1)
PY

"""Break a rocprofv3 kernel trace down by conv kernel instance AND grid size (workgroups): the default bench command
runs the one-frame and the batched x4 configuration with the same kernels, and the grid tells them apart.
    python tools/kernel_by_grid.py <..._kernel_trace.csv>"""
import collections
import csv
import re
import sys

acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"]
    if "conv_" not in name and "linear_narrow" not in name:
        continue
    name = re.sub(r"\(.*", "", name).replace("void sv::", "")
    wgs = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    acc[(name, wgs)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(f"{'kernel instance':60s} {'grid':>6s} {'calls':>6s} {'avg us':>10s} {'total ms':>10s}")
for (name, wgs), v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print(f"{name:60s} {wgs:6d} {len(v):6d} {sum(v) / len(v):10.1f} {sum(v) / 1e3:10.2f}")

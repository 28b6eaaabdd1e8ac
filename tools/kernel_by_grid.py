"""Break a rocprofv3 kernel trace down by conv kernel instance and grid size (workgroups), and split a group whose
durations are BIMODAL: the default bench command runs several configurations with the same kernels - the grid tells
pyramid levels / batch sizes apart - but the level-0 `convtr7` launch (kernel volume 8, ~0.2 ms) shares its grid with the
dominant kernel-volume-27 layers (~2.3 ms), and the trace carries neither kernel arguments nor the dynamic LDS size, so
the two are told apart by the one thing that differs: a > 2x gap in the sorted durations.
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

rows = []
for (name, wgs), v in acc.items():
    v = sorted(v)
    gaps = [(v[i + 1] / max(v[i], 1e-9), i) for i in range(len(v) - 1)]
    ratio, cut = max(gaps) if gaps else (1.0, 0)
    if ratio > 2.0 and min(cut + 1, len(v) - cut - 1) >= 2:  # two populations of launches behind one (instance, grid)
        rows.append((name, wgs, "short mode (smaller kernel volume)", v[:cut + 1]))
        rows.append((name, wgs, "long mode", v[cut + 1:]))
    else:
        rows.append((name, wgs, "", v))
print(f"{'kernel instance':60s} {'grid':>6s} {'calls':>6s} {'avg us':>10s} {'total ms':>10s}  note")
for name, wgs, note, v in sorted(rows, key=lambda r: -sum(r[3]))[:48]:
    print(f"{name:60s} {wgs:6d} {len(v):6d} {sum(v) / len(v):10.1f} {sum(v) / 1e3:10.2f}  {note}")

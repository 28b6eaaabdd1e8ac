"""Per-step phase breakdown from a trace written by the diagnostic build of tools/conv_phases_build.py."""
import sys, numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64)
i = 0; L = []
while i < len(raw):
    n = int(raw[i + 1]); hdr = raw[i:i + 8].astype(np.int64); i += 8
    L.append((hdr, raw[i:i + 4 * n].reshape(n, 4))); i += 4 * n
hdr, t = L[-1]
steps = (t[:, 3] & np.uint64(0xffffff)).astype(np.float64); bar = (t[:, 3] >> np.uint64(24)).astype(np.float64)
m = steps > 20
print(f"instance <{hdr[2]},{hdr[3]},{hdr[4]}> grid {hdr[1]}; WGs with >20 steps: {m.sum()}")
for name, v in (("mfma loop", t[:, 0]), ("store_a (wait gathers + ds_write)", t[:, 1]), ("gather issue", t[:, 2]), ("advance + barrier", bar)):
    x = v.astype(np.float64)[m] / steps[m]
    print(f"  {name:36s} per step: mean {x.mean():8.1f}  p10 {np.percentile(x,10):8.1f}  p90 {np.percentile(x,90):8.1f} ticks")
tot = (t[:, 0].astype(np.float64) + t[:, 1].astype(np.float64) + t[:, 2].astype(np.float64) + bar)[m] / steps[m]
print(f"  total per step {tot.mean():.1f} ticks")

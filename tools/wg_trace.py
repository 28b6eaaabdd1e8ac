"""Analyse a per-workgroup trace written by SV_CONV_TRACE=<file> (csrc/sv_conv.hip): residency over time, per-CU
utilisation, workgroup durations versus start time.
    SV_CONV_TRACE=gpurun_out/wg.bin python tools/conv_microbench.py --level 0 --iters 1 ; python tools/wg_trace.py gpurun_out/wg.bin"""
import sys
import numpy as np

raw = np.fromfile(sys.argv[1], dtype=np.uint64)
launches = []
i = 0
while i < len(raw):
    assert raw[i] == 0x5356545243, "bad header"
    n = int(raw[i + 1]); hdr = raw[i:i + 8].astype(np.int64); i += 8
    launches.append((hdr, raw[i:i + 4 * n].reshape(n, 4))); i += 4 * n
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(launches) - 1
hdr, t = launches[which]
print(f"{len(launches)} launches in file; analysing #{which}: grid {hdr[1]} instance <{hdr[2]},{hdr[3]},{hdr[4]}> ny {hdr[5]} K {hdr[6]} Cin {hdr[7]}")
t0 = t[:, 0].astype(np.int64); t1 = t[:, 1].astype(np.int64)
base = t0.min(); t0 = (t0 - base) / 100.0; t1 = (t1 - base) / 100.0   # 100 MHz -> microseconds
hw = (t[:, 2] & np.uint64(0xffffffff)).astype(np.int64); xcc = (t[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7
cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
steps = (t[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
pro = ((t[:, 3] >> np.uint64(32)) & np.uint64(0xffff)).astype(np.float64) * 16 / 100.0   # us: start -> first pipeline step
epi = ((t[:, 3] >> np.uint64(48)) & np.uint64(0xffff)).astype(np.float64) * 16 / 100.0   # us: epilogue start -> end
if pro.max() > 0:
    print(f"prologue (start -> first step) us: median {np.median(pro):.1f} mean {pro.mean():.1f} p90 {np.percentile(pro, 90):.1f}; "
          f"epilogue us: median {np.median(epi):.1f} mean {epi.mean():.1f} p90 {np.percentile(epi, 90):.1f}; "
          f"share of workgroup time: {(pro.sum() + epi.sum()) / ((t[:, 1].astype(np.int64) - t[:, 0].astype(np.int64)).sum() / 100.0) * 100:.1f} %")
dur = t1 - t0
T = t1.max()
print(f"kernel span {T:.0f} us; workgroups {len(t)}; distinct CUs {len(np.unique(cuid))}; WG duration min/median/max {dur.min():.0f}/{np.median(dur):.0f}/{dur.max():.0f} us")
print(f"us per step: median {np.median(dur[steps > 0] / steps[steps > 0]):.2f}")
# residency over time
edges = np.linspace(0, T, 41)
ev = np.concatenate([np.stack([t0, np.ones_like(t0)], 1), np.stack([t1, -np.ones_like(t1)], 1)])
ev = ev[np.argsort(ev[:, 0])]
res_t = ev[:, 0]; res_n = np.cumsum(ev[:, 1])
print("time(us)  resident WGs  | started in slice | mean us/step of WGs started in slice")
for a, b in zip(edges[:-1], edges[1:]):
    m = (res_t >= a) & (res_t < b)
    started = (t0 >= a) & (t0 < b)
    ups = np.mean(dur[started & (steps > 0)] / steps[started & (steps > 0)]) if (started & (steps > 0)).any() else float("nan")
    print(f"{a:8.0f}  {res_n[m].mean() if m.any() else float('nan'):8.0f}      | {started.sum():6d}     | {ups:6.2f}")
# per-CU finish times
last = np.array([t1[cuid == c].max() for c in np.unique(cuid)])
print(f"per-CU last finish: min {last.min():.0f} median {np.median(last):.0f} max {last.max():.0f} us -> idle tail fraction {(T - last).mean() / T:.3f}")
busy = np.array([dur[cuid == c].sum() for c in np.unique(cuid)])
print(f"sum of WG durations per CU / span: mean {busy.mean() / T:.2f} (4 = fully resident)")

"""Timeline analysis of a rocprofv3 --kernel-trace CSV of bench.py: for the last `--frames` frames of the run, how busy
each queue was, how much kernels of different queues overlapped, and what ran while the big convolutions ran.
    python tools/timeline.py gpurun_out/<dir>/<pid>_kernel_trace.csv [--last-ms 150]"""
import argparse
import collections
import csv
import re

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--last-ms", type=float, default=150.0, help="analyse the last N ms of GPU activity (the timed region)")
args = ap.parse_args()

rows = []
for r in csv.DictReader(open(args.csv)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"],
                 int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)))
rows.sort()
t_end = max(r[1] for r in rows)
t0 = t_end - int(args.last_ms * 1e6)
win = [r for r in rows if r[0] >= t0]
span = (t_end - min(r[0] for r in win)) / 1e6


def short(n):
    s = n
    for pat in (r"conv_fwd_kernel<[^>]*>", r"conv_first_layer_kernel", r"conv_first_mfma_kernel", r"linear_narrow_kernel", r"sv::(\w+)",
                r"rocprim::detail::(\w+)", r"at::native::(\w+)", r"(\w+_kernel\w*)"):
        m = re.search(pat, n)
        if m:
            s = m.group(m.lastindex or 0)
            break
    s = re.sub(r"conv_fwd_kernel<(\d+), (\d+), (\d+), (true|false), (\d+), (true|false), (true|false)>",
               lambda k: f"conv<{k.group(1)},{k.group(2)},{k.group(3)}{',c' + k.group(5) if k.group(5) != '0' else ''}"
                         f"{',ring' if k.group(6) == 'true' else ''}{',full' if k.group(7) == 'true' else ''}>", s)
    return s[:48]


print(f"window {span:.1f} ms, {len(win)} kernels")
byq = collections.defaultdict(list)
for r in win:
    byq[r[2]].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(e - s for s, e, *_ in rs) / 1e6
    names = collections.Counter(short(r[3]) for r in rs).most_common(3)
    print(f"queue {q}: {len(rs):5d} kernels, busy {busy:7.2f} ms ({100 * busy / span:5.1f} % of window), top: {names}")
# union / concurrency profile
ev = []
for s, e, *_ in win:
    ev.append((s, 1))
    ev.append((e, -1))
ev.sort()
depth, last, hist = 0, ev[0][0], collections.Counter()
for t, d in ev:
    hist[depth] += t - last
    last = t
    depth += d
tot = sum(hist.values())
print("concurrent kernels -> share of window:", {k: f"{100 * v / tot:.1f}%" for k, v in sorted(hist.items())})
# per kernel name: count, total ms, avg us
agg = collections.defaultdict(lambda: [0, 0])
for s, e, q, n, wg in win:
    a = agg[short(n)]
    a[0] += 1
    a[1] += e - s
print(f"{'kernel':50s} {'n':>6s} {'total ms':>9s} {'avg us':>9s}")
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
    print(f"{n:50s} {c:6d} {t / 1e6:9.2f} {t / c / 1e3:9.1f}")
# big-kernel view: what fraction of each big conv's duration was shared with another big conv
big = [r for r in win if "conv_fwd_kernel<64" in r[3] or "conv_fwd_kernel<128" in r[3]]
big_busy = sum(e - s for s, e, *_ in big) / 1e6
ov = 0
for i, (s, e, q, n, wg) in enumerate(big):
    for s2, e2, q2, n2, wg2 in big[i + 1:]:
        if s2 >= e:
            break
        if q2 != q:
            ov += max(0, min(e, e2) - s2)
print(f"big convs (64/128-row tiles): {len(big)} launches, {big_busy:.2f} ms summed, {ov / 1e6:.2f} ms of pairwise overlap "
      f"between queues")

"""Condenses a rocprofv3 --kernel-trace CSV of a multi-stream bench run: mean duration per kernel
under overlap, fraction of time with 0, 1, 2, ... kernels in flight, and a sample of the timeline.
usage: python tools/overlap_trace.py <kernel_trace.csv> [n_timeline_rows]"""
import collections
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_\w+)", r["Kernel_Name"])
    name = m.group(1) if m else "other:" + r["Kernel_Name"][:24]
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r["Queue_Id"]))
rows.sort()
# the timed multi-stream region: library kernels on queues other than the first one
ms = [r for r in rows if r[3] != rows[0][3] and r[2].startswith("k_")]
t0, t1 = min(r[0] for r in ms), max(r[1] for r in ms)
a, b = t0 + 0.3 * (t1 - t0), t0 + 0.9 * (t1 - t0)
reg = [r for r in rows if r[0] >= a and r[1] <= b]
dur = collections.defaultdict(list)
for s, e, k, q in reg:
    dur[k].append(e - s)
print("window %.2f ms" % ((b - a) / 1e6))
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print("%-28s n=%4d mean %7.1f us total %.3f ms" % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e6))
ev = []
for s, e, k, q in reg:
    ev += [(s, 1), (e, -1)]
ev.sort()
depth, last, hist = 0, a, collections.Counter()
for t, d in ev:
    hist[depth] += t - last
    last = t
    depth += d
tot = sum(hist.values())
print("kernels in flight:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
for r in reg[:int(sys.argv[2]) if len(sys.argv) > 2 else 0]:
    print(r[3], "%-20s start %8.1f us  dur %7.1f us" % (r[2], (r[0] - a) / 1e3, (r[1] - r[0]) / 1e3))

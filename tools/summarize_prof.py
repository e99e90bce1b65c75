#!/usr/bin/env python3
"""Condenses the rocprofv3 outputs of tools/profile_gpu.sh into small per-kernel tables:
  <out>/kernel_stats.csv        (copied from the --kernel-trace --stats pass, our kernels only)
  <out>/pmc_per_kernel.csv      (mean counter value per launch for every k_* kernel)
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; FETCH_SIZE additionally counts 64 B per
128-B request on gfx950 for wide streaming reads and is doubled in the hbm_read_bytes column
(uncalibrated for narrower accesses: an upper bound there).  The last column is the fingerprint of the
library's sources (openvslam_amd._lib.source_fingerprint): bench.py marks the counters as stale when the
library it runs was built from other sources."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from openvslam_amd._lib import source_fingerprint  # noqa: E402


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = name.split("(")[0]
    return name.replace("void ", "").strip()


def main():
    out = Path(sys.argv[1])
    rows = []
    stats = out / "trace" / "trace_kernel_stats.csv"
    if stats.exists():
        with open(stats) as f:
            r = csv.DictReader(f)
            for row in r:
                n = short(row["Name"])
                if n.startswith("k_"):
                    rows.append([n, row["Calls"], row["TotalDurationNs"], row["AverageNs"], row["MinNs"], row["MaxNs"], row["StdDev"]])
        with open(out / "kernel_stats.csv", "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["kernel", "calls", "total_ns", "average_ns", "min_ns", "max_ns", "stddev"])
            w.writerows(rows)
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    meta = {}
    for p in ("sq1", "sq2", "sq3", "fetch", "write"):
        fn = out / p / (p + "_counter_collection.csv")
        if not fn.exists():
            continue
        with open(fn) as f:
            r = csv.DictReader(f)
            for row in r:
                n = short(row["Kernel_Name"])
                if not n.startswith("k_"):
                    continue
                a = acc[n][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
                meta[n] = (row.get("VGPR_Count", ""), row.get("Accum_VGPR_Count", ""), row.get("SGPR_Count", ""),
                           row.get("LDS_Block_Size", ""), row.get("Scratch_Size", ""), row.get("Workgroup_Size", ""), row.get("Grid_Size", ""))
    counters = sorted({c for k in acc.values() for c in k})
    with open(out / "pmc_per_kernel.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches_seen", "vgpr", "agpr", "sgpr", "lds_bytes", "scratch", "wg_size", "grid_size(last)"]
                   + counters + ["hbm_read_bytes(=2*FETCH_SIZE*1024)", "hbm_write_bytes(=WRITE_SIZE*1024)", "csrc_sha256"])
        fingerprint = source_fingerprint()  # the sources of the library these counters were taken from
        for n in sorted(acc):
            k = acc[n]
            launches = max(v[1] for v in k.values())
            vals = [("%.1f" % (k[c][0] / k[c][1])) if c in k and k[c][1] else "" for c in counters]
            rd = 2 * 1024 * k["FETCH_SIZE"][0] / k["FETCH_SIZE"][1] if "FETCH_SIZE" in k and k["FETCH_SIZE"][1] else ""
            wr = 1024 * k["WRITE_SIZE"][0] / k["WRITE_SIZE"][1] if "WRITE_SIZE" in k and k["WRITE_SIZE"][1] else ""
            w.writerow([n, launches, *meta[n]] + vals + [rd, wr, fingerprint])
    print(open(out / "pmc_per_kernel.csv").read())
    if (out / "kernel_stats.csv").exists():
        print(open(out / "kernel_stats.csv").read())


if __name__ == "__main__":
    main()

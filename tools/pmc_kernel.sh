#!/bin/bash
# usage: tools/pmc_kernel.sh <kernel substring> "<counters>" -- <python script and args>
# One rocprofv3 counter pass over a python command; prints the per-launch mean of every counter for
# the kernels whose name contains the substring.
set -e
KERNEL="$1"; COUNTERS="$2"; shift 3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_k
timeout -s KILL 120 rocprofv3 --pmc $COUNTERS -d /tmp/pmc_k -o p --output-format csv -- python3 "$@" > /dev/null 2>&1
python3 - "$KERNEL" <<'PY'
import csv, glob, collections, sys
f = glob.glob("/tmp/pmc_k/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"]
    if sys.argv[1] in k:
        acc[k[:50]][r["Counter_Name"]] += float(r["Counter_Value"])
        n[k[:50]].add(r["Dispatch_Id"])
for k in acc:
    print(k, "launches", len(n[k]), {c: round(v / len(n[k])) for c, v in sorted(acc[k].items())})
PY

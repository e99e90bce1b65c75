#!/bin/bash
# Runs on the GPU box: FETCH_SIZE and WRITE_SIZE passes over tools/calib/fetch_calib, prints
# counter value per kernel (KiB) next to the 524288 KiB each kernel moves.
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/calib
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/f" -o f -- $REPO/tools/calib/fetch_calib > "$OUT/f.log" 2>&1 || { tail -5 "$OUT/f.log"; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/w" -o w -- $REPO/tools/calib/fetch_calib > "$OUT/w.log" 2>&1 || { tail -5 "$OUT/w.log"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, sys
out = sys.argv[1]
for tag in ("f", "w"):
    with open(f"{out}/{tag}/{tag}_counter_collection.csv") as fh:
        for row in csv.DictReader(fh):
            n = row["Kernel_Name"].split("(")[0]
            if n.startswith(("read_", "write_")):
                print(tag, n, row["Counter_Name"], row["Counter_Value"], "KiB; moved 524288 KiB; ratio %.3f" % (float(row["Counter_Value"]) / 524288.0))
PY
rm -rf "$OUT/f" "$OUT/w"

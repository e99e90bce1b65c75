#!/bin/bash
# Runs on the GPU box (through gpurun): kernel trace + three counter passes of the headline bench.
# Each pass is its own rocprofv3 run (counters are never combined with API traces), the program
# follows `--` directly.  Outputs land under gpurun_out/prof_<tag>/.
#   usage: tools/profile_gpu.sh <tag>
set -u
TAG=${1:-r1}
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
# every kernel alone on the device: no second stream inside the extractor
export OVS_SERIAL_EXTRACT=1
cd /tmp
BENCH="python3 $REPO/bench.py --steps 3 --warmup 1 --no-extras --batch 255 --streams 1"

timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- $BENCH > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }
echo "trace pass done"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d "$OUT/sq1" -o sq1 -- $BENCH > "$OUT/sq1.log" 2>&1 || { echo "sq1 pass failed"; tail -5 "$OUT/sq1.log"; exit 1; }
echo "sq1 pass done"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT --output-format csv -d "$OUT/sq2" -o sq2 -- $BENCH > "$OUT/sq2.log" 2>&1 || { echo "sq2 pass failed"; tail -5 "$OUT/sq2.log"; exit 1; }
echo "sq2 pass done"
timeout -s KILL 240 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_I8 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM --output-format csv -d "$OUT/sq3" -o sq3 -- $BENCH > "$OUT/sq3.log" 2>&1 || { echo "sq3 pass failed (optional)"; tail -3 "$OUT/sq3.log"; }
echo "sq3 pass done"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -o fetch -- $BENCH > "$OUT/fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$OUT/fetch.log"; exit 1; }
echo "fetch pass done"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -o write -- $BENCH > "$OUT/write.log" 2>&1 || { echo "write pass failed"; tail -5 "$OUT/write.log"; exit 1; }
echo "write pass done"
python3 $REPO/tools/summarize_prof.py "$OUT" > "$OUT/summary.txt" 2>&1; tail -5 "$OUT/summary.txt"
grep -h "^{" "$OUT/trace.log" > "$OUT/bench_line.json"
rm -rf "$OUT/trace" "$OUT/sq1" "$OUT/sq2" "$OUT/sq3" "$OUT/fetch" "$OUT/write"
ls -la "$OUT"

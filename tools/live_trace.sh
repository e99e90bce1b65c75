# usage (GPU box): bash tools/live_trace.sh   kernel durations of the single-frame live path
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/live_trace
PYTHONPATH=$GRAFT_REPO_ROOT timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d gpurun_out/live_trace -o lt -- python3 tools/bench_extra.py live > gpurun_out/live_trace.log 2>&1 || { tail -5 gpurun_out/live_trace.log; exit 1; }
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/live_trace/**/*kernel_stats.csv',recursive=True)[0]
tot=0
for r in csv.DictReader(open(f)):
    print(r['Name'][:60], r['Calls'], r['AverageNs'])
PY
grep "^{" gpurun_out/live_trace.log | cut -c1-400

# usage: bash tools/ab_live.sh <variant> ...   (on the GPU box) single-frame latency (tools/bench_extra.py live) and the
# kernel durations of that path for each openvslam_amd/lib/ab_<variant>.so in turn
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
for v in "$@"; do
  cp openvslam_amd/lib/ab_$v.so openvslam_amd/lib/libovs_hip.so
  timeout -k 10 120 python3 tools/bench_extra.py live > gpurun_out/ab_live_$v.json 2> gpurun_out/ab_live_$v.err || { echo "variant $v failed"; tail -5 gpurun_out/ab_live_$v.err; exit 1; }
  echo "$v $(grep '^{' gpurun_out/ab_live_$v.json | cut -c1-330)"
  rm -rf gpurun_out/ab_live_trace_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_live_trace_$v -o lt -- python3 tools/bench_extra.py live > gpurun_out/ab_live_trace_$v.log 2>&1 || { tail -5 gpurun_out/ab_live_trace_$v.log; exit 1; }
  python3 - $v <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/ab_live_trace_%s/**/*kernel_stats.csv'%sys.argv[1],recursive=True)[0]
print(sys.argv[1], ' '.join('%s=%.1f'%(r['Name'].split('::')[1].split('(')[0].split('<')[0], float(r['AverageNs'])/1e3) for r in csv.DictReader(open(f)) if 'k_' in r['Name'] and 'at::' not in r['Name']))
PY
done

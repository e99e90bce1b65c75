# usage: bash tools/ab_run.sh <variant> ...   (on the GPU box) times the extractor kernels of each
# openvslam_amd/lib/ab_<variant>.so in turn, in one process each, on the same box
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export OVS_SERIAL_EXTRACT=1
for v in "$@"; do
  cp openvslam_amd/lib/ab_$v.so openvslam_amd/lib/libovs_hip.so
  rm -rf gpurun_out/ab_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$v -o fp -- python3 bench.py --steps 3 --warmup 1 --no-extras --no-cpu-baseline --batch 255 --streams 1 > gpurun_out/ab_$v.log 2>&1 || { echo "variant $v failed"; tail -5 gpurun_out/ab_$v.log; exit 1; }
  python3 - $v <<'PY'
import csv,glob,sys
f=glob.glob('gpurun_out/ab_%s/**/*kernel_stats.csv'%sys.argv[1],recursive=True)[0]
print(sys.argv[1], ' '.join('%s=%.1f'%(r['Name'].split('::')[1].split('(')[0].split('<')[0], float(r['AverageNs'])/1e3) for r in csv.DictReader(open(f)) if 'k_' in r['Name'] and 'at::' not in r['Name']))
PY
done

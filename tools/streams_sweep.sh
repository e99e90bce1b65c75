# usage (GPU box): bash tools/streams_sweep.sh   the KITTI step with different numbers of streams / sub-batch sizes
cd $GRAFT_REPO_ROOT
for cfg in "2 510" "3 765" "4 1020" "6 1530" "8 2040" "2 1020" "3 1020" "5 1020" "6 1020" "10 1020" "12 1020"; do set -- $cfg
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --streams $1 --batch $2 > gpurun_out/st_$1_$2.log 2>&1 || { echo fail; tail -3 gpurun_out/st_$1_$2.log; exit 1; }
  python3 - "$1" "$2" <<'PY'
import json,sys
l=[x for x in open('gpurun_out/st_%s_%s.log'%(sys.argv[1],sys.argv[2])) if x.startswith('{')][-1]
d=json.loads(l); print('streams',sys.argv[1],'batch',sys.argv[2],'frames/s %.0f ms/step %.3f'%(d['value'],d['ms_per_step']))
PY
done

# usage: bash tools/ab_step.sh <variant> ...   (on the GPU box) the default four-stream bench step with each
# openvslam_amd/lib/ab_<variant>.so in turn (frames/s and ms per 1020-frame step), same box, one process each
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in "$@"; do
  cp openvslam_amd/lib/ab_$v.so openvslam_amd/lib/libovs_hip.so
  timeout -k 10 200 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > gpurun_out/abs_$v.log 2>&1 || { echo "variant $v failed"; tail -5 gpurun_out/abs_$v.log; exit 1; }
  python3 - $v <<'PY'
import json,sys
l=[x for x in open('gpurun_out/abs_%s.log'%sys.argv[1]) if x.startswith('{')][-1]
d=json.loads(l)
print(sys.argv[1], 'frames/s %.0f  ms/step %.3f'%(d['value'], d['ms_per_step']))
PY
done

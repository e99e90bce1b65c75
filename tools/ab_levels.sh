# usage: bash tools/ab_levels.sh <variant> ...  per-launch durations of k_resize_rows (one per pyramid level) from the kernel trace of ab_run.sh
cd $GRAFT_REPO_ROOT
for v in "$@"; do python3 - $v <<'PY'
import csv,glob,sys,collections
f=glob.glob('gpurun_out/ab_%s/**/*kernel_trace.csv'%sys.argv[1],recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'k_resize_rows' in r['Kernel_Name']]
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows]
n=len(d)//7
lv=[sum(d[i*7+l] for i in range(n))/n for l in range(7)]
print(sys.argv[1], 'resize per level us:', ' '.join('%.1f'%x for x in lv), 'sum %.1f'%sum(lv))
PY
done

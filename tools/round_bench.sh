#!/bin/bash
# Runs on the GPU box: the default bench line, the two secondary workloads and tools/bench_extra.py,
# each into gpurun_out/<tag>_*.json.   usage: tools/round_bench.sh <tag>
TAG=${1:-r2}
cd $GRAFT_REPO_ROOT
export PYTHONPATH=$GRAFT_REPO_ROOT
timeout -k 10 600 python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || { echo "bench failed"; tail -5 gpurun_out/${TAG}_bench.err; exit 1; }
echo "default bench done"
timeout -k 10 600 python3 bench.py --workload equirect > gpurun_out/${TAG}_equirect.json 2> gpurun_out/${TAG}_equirect.err || { echo "equirect failed"; tail -5 gpurun_out/${TAG}_equirect.err; exit 1; }
echo "equirect done"
timeout -k 10 600 python3 bench.py --workload knn > gpurun_out/${TAG}_knn.json 2> gpurun_out/${TAG}_knn.err || { echo "knn failed"; tail -5 gpurun_out/${TAG}_knn.err; exit 1; }
echo "knn done"
timeout -k 10 600 python3 tools/bench_extra.py all > gpurun_out/${TAG}_extra.json 2> gpurun_out/${TAG}_extra.err || { echo "extra failed"; tail -5 gpurun_out/${TAG}_extra.err; exit 1; }
echo "extra done"

#!/bin/bash
# One GPU call for the end of the round: the GPU test suite, matcher parity once more with the single-wave resolve
# kernel forced, B = 1 match profiles for both kernel choices, the default bench line, the rocprofv3 passes.  Every step under
# its own time limit; a failed step ends the script.
set -u
cd "$GRAFT_REPO_ROOT"
export PYTHONPATH="$GRAFT_REPO_ROOT"
timeout -k 10 600 python3 -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread -p no:cacheprovider > gpurun_out/r3_head_tests.log 2>&1 || { echo "GPU suite failed"; tail -20 gpurun_out/r3_head_tests.log; exit 1; }
tail -1 gpurun_out/r3_head_tests.log
OVS_RESOLVE=wave timeout -k 10 300 python3 -u -m pytest tests/test_gpu_projection.py tests/test_gpu_robust_bow.py tests/test_gpu_threads.py -m gpu -x -q --timeout 120 --timeout-method thread -p no:cacheprovider > gpurun_out/r3_wave_tests.log 2>&1 || { echo "single-wave tests failed"; tail -20 gpurun_out/r3_wave_tests.log; exit 1; }
tail -1 gpurun_out/r3_wave_tests.log
timeout -k 10 120 python3 tools/live_match_profile.py > gpurun_out/r3_live_default.json 2> gpurun_out/r3_live_default.err || { echo "live default failed"; tail -5 gpurun_out/r3_live_default.err; exit 1; }
OVS_RESOLVE=wave timeout -k 10 120 python3 tools/live_match_profile.py > gpurun_out/r3_live_wave.json 2> gpurun_out/r3_live_wave.err || { echo "live wave failed"; tail -5 gpurun_out/r3_live_wave.err; exit 1; }
echo "live profiles done"
timeout -k 10 600 python3 bench.py > gpurun_out/r3_bench_head.json 2> gpurun_out/r3_bench_head.err || { echo "bench failed"; tail -5 gpurun_out/r3_bench_head.err; exit 1; }
echo "bench done"
bash tools/profile_gpu.sh r3 || exit 1

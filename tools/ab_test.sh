# usage: bash tools/ab_test.sh <variant>   extractor parity tests with openvslam_amd/lib/ab_<variant>.so
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
cp openvslam_amd/lib/ab_$1.so openvslam_amd/lib/libovs_hip.so
timeout -k 10 300 python -u -m pytest tests/test_gpu_extractor.py tests/test_gpu_batch.py tests/test_gpu_reference_cases.py -x -q --timeout 120 --timeout-method thread 2>&1 | tail -5

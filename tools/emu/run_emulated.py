#!/usr/bin/env python3
"""Runs test files written for the GPU against the library built for the host (no GPU):

    tools/emu/run_emulated.py tests/test_gpu_projection.py [pytest arguments]
    tools/emu/run_emulated.py tests -m gpu

The `gpu` marker is only a selection label; the tests call through the same C ABI.  Tests that are sized for
the device (tens of thousands of frames, a 20 M row database) or that link the device library into a C++
program are left out unless --full-size is given.  Environment: EMU_SHUFFLE=<seed> / EMU_REVERSE=1 (thread
order), EMU_STRICT_LDS=1, OVS_EMU_UNITS=a.hip,b.hip, OVS_RESOLVE / OVS_GROUP_LIVE as for the library.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import pytest  # noqa: E402

SIZED_FOR_THE_DEVICE = [
    "tests/test_gpu_batch.py::test_more_frames_than_one_grid_dimension_holds",
    "tests/test_gpu_hamming_knn.py::test_full_size_database_shards_find_planted_neighbours",
    "tests/test_undistort.py::test_gpu_full_size_batch_properties",
    "tests/test_gpu_projection.py::test_searched_frame_beyond_the_lds_plan",
    "tests/test_zz_no_alloc.py::test_noalloc_search_equals_the_allocating_search",
    # torch.distributed on the device (dist.py keeps its tensors on the GPU and all-gathers over RCCL)
    "tests/test_gpu_hamming_knn.py::test_sharded_search_equals_whole_database",
    "tests/test_gpu_hamming_knn.py::test_sharded_search_device_path_over_rccl_world_of_one",
    "tests/test_cpp_shim.py::test_shim_extract_and_match_on_gpu",
    "tests/test_cpp_shim.py::test_shim_from_four_threads_on_gpu",
]

if __name__ == "__main__":
    os.environ.setdefault("OVS_TEST_THREAD_ROUNDS", "2")
    args = sys.argv[1:]
    extra = []
    if "--full-size" in args:
        args.remove("--full-size")
    else:
        for t in SIZED_FOR_THE_DEVICE:
            extra += ["--deselect", t]
    sys.exit(pytest.main(["-p", "host_lib", "-q"] + extra + args))

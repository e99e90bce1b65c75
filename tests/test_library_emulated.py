"""The translation units of the HIP library -- host code and kernels, unchanged -- compiled for the
host against a stand-in for the HIP runtime (tools/emu/hip_host: kernels run as fibers, wave operations as
exchanges between the lanes that reach them, device memory and dynamic LDS surrounded by inaccessible address
space) and called through the same C ABI with the inputs of the GPU parity tests, in smaller sizes: results
must be the oracle's, and an access through a wild index ends the run with a report.  No GPU is involved:
this checks the index arithmetic and the data flow of the source that is also compiled for the device --
which kernel a call selects, what it reads and writes, that every workgroup barrier is reached, that the result
does not depend on the order in which threads run between synchronisation points -- not the timing and not the
hardware's memory model; the GPU parity tests stay the gate for the product.
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "emulated_cases.py")


def run(group, env=None):
    e = dict(os.environ)
    for k in ("OVS_RESOLVE", "OVS_GROUP_LIVE", "EMU_SHUFFLE", "EMU_REVERSE", "EMU_STRICT_LDS", "EMU_PREEMPT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, DRIVER, group], capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-5000:]
    m = re.search(r"^kernels: (.*)$", r.stdout, re.M)
    assert m and re.search(r"^OK \d+$", r.stdout, re.M), r.stdout[-2000:]
    return {kv.split("=")[0]: int(kv.split("=")[1]) for kv in m.group(1).split()}


def test_matchers_with_the_default_kernel_choice():
    # a call with one problem: grid build, candidates with 16 lanes per query, the two workgroup-wide resolve kernels
    # (without / with a ratio test); the single-wave kernel for the fuse and one-way kinds and when a query needs
    # more rows of candidates than the workgroup-wide kernel keeps
    k = run("matchers", {"EMU_STRICT_LDS": "1"})
    assert k["k_proj_resolve_wide"] > 0 and k["k_proj_resolve_jacobi"] > 0 and k["k_proj_resolve"] > 0
    assert k["k_proj_candidates"] > 0 and k["k_grid_build"] > 0


def test_matchers_with_the_workgroup_wide_kernels_forced_and_threads_shuffled():
    # (EMU_PREEMPT: every atomic operation is a possible switch to another thread)
    k = run("matchers", {"OVS_RESOLVE": "wide", "EMU_SHUFFLE": "11", "EMU_PREEMPT": "11"})
    assert k["k_proj_resolve_jacobi"] > 0 and k["k_proj_resolve_wide"] > 0


def test_matchers_on_the_single_wave_kernel_with_threads_in_reverse_order():
    # the batch configuration: eight lanes per query, one wave per problem
    k = run("matchers", {"OVS_RESOLVE": "wave", "OVS_GROUP_LIVE": "8", "EMU_REVERSE": "1"})
    assert "k_proj_resolve_wide" not in k and "k_proj_resolve_jacobi" not in k and k["k_proj_resolve"] > 0


def test_bow_transform_kernels():
    k = run("bow", {"EMU_SHUFFLE": "3"})
    assert k["k_bow_descend"] > 0 and k["k_bow_rank"] > 0 and k["k_bow_compact"] > 0


def test_extractor_and_stereo_kernels():
    # resize, FAST cells, quadtree (LDS and workspace instances), blur (both coefficient sets), orientation +
    # descriptors, the stereo matcher on the extractor's pyramids
    k = run("extract")
    for name in ("k_resize_rows", "k_fast_cells", "k_quadtree", "k_gauss7", "k_orient_desc", "k_stereo_match"):
        assert k[name] > 0, name


def test_hamming_database_triangulation_area_and_landmark_kernels():
    k = run("others", {"EMU_REVERSE": "1"})
    for name in ("k_hamming_pairs", "k_hamming_matrix", "k_knn2_partial", "k_window_list", "k_landmark_median", "k_proj_resolve"):
        assert k[name] > 0, name


def test_hostile_inputs_touch_no_foreign_memory():
    """tools/emu/fuzz_entry_points.py: keypoints and reprojections outside the image, NaN / infinite coordinates,
    octaves outside the pyramid, index lists past their arrays, damaged vocabularies, counts outside [0, capacity] --
    a dozen seeds per entry-point family on the host build, where a touch of memory that is not the library's own
    ends the process (on the device: a GPU memory fault).  Return codes are not judged."""
    fuzz = os.path.join(ROOT, "tools", "emu", "fuzz_entry_points.py")
    e = dict(os.environ)
    for k in ("OVS_RESOLVE", "OVS_GROUP_LIVE", "EMU_SHUFFLE", "EMU_REVERSE", "EMU_STRICT_LDS", "EMU_PREEMPT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, fuzz, "--seeds", "0:12"], capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-1500:] + r.stderr[-4000:]

"""The C-ABI library loads on a machine without a GPU and exports every function that
include/ovs.h declares; the CPU oracle exports the mirrored set.  No compute calls here."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = ROOT / "include" / "ovs.h"
DECL = re.compile(r"^(?:int|void|const char\*|uint32_t|int64_t|float)\s+(ovs_[a-z0-9_]+)\(", re.M)


def declared():
    names = DECL.findall(HEADER.read_text())
    assert len(names) >= 55, names
    return names


def test_header_is_plain_c(tmp_path):
    import subprocess
    src = tmp_path / "t.c"
    src.write_text('#include "ovs.h"\nint main(void) { return sizeof(ovs_keypoint) == 28 && sizeof(ovs_knn2) == 24 ? 0 : 1; }\n')
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    subprocess.run([str(exe)], check=True)


def test_library_exports_every_declared_symbol():
    from openvslam_amd import _lib
    L = _lib.lib()
    missing = [n for n in declared() if not hasattr(L, n)]
    assert not missing, missing


def test_python_binding_declares_argtypes_for_every_entry_point():
    from openvslam_amd import _lib
    L = _lib.lib()
    no_args = {"ovs_last_error", "ovs_device_count", "ovs_timing_reset"}
    undeclared = [n for n in declared() if n not in no_args and getattr(L, n).argtypes is None]
    assert not undeclared, undeclared


def test_oracle_mirrors_the_compute_entry_points():
    """Every function of the header that computes something on the device has a CPU restatement."""
    import oracle
    L = oracle.lib()
    not_compute = re.compile(r"ovs_(last_error|device_count|timing_|orb_params_default|orb_get_|orb_keypoint_capacity|orb_level_size"
                             r"|desc_db_(create|destroy|upload|attach_dev|device_bytes)|bow_vocab_|pack_batch_results|remap_(create|destroy)|map_)")
    renamed = {"ovs_orb_extract_batch_dev": "ovo_orb_extract", "ovs_orb_pyramid_level_dev": "ovo_orb_pyramid_level",
               "ovs_orb_download_pyramid_level": "ovo_orb_pyramid_level", "ovs_orb_download_blurred_level": "ovo_orb_blurred_level",
               "ovs_stereo_compute_dev": "ovo_stereo_compute", "ovs_stereo_compute_batch_dev": "ovo_stereo_compute", "ovs_stereo_compute_resident": "ovo_stereo_compute",
               "ovs_convert_to_grayscale_dev": "ovo_convert_to_grayscale", "ovs_equalize_histogram_dev": "ovo_equalize_histogram",
               "ovs_equalize_histogram_u16_dev": "ovo_equalize_histogram", "ovs_remap_apply_dev": "ovo_remap_linear",
               "ovs_remap_apply": "ovo_remap_linear",
               "ovs_desc_db_search": "ovo_knn2_search",
               "ovs_desc_db_search_dev": "ovo_knn2_search", "ovs_desc_db_search_noalloc_dev": "ovo_knn2_search", "ovs_knn2_merge_dev": "ovo_knn2_merge",
               "ovs_match_batch_current_and_last_dev": "ovo_match_batch_current_and_last",
               "ovs_convert_keypoints_to_bearings_dev": "ovo_convert_keypoints_to_bearings",
               "ovs_undistort_keypoints_dev": "ovo_undistort_keypoints",
               "ovs_bow_transform_batch_dev": "ovo_bow_transform"}
    missing = []
    for n in declared():
        if n == "ovs_orb_get_num_keypts_per_level" or not not_compute.match(n):
            if not hasattr(L, renamed.get(n, "ovo_" + n[4:])):
                missing.append(n)
    assert not missing, missing


def test_pod_layouts_match_the_header():
    from openvslam_amd import _abi as A
    assert ctypes.sizeof(A.Keypoint) == 28 == A.KEYPOINT_DTYPE.itemsize
    assert ctypes.sizeof(A.Knn2) == 24 == A.KNN2_DTYPE.itemsize
    assert ctypes.sizeof(A.Grid) == 40
    assert ctypes.sizeof(A.ImageView) == 24
    assert A.Grid.inv_cell_width.offset == 16 and A.Grid.num_cols.offset == 32


def test_host_only_entry_points_work_without_a_gpu():
    """Scale tables, cell indices, the cross-check and the top-2 merge are host code."""
    from openvslam_amd import _abi as A
    from openvslam_amd._lib import check, lib
    L = lib()
    out = np.zeros(8, np.float32)
    L.ovs_calc_scale_factors(8, 1.2, A.ptr(out, A.f32p))
    assert out[0] == 1.0 and abs(out[7] - 1.2 ** 7) < 1e-5
    b12 = np.array([2, -1, 0, 1], np.int32)
    b21 = np.array([2, 0, 0], np.int32)
    res = np.zeros(4, np.int32)
    num = ctypes.c_uint32()
    check(L.ovs_cross_check(A.ptr(b12, A.i32p), 4, A.ptr(b21, A.i32p), 3, A.ptr(res, A.i32p), ctypes.byref(num)))
    assert res.tolist() == [2, -1, 0, -1] and num.value == 2


def test_compute_entry_points_fail_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from openvslam_amd import _abi as A
    from openvslam_amd._lib import OvsError, check, lib
    a = np.zeros((4, 32), np.uint8)
    out = np.zeros(4, np.uint32)
    with pytest.raises(OvsError) as e:
        check(lib().ovs_hamming_pairs(A.ptr(a, A.u8p), A.ptr(a, A.u8p), 4, A.ptr(out, A.u32p), 0))
    assert e.value.code == A.OVS_ERR_HIP
    from openvslam_amd.feature import orb_extractor, orb_params
    with pytest.raises(OvsError):
        orb_extractor(orb_params())

"""ctypes loader for the CPU oracle (oracle/libovs_oracle.so) -- test infrastructure only.

Nothing under openvslam_amd/ imports this module; it is used by tests/, by
__graft_entry__.smoke() and by the cpu_baseline leg of bench.py as the checker / baseline.
"""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

from openvslam_amd import _abi as A

ROOT = Path(__file__).resolve().parent.parent
LIB_PATH = ROOT / "oracle" / "libovs_oracle.so"


def build(force=False):
    if force or not LIB_PATH.exists():
        subprocess.run(["make", "-C", str(ROOT / "oracle")], check=True, capture_output=True)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(LIB_PATH))
        _declare(_lib)
    return _lib


def _declare(L):
    sz = C.c_size_t
    i = C.c_int
    L.ovo_last_error.restype = C.c_char_p
    L.ovo_cv_round_f.argtypes = [C.c_float]
    L.ovo_cv_round_d.argtypes = [C.c_double]
    L.ovo_cv_floor_f.argtypes = [C.c_float]
    L.ovo_cv_ceil_f.argtypes = [C.c_float]
    L.ovo_cv_ceil_d.argtypes = [C.c_double]
    L.ovo_fast_atan2.argtypes = [C.c_float, C.c_float]
    L.ovo_fast_atan2.restype = C.c_float
    L.ovo_cos.argtypes = [C.c_float]
    L.ovo_cos.restype = C.c_float
    L.ovo_sin.argtypes = [C.c_float]
    L.ovo_sin.restype = C.c_float
    L.ovo_resize_linear_u8.argtypes = [A.u8p, i, i, sz, A.u8p, i, i, sz]
    L.ovo_fast_9_16.argtypes = [A.u8p, i, i, sz, i, A.kpp, i]
    L.ovo_gaussian_blur_7x7.argtypes = [A.u8p, i, i, sz, A.u8p, sz]
    L.ovo_gaussian_kernel_7_q8.argtypes = [A.i32p]
    L.ovo_gaussian_kernel_7_q8_normalised.argtypes = [A.i32p]
    L.ovo_gaussian_blur_7x7_variant.argtypes = [A.u8p, i, i, sz, A.u8p, sz, i]
    L.ovo_orb_set_blur_kernel.argtypes = [C.c_void_p, i]
    for name in ("scale_factors", "inv_scale_factors", "level_sigma_sq", "inv_level_sigma_sq"):
        getattr(L, "ovo_calc_" + name).argtypes = [C.c_uint32, C.c_float, A.f32p]
        getattr(L, "ovo_calc_" + name).restype = None
    L.ovo_orb_create.argtypes = [C.POINTER(A.OrbParams), C.POINTER(C.c_void_p)]
    L.ovo_orb_destroy.argtypes = [C.c_void_p]
    L.ovo_orb_destroy.restype = None
    L.ovo_orb_set_max_num_keypoints.argtypes = [C.c_void_p, C.c_uint32]
    L.ovo_orb_set_scale_factor.argtypes = [C.c_void_p, C.c_float]
    L.ovo_orb_set_num_scale_levels.argtypes = [C.c_void_p, C.c_uint32]
    L.ovo_orb_set_initial_fast_threshold.argtypes = [C.c_void_p, C.c_uint32]
    L.ovo_orb_set_minimum_fast_threshold.argtypes = [C.c_void_p, C.c_uint32]
    L.ovo_orb_get_num_keypts_per_level.argtypes = [C.c_void_p, A.u32p]
    L.ovo_orb_get_u_max.argtypes = [C.c_void_p, A.i32p]
    L.ovo_orb_extract.argtypes = [C.c_void_p, A.u8p, i, i, sz, A.u8p, sz, A.kpp, A.u8p, i, C.POINTER(i)]
    view = [C.c_void_p, i, C.POINTER(A.u8p), C.POINTER(i), C.POINTER(i), C.POINTER(sz)]
    L.ovo_orb_pyramid_level.argtypes = view
    L.ovo_orb_blurred_level.argtypes = view
    L.ovo_orb_candidates.argtypes = [C.c_void_p, i, A.kpp, i]
    L.ovo_orb_stage_seconds.argtypes = [C.c_void_p, A.f64p]
    L.ovo_create_rectangle_mask.argtypes = [C.POINTER(A.OrbParams), i, i, A.u8p]
    L.ovo_hamming_32.argtypes = [A.u8p, A.u8p]
    L.ovo_hamming_32.restype = C.c_uint32
    L.ovo_hamming_64.argtypes = [A.u8p, A.u8p]
    L.ovo_hamming_64.restype = C.c_uint32
    L.ovo_hamming_pairs.argtypes = [A.u8p, A.u8p, i, A.u32p]
    L.ovo_hamming_matrix.argtypes = [A.u8p, i, A.u8p, i, A.u16p]
    L.ovo_angle_checker_create.argtypes = [C.c_uint32, C.c_uint32]
    L.ovo_angle_checker_create.restype = C.c_void_p
    L.ovo_angle_checker_destroy.argtypes = [C.c_void_p]
    L.ovo_angle_checker_destroy.restype = None
    L.ovo_angle_checker_append.argtypes = [C.c_void_p, C.c_float, i]
    L.ovo_angle_checker_append.restype = None
    L.ovo_angle_checker_valid.argtypes = [C.c_void_p, A.i32p, i]
    L.ovo_angle_checker_invalid.argtypes = [C.c_void_p, A.i32p, i]
    L.ovo_angle_checker_partition.argtypes = [A.f32p, i, i, A.u8p]
    L.ovo_get_cell_indices.argtypes = [C.POINTER(A.Grid), C.c_float, C.c_float, C.POINTER(i), C.POINTER(i)]
    fv = C.POINTER(A.FrameView)
    pq = C.POINTER(A.ProjQueries)
    L.ovo_get_keypoints_in_cell.argtypes = [fv, i, A.f32p, A.f32p, A.i32p, A.i32p, A.u32p, i, A.u32p]
    L.ovo_match_current_and_last_frames.argtypes = [fv, pq, A.u8p, C.c_float, i, i, C.c_float, i, A.i32p, A.u32p]
    L.ovo_match_frame_and_landmarks.argtypes = [fv, pq, A.u8p, C.c_float, C.c_float, A.i32p, A.u32p]
    L.ovo_match_frame_and_keyframe.argtypes = [fv, pq, A.u8p, C.c_float, C.c_uint32, i, A.i32p, A.u32p]
    L.ovo_match_by_sim3_transform.argtypes = [fv, pq, A.u8p, C.c_float, A.i32p, A.u32p]
    L.ovo_match_keyframe_one_way.argtypes = [fv, pq, C.c_float, A.i32p]
    L.ovo_cross_check.argtypes = [A.i32p, i, A.i32p, i, A.i32p, A.u32p]
    iv = C.POINTER(A.ImageView)
    L.ovo_stereo_compute.argtypes = [iv, iv, i, A.kpp, i, A.kpp, i, A.u8p, A.u8p, A.f32p, A.f32p,
                                     C.c_float, C.c_float, A.f32p, A.f32p]
    L.ovo_brute_force_match.argtypes = [A.u8p, A.f32p, i, A.u8p, A.f32p, A.u8p, i, C.c_float, i, A.i32p, A.u32p]
    bv = C.POINTER(A.BowFeatVec)
    L.ovo_bow_match_frame_and_keyframe.argtypes = [bv, A.u8p, A.f32p, A.u8p, i, bv, A.u8p, A.f32p, i,
                                                   C.c_float, i, A.i32p, A.u32p]
    L.ovo_bow_transform.argtypes = [C.POINTER(A.BowVocabView), A.u8p, i, i, A.u32p, A.u32p, A.u32p, A.f64p, C.POINTER(C.c_int32),
                                    A.u32p, A.u32p, A.u32p, C.POINTER(C.c_int32)]
    L.ovo_bow_match_keyframes.argtypes = [bv, A.u8p, A.f32p, A.u8p, i, bv, A.u8p, A.f32p, A.u8p, i,
                                          C.c_float, i, A.i32p, A.u32p]
    L.ovo_match_for_triangulation.argtypes = [bv, A.u8p, A.kpp, A.f64p, A.f32p, A.u8p, i,
                                              bv, A.u8p, A.kpp, A.f64p, A.f32p, A.u8p, i,
                                              A.f64p, A.f64p, A.f32p, i, i, A.u32p, A.u32p]
    L.ovo_match_in_consistent_area.argtypes = [A.kpp, A.u8p, i, fv, A.f32p, i, C.c_float, i, A.i32p, A.u32p]
    L.ovo_landmark_compute_descriptors.argtypes = [A.u8p, A.u32p, i, A.i32p]
    L.ovo_fuse_best_candidates.argtypes = [fv, pq, A.f64p, A.f32p, C.c_float, i, A.i32p]
    L.ovo_convert_to_grayscale.argtypes = [A.u8p, i, i, i, sz, i, A.u8p, sz]
    L.ovo_equalize_histogram.argtypes = [C.c_void_p, i, i, sz, i, A.u8p, sz]
    L.ovo_remap_linear.argtypes = [A.u8p, i, i, sz, A.f32p, A.f32p, i, i, A.u8p, sz]
    L.ovo_remap_linear_table.argtypes = [C.POINTER(C.c_int16)]
    L.ovo_remap_linear_table.restype = None
    L.ovo_knn2_search.argtypes = [A.u8p, C.c_int64, C.c_int64, A.u8p, i, A.knn2p]
    L.ovo_knn2_merge.argtypes = [A.knn2p, i, i, A.knn2p]
    L.ovo_match_batch_current_and_last.argtypes = [A.kpp, A.u8p, A.i32p, i, i, A.f32p, C.POINTER(A.Grid),
                                                   A.f32p, i, C.c_float, i, A.i32p, A.u32p]
    L.ovo_convert_keypoints_to_bearings.argtypes = [C.POINTER(A.Camera), A.kpp, i, A.f64p]
    L.ovo_undistort_keypoints.argtypes = [C.POINTER(A.Camera), A.kpp, i, A.kpp]


def check(rc):
    if rc != 0:
        raise RuntimeError("oracle error %d: %s" % (rc, lib().ovo_last_error().decode()))


class OrbExtractor:
    """Oracle counterpart of feature::orb_extractor."""

    def __init__(self, **params):
        self._params = A.make_orb_params(**params)
        h = C.c_void_p()
        check(lib().ovo_orb_create(self._params.ref(), C.byref(h)))
        self._h = h
        self.num_levels = self._params.struct.num_levels
        self.max_num_keypts = self._params.struct.max_num_keypts

    def __del__(self):
        if getattr(self, "_h", None):
            lib().ovo_orb_destroy(self._h)
            self._h = None

    def set_blur_kernel(self, variant):
        check(lib().ovo_orb_set_blur_kernel(self._h, variant))

    def extract(self, img, mask=None, cap=None):
        img = np.ascontiguousarray(img, dtype=np.uint8)
        h, w = img.shape
        cap = cap or 4 * self.max_num_keypts + 1024
        kps = np.zeros(cap, dtype=A.KEYPOINT_DTYPE)
        desc = np.zeros((cap, 32), dtype=np.uint8)
        n = C.c_int(0)
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        check(lib().ovo_orb_extract(self._h, A.ptr(img, A.u8p), w, h, w, A.ptr(m, A.u8p), w if m is not None else 0,
                                    A.ptr(kps, A.kpp), A.ptr(desc, A.u8p), cap, C.byref(n)))
        return kps[:n.value].copy(), desc[:n.value].copy()

    def _level(self, fn, level):
        p = A.u8p()
        w, h, s = C.c_int(), C.c_int(), C.c_size_t()
        check(fn(self._h, level, C.byref(p), C.byref(w), C.byref(h), C.byref(s)))
        if not p:
            return None
        buf = np.ctypeslib.as_array(p, shape=(h.value, s.value))
        return buf[:, :w.value].copy()

    def pyramid_level(self, level):
        return self._level(lib().ovo_orb_pyramid_level, level)

    def blurred_level(self, level):
        return self._level(lib().ovo_orb_blurred_level, level)

    def candidates(self, level, cap=400000):
        out = np.zeros(cap, dtype=A.KEYPOINT_DTYPE)
        n = lib().ovo_orb_candidates(self._h, level, A.ptr(out, A.kpp), cap)
        assert 0 <= n <= cap
        return out[:n].copy()

    def num_keypts_per_level(self):
        out = np.zeros(self.num_levels, dtype=np.uint32)
        check(lib().ovo_orb_get_num_keypts_per_level(self._h, A.ptr(out, A.u32p)))
        return out

    def stage_seconds(self):
        out = np.zeros(6, dtype=np.float64)
        check(lib().ovo_orb_stage_seconds(self._h, A.ptr(out, A.f64p)))
        return dict(zip(("pyramid", "fast", "quadtree", "orientation", "blur", "descriptors"), out.tolist()))


def scale_table(name, num_levels, scale_factor):
    out = np.zeros(num_levels, dtype=np.float32)
    getattr(lib(), "ovo_calc_" + name)(num_levels, scale_factor, A.ptr(out, A.f32p))
    return out

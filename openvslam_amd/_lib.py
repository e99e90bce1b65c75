"""ctypes loader of the HIP library (openvslam_amd/lib/libovs_hip.so).

There is no CPU fallback: if the shared library has not been built (openvslam_amd/csrc/Makefile, or
__graft_entry__.build()) or cannot be loaded, importing symbols from it raises.
"""
import ctypes as C
from pathlib import Path

from . import _abi as A

LIB_PATH = Path(__file__).resolve().parent / "lib" / "libovs_hip.so"

# status codes of include/ovs.h
OVS_OK, OVS_ERR_INVALID_ARGUMENT, OVS_ERR_CAPACITY, OVS_ERR_HIP, OVS_ERR_UNSUPPORTED, OVS_ERR_OUT_OF_RANGE = 0, -1, -2, -3, -4, -5

_lib = None


class OvsError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("ovs error %d: %s" % (code, message))
        self.code = code
        self.message = message


def source_fingerprint():
    """sha256 over the sources the library is built from (csrc/*.hip, csrc/*.h, include/ovs.h, in name
    order): stored with counter profiles so that a profile taken from other kernels can be told apart."""
    import hashlib
    root = Path(__file__).resolve().parent
    files = sorted((root / "csrc").glob("*.hip")) + sorted((root / "csrc").glob("*.h")) + [root.parent / "include" / "ovs.h"]
    h = hashlib.sha256()
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()


def lib():
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError("%s is missing: build it with `make -C openvslam_amd/csrc` "
                              "(the HIP extension is mandatory, there is no CPU path)" % LIB_PATH)
        try:
            # PyTorch-ROCm ships its own copy of the HIP runtime; when both live in one process the
            # copy torch loads must come first or torch later sees no devices
            import torch  # noqa: F401
        except ImportError:
            pass
        _lib = C.CDLL(str(LIB_PATH))
        _declare(_lib)
    return _lib


def check(rc):
    if rc != 0:
        raise OvsError(rc, lib().ovs_last_error().decode())


def _declare(L):
    sz, i, vp = C.c_size_t, C.c_int, C.c_void_p
    L.ovs_last_error.restype = C.c_char_p
    L.ovs_device_count.restype = i
    L.ovs_orb_params_default.argtypes = [C.POINTER(A.OrbParams)]
    L.ovs_orb_params_default.restype = None
    for name in ("scale_factors", "inv_scale_factors", "level_sigma_sq", "inv_level_sigma_sq"):
        getattr(L, "ovs_calc_" + name).argtypes = [C.c_uint32, C.c_float, A.f32p]
        getattr(L, "ovs_calc_" + name).restype = None
        getattr(L, "ovs_orb_get_" + name).argtypes = [vp, A.f32p]
    L.ovs_orb_set_blur_kernel.argtypes = [vp, i]
    L.ovs_orb_get_blur_kernel.argtypes = [vp]
    L.ovs_orb_create.argtypes = [C.POINTER(A.OrbParams), i, C.POINTER(vp)]
    L.ovs_orb_destroy.argtypes = [vp]
    L.ovs_orb_destroy.restype = None
    for name, typ in (("max_num_keypoints", C.c_uint32), ("scale_factor", C.c_float), ("num_scale_levels", C.c_uint32),
                      ("initial_fast_threshold", C.c_uint32), ("minimum_fast_threshold", C.c_uint32)):
        getattr(L, "ovs_orb_get_" + name).argtypes = [vp]
        getattr(L, "ovs_orb_get_" + name).restype = typ
        getattr(L, "ovs_orb_set_" + name).argtypes = [vp, typ]
    L.ovs_orb_get_num_keypts_per_level.argtypes = [vp, A.u32p]
    L.ovs_orb_keypoint_capacity.argtypes = [vp, i, i]
    L.ovs_orb_extract.argtypes = [vp, A.u8p, i, i, sz, A.u8p, sz, A.kpp, A.u8p, i, C.POINTER(i)]
    L.ovs_orb_extract_batch_dev.argtypes = [vp, vp, i, i, i, sz, sz, vp, sz, vp, vp, vp, i, vp]
    L.ovs_orb_pyramid_level_dev.argtypes = [vp, i, i, C.POINTER(vp), C.POINTER(i), C.POINTER(i), C.POINTER(sz)]
    L.ovs_orb_download_pyramid_level.argtypes = [vp, i, i, A.u8p, sz]
    L.ovs_orb_download_blurred_level.argtypes = [vp, i, i, A.u8p, sz]
    L.ovs_orb_level_size.argtypes = [vp, i, i, i, C.POINTER(i), C.POINTER(i)]
    L.ovs_timing_enable.argtypes = [i]
    L.ovs_timing_enable.restype = None
    L.ovs_timing_reset.restype = None
    L.ovs_timing_query.argtypes = [C.c_char_p, C.POINTER(C.c_double), C.POINTER(i)]
    L.ovs_timing_names.argtypes = [C.c_char_p, i]
    L.ovs_hamming_pairs.argtypes = [A.u8p, A.u8p, i, A.u32p, i]
    L.ovs_hamming_matrix.argtypes = [A.u8p, i, A.u8p, i, A.u16p, i]
    L.ovs_get_cell_indices.argtypes = [C.POINTER(A.Grid), C.c_float, C.c_float, C.POINTER(i), C.POINTER(i)]
    fv, pq = C.POINTER(A.FrameView), C.POINTER(A.ProjQueries)
    L.ovs_get_keypoints_in_cell.argtypes = [fv, i, A.f32p, A.f32p, A.i32p, A.i32p, A.u32p, i, A.u32p, i]
    L.ovs_match_current_and_last_frames.argtypes = [fv, pq, A.u8p, C.c_float, i, i, C.c_float, i, A.i32p, A.u32p, i]
    L.ovs_match_frame_and_landmarks.argtypes = [fv, pq, A.u8p, C.c_float, C.c_float, A.i32p, A.u32p, i]
    L.ovs_match_frame_and_keyframe.argtypes = [fv, pq, A.u8p, C.c_float, C.c_uint32, i, A.i32p, A.u32p, i]
    L.ovs_match_by_sim3_transform.argtypes = [fv, pq, A.u8p, C.c_float, A.i32p, A.u32p, i]
    L.ovs_match_keyframe_one_way.argtypes = [fv, pq, C.c_float, A.i32p, i]
    L.ovs_cross_check.argtypes = [A.i32p, i, A.i32p, i, A.i32p, A.u32p]
    L.ovs_angle_checker_partition.argtypes = [A.f32p, i, i, A.u8p, i]
    iv = C.POINTER(A.ImageView)
    L.ovs_stereo_compute.argtypes = [iv, iv, i, A.kpp, i, A.kpp, i, A.u8p, A.u8p, A.f32p, A.f32p,
                                     C.c_float, C.c_float, A.f32p, A.f32p, i]
    L.ovs_stereo_compute_dev.argtypes = [vp, i, vp, i, vp, vp, vp, vp, vp, vp, i, C.c_float, C.c_float, vp, vp, vp, i, vp]
    sz = C.c_size_t
    L.ovs_convert_to_grayscale_dev.argtypes = [vp, i, i, i, i, sz, sz, i, vp, sz, sz, i, vp]
    L.ovs_convert_to_grayscale.argtypes = [A.u8p, i, i, i, sz, i, A.u8p, sz, i]
    L.ovs_equalize_histogram_dev.argtypes = [vp, i, i, i, sz, sz, vp, i, vp]
    L.ovs_equalize_histogram_u16_dev.argtypes = [vp, i, i, i, sz, sz, vp, sz, sz, vp, i, vp]
    L.ovs_equalize_histogram.argtypes = [vp, i, i, sz, i, A.u8p, sz, i]
    L.ovs_remap_create.argtypes = [A.f32p, A.f32p, i, i, i, C.POINTER(vp)]
    L.ovs_remap_destroy.argtypes = [vp]
    L.ovs_remap_destroy.restype = None
    L.ovs_remap_apply_dev.argtypes = [vp, vp, i, i, i, sz, sz, vp, sz, sz, vp]
    L.ovs_remap_apply.argtypes = [vp, A.u8p, i, i, sz, A.u8p, sz]
    L.ovs_map_open.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.ovs_map_close.argtypes = [vp]
    L.ovs_map_close.restype = None
    L.ovs_map_num_keyframes.argtypes = [vp]
    L.ovs_map_num_descriptors.argtypes = [vp]
    L.ovs_map_num_descriptors.restype = C.c_int64
    L.ovs_map_keyframe_rows.argtypes = [vp, A.i32p, C.POINTER(C.c_int64)]
    L.ovs_map_copy_descriptors.argtypes = [vp, A.u8p]
    L.ovs_stereo_compute_resident.argtypes = [vp, vp, A.kpp, i, A.kpp, i, A.u8p, A.u8p, C.c_float, C.c_float, A.f32p, A.f32p, i]
    L.ovs_stereo_compute_batch_dev.argtypes = [vp, vp, i, vp, vp, vp, vp, vp, vp, i, C.c_float, C.c_float, vp, vp, vp, i, vp]
    L.ovs_brute_force_match.argtypes = [A.u8p, A.f32p, i, A.u8p, A.f32p, A.u8p, i, C.c_float, i, A.i32p, A.u32p, i]
    bv = C.POINTER(A.BowFeatVec)
    L.ovs_bow_match_frame_and_keyframe.argtypes = [bv, A.u8p, A.f32p, A.u8p, i, bv, A.u8p, A.f32p, i,
                                                   C.c_float, i, A.i32p, A.u32p, i]
    L.ovs_bow_match_keyframes.argtypes = [bv, A.u8p, A.f32p, A.u8p, i, bv, A.u8p, A.f32p, A.u8p, i,
                                          C.c_float, i, A.i32p, A.u32p, i]
    L.ovs_match_for_triangulation.argtypes = [bv, A.u8p, A.kpp, A.f64p, A.f32p, A.u8p, i,
                                              bv, A.u8p, A.kpp, A.f64p, A.f32p, A.u8p, i,
                                              A.f64p, A.f64p, A.f32p, i, i, A.u32p, A.u32p, i]
    L.ovs_pack_batch_results_dev.argtypes = [vp, i, i, vp, vp, vp, vp, vp, vp, vp, i, vp]
    vv = C.POINTER(A.BowVocabView)
    i32 = C.POINTER(C.c_int32)
    L.ovs_bow_vocab_create.argtypes = [vv, i, C.POINTER(vp)]
    L.ovs_bow_vocab_destroy.argtypes = [vp]
    L.ovs_bow_vocab_destroy.restype = None
    L.ovs_bow_transform.argtypes = [vp, A.u8p, i, i, A.u32p, A.u32p, A.u32p, A.f64p, i32, A.u32p, A.u32p, A.u32p, i32]
    L.ovs_bow_transform_batch_dev.argtypes = [vp, vp, vp, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.ovs_match_in_consistent_area.argtypes = [A.kpp, A.u8p, i, fv, A.f32p, i, C.c_float, i, A.i32p, A.u32p, i]
    L.ovs_landmark_compute_descriptors.argtypes = [A.u8p, A.u32p, i, A.i32p, i]
    L.ovs_fuse_best_candidates.argtypes = [fv, pq, A.f64p, A.f32p, C.c_float, i, A.i32p, i]
    L.ovs_desc_db_create.argtypes = [i, C.POINTER(vp)]
    L.ovs_desc_db_destroy.argtypes = [vp]
    L.ovs_desc_db_destroy.restype = None
    L.ovs_desc_db_upload.argtypes = [vp, A.u8p, C.c_int64, C.c_int64]
    L.ovs_desc_db_attach_dev.argtypes = [vp, vp, C.c_int64, C.c_int64]
    L.ovs_desc_db_device_bytes.argtypes = [vp, C.POINTER(C.c_int64)]
    L.ovs_desc_db_search.argtypes = [vp, A.u8p, i, A.knn2p]
    L.ovs_desc_db_search_dev.argtypes = [vp, vp, i, vp, vp]
    L.ovs_desc_db_search_noalloc_dev.argtypes = [vp, vp, i, vp, vp]
    L.ovs_knn2_merge.argtypes = [A.knn2p, i, i, A.knn2p]
    L.ovs_knn2_merge_dev.argtypes = [vp, i, i, vp, vp]
    L.ovs_match_batch_current_and_last_dev.argtypes = [vp, vp, vp, i, i, vp, C.POINTER(A.Grid), A.f32p, i,
                                                       C.c_float, i, vp, vp, i, vp]
    L.ovs_convert_keypoints_to_bearings_dev.argtypes = [C.POINTER(A.Camera), vp, vp, i, i, vp, i, vp]
    L.ovs_convert_keypoints_to_bearings.argtypes = [C.POINTER(A.Camera), A.kpp, i, A.f64p, i]
    L.ovs_undistort_keypoints_dev.argtypes = [C.POINTER(A.Camera), vp, vp, i, i, vp, vp, i, vp]
    L.ovs_undistort_keypoints.argtypes = [C.POINTER(A.Camera), A.kpp, i, A.kpp, i]


def timing_report():
    """{kernel name: (total ms, launches)} accumulated since the last reset on this thread."""
    L = lib()
    buf = C.create_string_buffer(64 * 64)
    n = L.ovs_timing_names(buf, 64)
    out = {}
    for k in range(min(n, 64)):
        name = buf.raw[64 * k:64 * (k + 1)].split(b"\0", 1)[0]
        ms, cnt = C.c_double(), C.c_int()
        check(L.ovs_timing_query(name, C.byref(ms), C.byref(cnt)))
        out[name.decode()] = (ms.value, cnt.value)
    return out

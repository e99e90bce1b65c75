"""ctypes mirror of the POD types of include/ovs.h (shared by the product binding and the tests)."""
import ctypes as C

import numpy as np

OVS_OK = 0
OVS_ERR_INVALID_ARGUMENT = -1
OVS_ERR_CAPACITY = -2
OVS_ERR_HIP = -3
OVS_ERR_UNSUPPORTED = -4
OVS_ERR_OUT_OF_RANGE = -5

DIR_NONE, DIR_FORWARD, DIR_BACKWARD = 0, 1, 2
BLUR_ROUNDED, BLUR_NORMALISED = 0, 1

# binary layout of cv::KeyPoint (28 bytes)
KEYPOINT_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"),
                           ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
assert KEYPOINT_DTYPE.itemsize == 28

KNN2_DTYPE = np.dtype([("dist", "<u4", (2,)), ("row", "<i8", (2,))])
assert KNN2_DTYPE.itemsize == 24

u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)
u32p = C.POINTER(C.c_uint32)
u16p = C.POINTER(C.c_uint16)


class Keypoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32), ("class_id", C.c_int32)]


kpp = C.POINTER(Keypoint)


class OrbParams(C.Structure):
    _fields_ = [("max_num_keypts", C.c_uint32), ("scale_factor", C.c_float), ("num_levels", C.c_uint32),
                ("ini_fast_thr", C.c_uint32), ("min_fast_thr", C.c_uint32),
                ("num_mask_rects", C.c_uint32), ("mask_rects", f32p)]


class Grid(C.Structure):
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float),
                ("inv_cell_width", C.c_double), ("inv_cell_height", C.c_double),
                ("num_cols", C.c_uint32), ("num_rows", C.c_uint32)]


class FrameView(C.Structure):
    _fields_ = [("num_keypts", C.c_int32), ("undist_keypts", kpp), ("descriptors", u8p),
                ("stereo_x_right", f32p), ("scale_factors", f32p), ("num_scale_levels", C.c_int32),
                ("grid", Grid)]


class ProjQueries(C.Structure):
    _fields_ = [("n", C.c_int32), ("reproj_xy", f32p), ("x_right", f32p), ("scale_level", i32p),
                ("angle", f32p), ("desc", u8p), ("valid", u8p), ("has_observation", u8p)]


class ImageView(C.Structure):
    _fields_ = [("data", u8p), ("width", C.c_int32), ("height", C.c_int32), ("stride", C.c_size_t)]


class BowFeatVec(C.Structure):
    _fields_ = [("num_nodes", C.c_int32), ("node_ids", u32p), ("offsets", u32p), ("indices", u32p)]


class BowVocabView(C.Structure):
    _fields_ = [("branching_factor", C.c_int32), ("depth_levels", C.c_int32), ("weighting", C.c_int32), ("scoring", C.c_int32),
                ("num_nodes", C.c_int32), ("node_descs", u8p), ("child_offsets", u32p), ("children", u32p), ("word_ids", u32p),
                ("weights", f64p)]


BOW_TF_IDF, BOW_TF, BOW_IDF, BOW_BINARY = 0, 1, 2, 3
BOW_L1_NORM, BOW_L2_NORM, BOW_CHI_SQUARE, BOW_KL, BOW_BHATTACHARYYA, BOW_DOT_PRODUCT = 0, 1, 2, 3, 4, 5


class Knn2(C.Structure):
    _fields_ = [("dist", C.c_uint32 * 2), ("row", C.c_int64 * 2)]


knn2p = C.POINTER(Knn2)

CAMERA_PERSPECTIVE, CAMERA_FISHEYE, CAMERA_EQUIRECTANGULAR = 0, 1, 2


class Camera(C.Structure):
    _fields_ = [("model", C.c_int32), ("cols", C.c_uint32), ("rows", C.c_uint32), ("reserved", C.c_int32),
                ("fx", C.c_double), ("fy", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("dist", C.c_double * 5)]


assert C.sizeof(Knn2) == 24 and C.sizeof(Keypoint) == 28


def ptr(arr, typ):
    """Pointer of ctypes pointer type `typ` to the data of a C-contiguous numpy array (or None)."""
    if arr is None:
        return C.cast(None, typ)
    assert arr.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return C.cast(arr.ctypes.data, typ)


def make_grid(min_x, min_y, max_x, max_y, num_cols=64, num_rows=48):
    """camera::base grid geometry (src/openvslam/camera/perspective.cc:27-30)."""
    g = Grid()
    g.min_x, g.min_y, g.max_x, g.max_y = np.float32(min_x), np.float32(min_y), np.float32(max_x), np.float32(max_y)
    # static_cast<double>(num_grid_cols_) / (img_bounds_.max_x_ - img_bounds_.min_x_): float difference
    g.inv_cell_width = float(num_cols) / float(np.float32(max_x) - np.float32(min_x))
    g.inv_cell_height = float(num_rows) / float(np.float32(max_y) - np.float32(min_y))
    g.num_cols, g.num_rows = num_cols, num_rows
    return g


class Holder:
    """Keeps numpy arrays alive next to the ctypes struct that points into them."""

    def __init__(self, struct, *keep):
        self.struct = struct
        self.keep = keep

    def ref(self):
        return C.byref(self.struct)


def make_frame_view(keypts, descs, scale_factors, grid, stereo_x_right=None):
    keypts = np.ascontiguousarray(keypts, dtype=KEYPOINT_DTYPE)
    descs = np.ascontiguousarray(descs, dtype=np.uint8)
    sf = np.ascontiguousarray(scale_factors, dtype=np.float32)
    sxr = None if stereo_x_right is None else np.ascontiguousarray(stereo_x_right, dtype=np.float32)
    v = FrameView()
    v.num_keypts = len(keypts)
    v.undist_keypts = ptr(keypts, kpp)
    v.descriptors = ptr(descs, u8p)
    v.stereo_x_right = ptr(sxr, f32p)
    v.scale_factors = ptr(sf, f32p)
    v.num_scale_levels = len(sf)
    v.grid = grid
    return Holder(v, keypts, descs, sf, sxr)


def make_proj_queries(reproj_xy, scale_level, desc, valid=None, x_right=None, angle=None, has_observation=None):
    n = len(scale_level)
    reproj_xy = np.ascontiguousarray(reproj_xy, dtype=np.float32).reshape(n, 2)
    scale_level = np.ascontiguousarray(scale_level, dtype=np.int32)
    desc = np.ascontiguousarray(desc, dtype=np.uint8).reshape(n, 32)
    valid = np.ones(n, np.uint8) if valid is None else np.ascontiguousarray(valid, dtype=np.uint8)
    x_right = None if x_right is None else np.ascontiguousarray(x_right, dtype=np.float32)
    angle = np.zeros(n, np.float32) if angle is None else np.ascontiguousarray(angle, dtype=np.float32)
    has_observation = None if has_observation is None else np.ascontiguousarray(has_observation, dtype=np.uint8)
    q = ProjQueries()
    q.n = n
    q.reproj_xy = ptr(reproj_xy, f32p)
    q.x_right = ptr(x_right, f32p)
    q.scale_level = ptr(scale_level, i32p)
    q.angle = ptr(angle, f32p)
    q.desc = ptr(desc, u8p)
    q.valid = ptr(valid, u8p)
    q.has_observation = ptr(has_observation, u8p)
    return Holder(q, reproj_xy, scale_level, desc, valid, x_right, angle, has_observation)


def make_bow_feat_vec(node_to_indices):
    """node_to_indices: dict {node_id: [keypoint indices]} (a DBoW2::FeatureVector)."""
    ids = sorted(node_to_indices)
    node_ids = np.array(ids, dtype=np.uint32)
    offsets = np.zeros(len(ids) + 1, dtype=np.uint32)
    flat = []
    for k, nid in enumerate(ids):
        flat.extend(node_to_indices[nid])
        offsets[k + 1] = len(flat)
    indices = np.array(flat, dtype=np.uint32)
    v = BowFeatVec()
    v.num_nodes = len(ids)
    v.node_ids = ptr(node_ids, u32p)
    v.offsets = ptr(offsets, u32p)
    v.indices = ptr(indices, u32p)
    return Holder(v, node_ids, offsets, indices)


def make_bow_vocab_view(k, L, node_descs, child_offsets, children, word_ids, weights, weighting=BOW_TF_IDF, scoring=BOW_L1_NORM):
    """A DBoW2 vocabulary tree as arrays (node 0 = root, CSR children lists)."""
    nd = np.ascontiguousarray(node_descs, dtype=np.uint8).reshape(-1, 32)
    co = np.ascontiguousarray(child_offsets, dtype=np.uint32)
    ch = np.ascontiguousarray(children, dtype=np.uint32)
    wi = np.ascontiguousarray(word_ids, dtype=np.uint32)
    we = np.ascontiguousarray(weights, dtype=np.float64)
    v = BowVocabView()
    v.branching_factor, v.depth_levels, v.weighting, v.scoring = k, L, weighting, scoring
    v.num_nodes = len(co) - 1
    v.node_descs = ptr(nd, u8p)
    v.child_offsets = ptr(co, u32p)
    v.children = ptr(ch, u32p)
    v.word_ids = ptr(wi, u32p)
    v.weights = ptr(we, f64p)
    return Holder(v, nd, co, ch, wi, we)


def make_orb_params(max_num_keypts=2000, scale_factor=1.2, num_levels=8, ini_fast_thr=20, min_fast_thr=7,
                    mask_rects=()):
    rects = np.ascontiguousarray(np.array(mask_rects, dtype=np.float32).reshape(-1, 4))
    p = OrbParams()
    p.max_num_keypts = max_num_keypts
    p.scale_factor = scale_factor
    p.num_levels = num_levels
    p.ini_fast_thr = ini_fast_thr
    p.min_fast_thr = min_fast_thr
    p.num_mask_rects = len(rects)
    p.mask_rects = ptr(rects, f32p) if len(rects) else C.cast(None, f32p)
    return Holder(p, rects)

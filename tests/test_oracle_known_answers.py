"""Pins the CPU oracle against every known-answer vector the reference's own unit tests hold for
the hot path (the reference's tests are gtest programs that need OpenCV; their inputs and expected
values are restated here)."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle
from openvslam_amd import _abi as A

L = oracle.lib()


def _desc(byte):
    return np.full(32, byte, dtype=np.uint8)


# ---- test/openvslam/match/base.cc:11-57 ------------------------------------------------------
@pytest.mark.parametrize("a,b,expected", [(0x55, 0x55, 0), (0x55, 0xAA, 256), (0x66, 0x3C, 128)])
def test_hamming_known_answers(a, b, expected):
    da, db = _desc(a), _desc(b)
    assert L.ovo_hamming_32(A.ptr(da, A.u8p), A.ptr(db, A.u8p)) == expected
    assert L.ovo_hamming_64(A.ptr(da, A.u8p), A.ptr(db, A.u8p)) == expected


def test_hamming_random_vs_numpy():
    rng = np.random.default_rng(1)
    a = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (200, 32), dtype=np.uint8)
    out = np.zeros(200, dtype=np.uint32)
    oracle.check(L.ovo_hamming_pairs(A.ptr(a, A.u8p), A.ptr(b, A.u8p), 200, A.ptr(out, A.u32p)))
    ref = np.unpackbits(a ^ b, axis=1).sum(axis=1)
    assert np.array_equal(out, ref)
    for i in range(20):
        assert L.ovo_hamming_64(A.ptr(a[i], A.u8p), A.ptr(b[i], A.u8p)) == ref[i]


# ---- test/openvslam/match/angle_checker.cc:12-180 --------------------------------------------
ANGLE_CASES = [(34.8, 35), (34.9, 35), (35.0, 35), (35.1, 35), (35.2, 35),
               (323.8, 324), (323.9, 324), (324.1, 324), (324.2, 324),
               (126.9, 127), (127.0, 127), (127.1, 127),
               (0.0, 0), (90.0, 60), (180.0, 180), (270.0, 270)]


def _angle_partition(num_bins_thr):
    c = L.ovo_angle_checker_create(30, num_bins_thr)
    assert c
    for ang, m in ANGLE_CASES:
        L.ovo_angle_checker_append(c, ang, m)
    buf = np.zeros(64, dtype=np.int32)
    nv = L.ovo_angle_checker_valid(c, A.ptr(buf, A.i32p), 64)
    valid = buf[:nv].tolist()
    ni = L.ovo_angle_checker_invalid(c, A.ptr(buf, A.i32p), 64)
    invalid = buf[:ni].tolist()
    L.ovo_angle_checker_destroy(c)
    return valid, invalid


@pytest.mark.parametrize("thr,valid_set", [(1, {35}), (2, {35, 324}), (3, {35, 324, 127}),
                                           (30, {35, 324, 127, 0, 60, 180, 270})])
def test_angle_checker_partitions(thr, valid_set):
    valid, invalid = _angle_partition(thr)
    every = {35, 324, 127, 0, 60, 180, 270}
    assert valid and set(valid) == valid_set
    assert set(invalid) == every - valid_set
    assert len(valid) + len(invalid) == len(ANGLE_CASES)


def test_angle_checker_asserts():
    assert L.ovo_angle_checker_create(30, 31) is None  # assert(num_bins_thr_ <= histogram_length_)
    for ok in (3, 30):
        c = L.ovo_angle_checker_create(30, ok)
        assert c
        L.ovo_angle_checker_destroy(c)


# ---- test/openvslam/feature/orb_params.cc:152-196 --------------------------------------------
@pytest.mark.parametrize("num_levels,scale", [(10, 1.26), (8, 1.2), (1, 1.2), (5, 2.0)])
def test_scale_tables(num_levels, scale):
    sf = oracle.scale_table("scale_factors", num_levels, scale)
    isf = oracle.scale_table("inv_scale_factors", num_levels, scale)
    sig = oracle.scale_table("level_sigma_sq", num_levels, scale)
    isig = oracle.scale_table("inv_level_sigma_sq", num_levels, scale)
    s = float(np.float32(scale))
    for lvl in range(num_levels):
        # EXPECT_FLOAT_EQ = within 4 ulp
        assert sf[lvl] == pytest.approx(np.float32(s ** lvl), rel=4 * 2 ** -23)
        assert isf[lvl] == pytest.approx(np.float32(s ** -lvl), rel=4 * 2 ** -23)
        assert sig[lvl] == pytest.approx(np.float32(s ** (2 * lvl)), rel=4 * 2 ** -23)
        assert isig[lvl] == pytest.approx(np.float32(s ** (-2 * lvl)), rel=4 * 2 ** -23)


# ---- test/openvslam/feature/orb_params.cc:62-150 (mask rectangle validation) ------------------
@pytest.mark.parametrize("rect,msg", [([0.5, 0.2, 0.0, 1.0], "x_max must be greater than x_min"),
                                      ([0.2, 0.2, 0.0, 1.0], "x_max must be greater than x_min"),
                                      ([0.0, 1.0, 0.7, 0.3], "y_max must be greater than x_min"),
                                      ([0.0, 1.0, 0.3, 0.3], "y_max must be greater than x_min")])
def test_mask_rect_validation(rect, msg):
    p = A.make_orb_params(mask_rects=[rect])
    h = C.c_void_p()
    rc = L.ovo_orb_create(p.ref(), C.byref(h))
    assert rc == A.OVS_ERR_INVALID_ARGUMENT
    assert L.ovo_last_error().decode() == msg


def test_mask_rect_valid():
    p = A.make_orb_params(mask_rects=[[0.0, 1.0, 0.0, 0.1], [0.2, 0.5, 0.3, 0.9]])
    h = C.c_void_p()
    assert L.ovo_orb_create(p.ref(), C.byref(h)) == 0
    L.ovo_orb_destroy(h)


# ---- test/openvslam/util/trigonometric.cc:8-20 -----------------------------------------------
def test_trig_tolerance():
    for k in range(0, 3601):
        deg = k * 0.1
        rad = np.float32(deg * math.pi / 180.0)
        assert abs(L.ovo_cos(rad) - math.cos(rad)) < 1e-3
        assert abs(L.ovo_sin(rad) - math.sin(rad)) < 1e-3


# ---- test/openvslam/data/common_get_cell_indices.cc:59-111 (zero-distortion camera) ----------
def _cam_2000x1000():
    # perspective camera without distortion: image bounds are the image rectangle
    return A.make_grid(0.0, 0.0, 2000.0, 1000.0)


def _cell(g, x, y):
    cx, cy = C.c_int(-1), C.c_int(-1)
    found = L.ovo_get_cell_indices(C.byref(g), np.float32(x), np.float32(y), C.byref(cx), C.byref(cy))
    return found, cx.value, cy.value


def test_get_cell_indices_valid_cases_2():
    g = _cam_2000x1000()
    eps = np.float32(0.01)
    cw, ch = 1.0 / g.inv_cell_width, 1.0 / g.inv_cell_height
    for ix in range(64):
        for iy in range(48):
            corners = [(np.float32(ix * cw + eps), np.float32(iy * ch + eps)),
                       (np.float32((ix + 1) * cw - eps), np.float32(iy * ch + eps)),
                       (np.float32(ix * cw + eps), np.float32((iy + 1) * ch - eps)),
                       (np.float32((ix + 1) * cw - eps), np.float32((iy + 1) * ch - eps))]
            for x, y in corners:
                assert _cell(g, x, y) == (1, ix, iy)


def distorted_camera_2000x1000():
    """create_perspective_camera(2000, 1000, -0.1, 0.1) of the reference test (common_get_cell_indices.cc:8-14):
    fx = fy = rows, cx = cols / 2, cy = rows / 2, k1 and k2 handed over as floats."""
    cam = A.Camera()
    cam.model = A.CAMERA_PERSPECTIVE
    cam.cols, cam.rows = 2000, 1000
    cam.fx = cam.fy = 1000.0
    cam.cx, cam.cy = 1000.0, 500.0
    for i, v in enumerate((float(np.float32(-0.1)), float(np.float32(0.1)), 0.0, 0.0, 0.0)):
        cam.dist[i] = v
    return cam


def image_bounds_grid(undistort, cam):
    """camera::perspective::compute_image_bounds (camera/perspective.cc:72-97) + the cell sizes of
    perspective.cc:29-30: the four image corners are undistorted, the bounds are their extremes."""
    corners = np.zeros(4, dtype=A.KEYPOINT_DTYPE)
    corners["x"] = [0.0, cam.cols, 0.0, cam.cols]
    corners["y"] = [0.0, 0.0, cam.rows, cam.rows]
    corners["size"] = 1.0
    und = undistort(cam, corners)
    return A.make_grid(min(und["x"][0], und["x"][2]), min(und["y"][0], und["y"][1]),
                       max(und["x"][1], und["x"][3]), max(und["y"][2], und["y"][3]))


def cell_index_cases(g, cols=2000, rows=1000):
    """valid_cases_1 (common_get_cell_indices.cc:16-57) and invalid_cases (:113-144): keypoint
    coordinates (float arithmetic as in the C++ initialisers) and the expected cells."""
    eps = np.float32(0.01)
    f = np.float32
    nc, nr = g.num_cols, g.num_rows
    min_x, max_x, min_y, max_y = f(g.min_x), f(g.max_x), f(g.min_y), f(g.max_y)
    valid = [((min_x, min_y), (0, 0)), ((max_x - eps, min_y), (nc - 1, 0)), ((min_x, max_y - eps), (0, nr - 1)),
             ((max_x - eps, max_y - eps), (nc - 1, nr - 1)),
             ((f(cols / 2.0), min_y), (nc // 2 - 1, 0)), ((f(cols / 2.0), max_y - eps), (nc // 2 - 1, nr - 1)),
             ((min_x, f(rows / 2.0)), (0, nr // 2 - 1)), ((max_x - eps, f(rows / 2.0)), (nc - 1, nr // 2 - 1))]
    invalid = [(min_x - eps, min_y - eps), (max_x, min_y - eps), (min_x - eps, max_y), (max_x, max_y),
               (f(cols / 2.0), min_y - eps), (f(cols / 2.0), max_y), (min_x - eps, f(rows / 2.0)), (max_x, f(rows / 2.0))]
    return valid, invalid


def _oracle_undistort(cam, kps):
    out = np.zeros(len(kps), dtype=A.KEYPOINT_DTYPE)
    assert L.ovo_undistort_keypoints(C.byref(cam), A.ptr(kps, A.kpp), len(kps), A.ptr(out, A.kpp)) == 0
    return out


def test_get_cell_indices_distorted_camera_valid_and_invalid_cases():
    """test/openvslam/data/common_get_cell_indices.cc:16-57,113-144 as written: the 2000 x 1000 camera
    with k1 = -0.1, k2 = 0.1, whose image bounds come from undistorting the image corners.  The top
    and bottom centre cases expect cell num_grid_cols / 2 - 1 = 31, which only holds because the
    undistorted bounds are not exactly symmetric in float: the case pins the undistortion, too."""
    cam = distorted_camera_2000x1000()
    g = image_bounds_grid(_oracle_undistort, cam)
    assert g.min_x > 0 and g.max_x < 2000 and g.min_y > 0 and g.max_y < 1000   # barrel model: the corners move inwards
    valid, invalid = cell_index_cases(g)
    for (x, y), (ix, iy) in valid:
        assert _cell(g, x, y) == (1, ix, iy), (x, y)
    for x, y in invalid:
        assert _cell(g, x, y)[0] == 0, (x, y)


# ---- OpenCV helper semantics ------------------------------------------------------------------
def test_cv_round_floor_ceil():
    assert [L.ovo_cv_round_f(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]
    assert [L.ovo_cv_round_d(v) for v in (0.5, 1.5, 2.5, -2.5)] == [0, 2, 2, -2]
    assert [L.ovo_cv_floor_f(v) for v in (1.0, 1.9, -0.1, -1.0, -1.1)] == [1, 1, -1, -1, -2]
    assert [L.ovo_cv_ceil_f(v) for v in (1.0, 1.1, -0.1, -1.0, -1.9)] == [1, 2, 0, -1, -1]
    assert [L.ovo_cv_ceil_d(v) for v in (150.5, 150.0)] == [151, 150]


def test_fast_atan2_accuracy():
    rng = np.random.default_rng(3)
    for _ in range(2000):
        y, x = rng.normal(size=2) * 1e4
        a = L.ovo_fast_atan2(np.float32(y), np.float32(x))
        ref = math.degrees(math.atan2(np.float32(y), np.float32(x))) % 360.0
        d = abs(a - ref)
        assert min(d, 360 - d) < 0.3
        assert 0.0 <= a <= 360.0
    assert L.ovo_fast_atan2(0.0, 0.0) == 0.0
    assert L.ovo_fast_atan2(0.0, 1.0) == 0.0
    assert abs(L.ovo_fast_atan2(1.0, 0.0) - 90.0) < 1e-3


def test_gaussian_kernel():
    k = np.zeros(7, dtype=np.int32)
    L.ovo_gaussian_kernel_7_q8(A.ptr(k, A.i32p))
    assert k.tolist() == [18, 34, 49, 55, 49, 34, 18]


def test_gaussian_kernel_normalised_variant():
    """The coefficient set of OpenCV >= 3.4.9 / 4.2.0 (error diffusion, sum exactly 256) and what the two sets
    do to an image: identical arithmetic, so a constant image stays constant under the normalised set while
    the rounded set (sum 257, gain 257^2 / 2^16 = 1.0078) lifts bright constants by up to two grey levels; on
    noise the two differ by at most two grey levels.  Parity unpinned: no OpenCV here to blur with."""
    k = np.zeros(7, dtype=np.int32)
    L.ovo_gaussian_kernel_7_q8_normalised(A.ptr(k, A.i32p))
    assert k.tolist() == [18, 34, 48, 56, 48, 34, 18] and k.sum() == 256
    # by hand: 17.9608 -> 18 (err -0.0392); 33.5552 - 0.0392 -> 34 (err -0.4840); 48.8225 - 0.4840 -> 48; centre 256 - 200
    rng = np.random.default_rng(1)
    for img in (np.full((40, 50), 200, np.uint8), np.full((40, 50), 77, np.uint8), rng.integers(0, 256, (64, 80), dtype=np.uint8)):
        out = [np.zeros_like(img) for _ in range(2)]
        for variant in (0, 1):
            oracle.check(L.ovo_gaussian_blur_7x7_variant(A.ptr(img, A.u8p), img.shape[1], img.shape[0], img.shape[1], A.ptr(out[variant], A.u8p),
                                                         img.shape[1], variant))
        plain = np.zeros_like(img)
        oracle.check(L.ovo_gaussian_blur_7x7(A.ptr(img, A.u8p), img.shape[1], img.shape[0], img.shape[1], A.ptr(plain, A.u8p), img.shape[1]))
        assert np.array_equal(plain, out[0])
        assert np.abs(out[0].astype(int) - out[1].astype(int)).max() <= 2
        if img.min() == img.max():
            v = int(img[0, 0])
            assert (out[1] == v).all()
            assert (out[0] == (v * 257 * 257 + 32768) >> 16).all()  # 200 -> 202, 77 -> 78
    # brute force of the double sum on one pixel, reflected borders included
    img = rng.integers(0, 256, (9, 9), dtype=np.uint8)
    got = np.zeros_like(img)
    oracle.check(L.ovo_gaussian_blur_7x7_variant(A.ptr(img, A.u8p), 9, 9, 9, A.ptr(got, A.u8p), 9, 1))
    refl = lambda p, n: -p if p < 0 else (2 * n - 2 - p if p >= n else p)  # noqa: E731
    for (y, x) in ((0, 0), (4, 4), (8, 1)):
        s = sum(int(k[j]) * int(k[i]) * int(img[refl(y + j - 3, 9), refl(x + i - 3, 9)]) for j in range(7) for i in range(7))
        assert got[y, x] == min((s + 32768) >> 16, 255)

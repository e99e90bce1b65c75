"""The reference's extractor test cases (test/openvslam/feature/orb_extractor.cc) on the GPU: its two
test images with and without masks, bit-exact against the CPU oracle, plus the committed
oracle outputs of tests/golden/extract_320x240.npz."""
import numpy as np
import pytest

import oracle
import refdata
from openvslam_amd.feature import orb_extractor, orb_params

pytestmark = pytest.mark.gpu


def both(img, mask=None, **params):
    want_k, want_d = oracle.OrbExtractor(**params).extract(img, mask)
    got_k, got_d = orb_extractor(orb_params(**params)).extract(img, mask)
    assert len(got_k) == len(want_k)
    assert got_k.tobytes() == want_k.tobytes()
    assert np.array_equal(got_d, want_d)
    return got_k


@pytest.mark.parametrize("n,rotate", [(1, False), (2, False), (1, True), (2, True)])
def test_equirectangular_without_mask(n, rotate):
    img = refdata.equirectangular(n)
    if rotate:
        img = np.ascontiguousarray(np.rot90(img, -1))
    assert len(both(img)) >= 2000


def test_equirectangular_with_image_masks():
    img = refdata.equirectangular(1)
    h, w = img.shape
    mask = np.ones_like(img)
    mask[:h // 4] = 0
    mask[3 * h // 4:h - 1] = 0
    k = both(img, mask)
    assert k["y"].min() >= h // 4 and k["y"].max() <= 3 * h // 4
    yy, xx = np.mgrid[0:h, 0:w]
    mask = ((xx - w // 2) ** 2 + (yy - h // 2) ** 2 > 321 ** 2).astype(np.uint8)
    k = both(img, mask)
    assert np.all((k["x"] - w // 2) ** 2 + (k["y"] - h // 2) ** 2 >= 320 * 320)


def test_equirectangular_with_rectangle_masks():
    img = refdata.equirectangular(2)
    h, w = img.shape
    rects = [(0.0, 0.2, 0.0, 1.0), (0.8, 1.0, 0.0, 1.0), (0.0, 1.0, 0.0, 0.2), (0.0, 1.0, 0.8, 1.0)]
    k = both(img, mask_rects=rects)
    assert k["x"].min() >= w // 5 and k["x"].max() <= 4 * w // 5
    assert k["y"].min() >= h // 5 and k["y"].max() <= 4 * h // 5


def test_committed_golden_outputs():
    g = np.load(refdata.GOLDEN / "extract_320x240.npz")
    ext = orb_extractor(orb_params(max_num_keypts=400))
    k0, d0 = ext.extract(g["frame0"])
    assert np.array_equal(ext.image_pyramid(3), g["pyramid_level3"])
    assert np.array_equal(ext.blurred_level(2), g["blurred_level2"])
    assert k0.tobytes() == g["kps0"].tobytes() and np.array_equal(d0, g["desc0"])
    k1, d1 = ext.extract(g["frame1"])
    assert k1.tobytes() == g["kps1"].tobytes() and np.array_equal(d1, g["desc1"])


def test_equirectangular_4000_keypoints_with_tutorial_masks():
    """The 1920x960 THETA-style configuration: 4000 keypoints, the rectangle masks of
    docs/simple_tutorial.rst:297-301, and an image mask on top (bit-exact against the oracle)."""
    img = refdata.equirectangular(1)
    h, w = img.shape
    rects = [(0.0, 1.0, 0.0, 0.1), (0.0, 1.0, 0.84, 1.0), (0.0, 0.2, 0.7, 1.0), (0.8, 1.0, 0.7, 1.0)]
    k = both(img, max_num_keypts=4000, mask_rects=rects)
    assert len(k) >= 4000
    assert k["y"].min() >= int(0.1 * h) and k["y"].max() <= int(0.84 * h) + 1
    yy, xx = np.mgrid[0:h, 0:w]
    mask = ((xx - w // 2) ** 2 + (yy - h // 2) ** 2 > 200 ** 2).astype(np.uint8) * 255
    k = both(img, mask, max_num_keypts=4000)   # an image mask overrides the rectangle mask
    assert len(k) >= 4000
    assert np.all((k["x"] - w // 2) ** 2 + (k["y"] - h // 2) ** 2 >= 199 ** 2)


# ---- test/openvslam/match/angle_checker.cc:12-180 on the device --------------------------------------
@pytest.mark.parametrize("thr,valid_set", [(1, {35}), (2, {35, 324}), (3, {35, 324, 127}), (30, {35, 324, 127, 0, 60, 180, 270})])
def test_angle_checker_partitions_on_the_device(thr, valid_set):
    """The reference's four partition cases (top 1 / 2 / 3 / all bins) through the device code of
    match::angle_checker (the same histogram, bin rounding and sort the matcher kernels use)."""
    import ctypes as C
    from test_oracle_known_answers import ANGLE_CASES
    from openvslam_amd import _abi as A
    from openvslam_amd import _lib
    angles = np.array([a for a, _ in ANGLE_CASES], dtype=np.float32)
    matches = [m for _, m in ANGLE_CASES]
    flags = np.full(len(angles), 7, dtype=np.uint8)
    _lib.check(_lib.lib().ovs_angle_checker_partition(A.ptr(angles, A.f32p), len(angles), thr, A.ptr(flags, A.u8p), 0))
    valid = {m for m, f in zip(matches, flags) if f == 1}
    invalid = {m for m, f in zip(matches, flags) if f == 0}
    assert set(flags.tolist()) <= {0, 1}
    assert valid == valid_set and invalid == {35, 324, 127, 0, 60, 180, 270} - valid_set


@pytest.mark.parametrize("seed,n", [(1, 500), (2, 31), (3, 5000)])
def test_angle_checker_random_against_oracle(seed, n):
    """Equally full bins are common here: the order std::sort leaves them in decides the partition."""
    from openvslam_amd import _abi as A
    from openvslam_amd import _lib
    rng = np.random.default_rng(seed)
    angles = (rng.integers(0, 12, n) * 30 + rng.uniform(-14, 14, n)).astype(np.float32)
    angles[rng.random(n) < 0.2] -= 360.0
    for thr in (1, 3, 7):
        got = np.full(n, 7, dtype=np.uint8)
        want = np.full(n, 9, dtype=np.uint8)
        _lib.check(_lib.lib().ovs_angle_checker_partition(A.ptr(angles, A.f32p), n, thr, A.ptr(got, A.u8p), 0))
        oracle.check(oracle.lib().ovo_angle_checker_partition(A.ptr(angles, A.f32p), n, thr, A.ptr(want, A.u8p)))
        assert np.array_equal(got, want)


# ---- test/openvslam/data/common_get_cell_indices.cc:16-57,113-144 on the device -----------------------
def test_get_cell_indices_distorted_camera_on_the_device():
    """The reference's distorted 2000 x 1000 camera: image bounds from the DEVICE undistortion of the
    image corners (camera/perspective.cc:72-97), then the valid and invalid keypoint cases."""
    import ctypes as C
    from test_oracle_known_answers import cell_index_cases, distorted_camera_2000x1000, image_bounds_grid
    from openvslam_amd import _abi as A
    from openvslam_amd import _lib
    L = _lib.lib()

    def undistort(cam, kps):
        out = np.zeros(len(kps), dtype=A.KEYPOINT_DTYPE)
        _lib.check(L.ovs_undistort_keypoints(C.byref(cam), A.ptr(kps, A.kpp), len(kps), A.ptr(out, A.kpp), 0))
        return out
    g = image_bounds_grid(undistort, distorted_camera_2000x1000())
    valid, invalid = cell_index_cases(g)

    def cell(x, y):
        cx, cy = C.c_int(-1), C.c_int(-1)
        found = L.ovs_get_cell_indices(C.byref(g), np.float32(x), np.float32(y), C.byref(cx), C.byref(cy))
        return found, cx.value, cy.value
    for (x, y), (ix, iy) in valid:
        assert cell(x, y) == (1, ix, iy), (x, y)
    for x, y in invalid:
        assert cell(x, y)[0] == 0, (x, y)

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

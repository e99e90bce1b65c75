"""The reference's own extractor tests (test/openvslam/feature/orb_extractor.cc), run against the
CPU oracle: toy rectangles whose only corner is known, the two equirectangular test images with
image masks and rectangle masks.  These are the known answers the reference holds for extract();
it holds no golden keypoints or descriptors."""
import numpy as np
import pytest

import oracle
import refdata


def basic_checks(kps, desc):
    assert len(kps) > 0 and len(desc) == len(kps) and desc.dtype == np.uint8 and desc.shape[1] == 32


# orb_extractor.cc:24-49, 51-76, 334-359 of the reference's test: (w, h, rect, corner)
@pytest.mark.parametrize("w,h,rect,corner", [
    (600, 600, (300, 300, 600, 600), (300, 300)),
    (2000, 2000, (0, 0, 1800, 1800), (1800, 1800)),
    (600, 1200, (300, 600, 600, 1200), (300, 600)),
])
def test_extract_toy_samples(w, h, rect, corner):
    ext = oracle.OrbExtractor()
    kps, desc = ext.extract(refdata.toy_sample(w, h, *rect))
    basic_checks(kps, desc)
    sf = oracle.scale_table("scale_factors", 8, 1.2)
    tol = 2.0 * sf[kps["octave"]]
    assert np.all(np.abs(kps["x"] - corner[0]) <= tol)
    assert np.all(np.abs(kps["y"] - corner[1]) <= tol)


@pytest.mark.parametrize("n,rotate", [(1, False), (2, False), (1, True), (2, True)])
def test_extract_without_mask(n, rotate):
    img = refdata.equirectangular(n)
    if rotate:  # cv::ROTATE_90_CLOCKWISE
        img = np.ascontiguousarray(np.rot90(img, -1))
    kps, desc = oracle.OrbExtractor().extract(img)
    basic_checks(kps, desc)
    assert len(kps) >= 2000
    h, w = img.shape
    assert kps["x"].min() >= 0 and kps["x"].max() < w and kps["y"].min() >= 0 and kps["y"].max() < h
    assert np.all((0 <= kps["angle"]) & (kps["angle"] < 360))
    # unsigned int scaled_patch_size = fast_patch_size_ * scale_factors_.at(level)  (orb_extractor.cc:380)
    assert np.all(kps["size"] == np.floor(np.float32(31.0) * oracle.scale_table("scale_factors", 8, 1.2)[kps["octave"]]))


def test_extract_with_image_mask_rows():
    img = refdata.equirectangular(1)
    h, w = img.shape
    mask = np.ones_like(img)
    mask[:h // 4] = 0
    mask[3 * h // 4:h - 1] = 0
    kps, desc = oracle.OrbExtractor().extract(img, mask)
    basic_checks(kps, desc)
    assert kps["y"].min() >= h // 4 and kps["y"].max() <= 3 * h // 4


def test_extract_with_image_mask_cols():
    img = refdata.equirectangular(2)
    h, w = img.shape
    mask = np.ones_like(img)
    mask[:, :w // 4] = 0
    mask[:, 3 * w // 4:w - 1] = 0
    kps, desc = oracle.OrbExtractor().extract(img, mask)
    basic_checks(kps, desc)
    assert kps["x"].min() >= w // 4 and kps["x"].max() <= 3 * w // 4


def test_extract_with_image_mask_circle():
    img = refdata.equirectangular(1)
    h, w = img.shape
    yy, xx = np.mgrid[0:h, 0:w]
    radius = 320
    # one pixel wider than the tested radius, as the anti-aliased cv::circle edge is
    mask = ((xx - w // 2) ** 2 + (yy - h // 2) ** 2 > (radius + 1) ** 2).astype(np.uint8)
    kps, desc = oracle.OrbExtractor().extract(img, mask)
    basic_checks(kps, desc)
    assert np.all((kps["x"] - w // 2) ** 2 + (kps["y"] - h // 2) ** 2 >= radius * radius)


@pytest.mark.parametrize("n,rects,axes", [
    (1, [(0.0, 1.0, 0.0, 0.2), (0.0, 1.0, 0.8, 1.0)], "y"),
    (2, [(0.0, 0.2, 0.0, 1.0), (0.8, 1.0, 0.0, 1.0)], "x"),
    (2, [(0.0, 0.2, 0.0, 1.0), (0.8, 1.0, 0.0, 1.0), (0.0, 1.0, 0.0, 0.2), (0.0, 1.0, 0.8, 1.0)], "xy"),
])
def test_extract_with_rectangle_mask(n, rects, axes):
    img = refdata.equirectangular(n)
    h, w = img.shape
    kps, desc = oracle.OrbExtractor(mask_rects=rects).extract(img)
    basic_checks(kps, desc)
    if "x" in axes:
        assert kps["x"].min() >= w // 5 and kps["x"].max() <= 4 * w // 5
    if "y" in axes:
        assert kps["y"].min() >= h // 5 and kps["y"].max() <= 4 * h // 5


def test_oracle_reproduces_its_committed_outputs():
    """Regression pin: tests/golden/extract_320x240.npz (tools/gen_golden.py) are outputs of this
    oracle, not of the reference."""
    g = np.load(refdata.GOLDEN / "extract_320x240.npz")
    ext = oracle.OrbExtractor(max_num_keypts=400)
    k0, d0 = ext.extract(g["frame0"])
    assert np.array_equal(ext.pyramid_level(3), g["pyramid_level3"])
    assert np.array_equal(ext.blurred_level(2), g["blurred_level2"])
    assert k0.tobytes() == g["kps0"].tobytes() and np.array_equal(d0, g["desc0"])
    k1, d1 = ext.extract(g["frame1"])
    assert k1.tobytes() == g["kps1"].tobytes() and np.array_equal(d1, g["desc1"])

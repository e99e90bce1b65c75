"""GPU parity of the ORB extractor against the CPU oracle, through the C ABI.

Bit-exact bar: pyramid pixels, blurred pixels, keypoints (x, y, size, angle, response, octave) and
descriptors must be identical to the oracle's on the same inputs.  (The north-star tolerance of
99.9 % descriptor bits is for float angle rounding; oracle and device evaluate the same fp32
operation sequence, so the tests require 100 %.)
"""
import numpy as np
import pytest

import oracle
from openvslam_amd import synth
from openvslam_amd.feature import orb_extractor, orb_params

pytestmark = pytest.mark.gpu


def _compare(img, mask=None, **params):
    ref = oracle.OrbExtractor(**params)
    rk, rd = ref.extract(img, mask)
    ext = orb_extractor(orb_params(**params))
    gk, gd = ext.extract(img, mask)
    levels = params.get("num_levels", 8)
    for lvl in range(1, levels):
        assert np.array_equal(ext.image_pyramid(lvl), ref.pyramid_level(lvl)), "pyramid level %d differs" % lvl
    for lvl in range(levels):
        rb = ref.blurred_level(lvl)
        if rb is not None:
            assert np.array_equal(ext.blurred_level(lvl), rb), "blurred level %d differs" % lvl
    assert len(gk) == len(rk), "keypoint count %d vs %d" % (len(gk), len(rk))
    for field in ("octave", "x", "y", "response", "size", "class_id"):
        assert np.array_equal(gk[field], rk[field]), "keypoint field %s differs" % field
    assert np.array_equal(gk["angle"].view(np.uint32), rk["angle"].view(np.uint32)), "angles differ"
    assert np.array_equal(gd, rd), "descriptors differ"
    return gk, gd


@pytest.mark.parametrize("w,h,n", [(640, 480, 1000), (1241, 376, 2000), (752, 480, 1000)])
def test_extract_matches_oracle(w, h, n):
    img = synth.make_frame(w, h, seed=1234 + w)
    gk, _ = _compare(img, max_num_keypts=n)
    assert len(gk) >= n // 2


def test_extract_low_texture_uses_min_threshold():
    # faint texture: most cells are empty at the initial threshold and fall back to the minimum one
    rng = np.random.default_rng(7)
    img = np.clip(128 + rng.normal(0, 3.5, (480, 640)), 0, 255).astype(np.uint8)
    img[200:280, 300:420] += 30
    _compare(img, max_num_keypts=1000)


@pytest.mark.parametrize("variant", [0, 1])
@pytest.mark.parametrize("w", [253, 254, 255, 256, 257, 259, 260, 261, 262, 263, 511, 514, 517, 520, 771])
def test_blur_borders_every_width_class(w, variant):
    # the blur's tiles are 256 columns wide: widths around the tile boundaries put the right image
    # border at every byte position of a lane's dwords (reflected columns inside the lane's own dword,
    # in its right neighbour, in the first dword of the next tile), with one, two and three column tiles;
    # level widths w / 1.2^k add further residues, and odd widths put the rows of the uploaded image at
    # odd addresses.  Noise: every pixel differs from its neighbours.  Both coefficient sets of
    # cv::GaussianBlur (ovs_orb_set_blur_kernel: rounded, sum 257; normalised, sum 256).
    rng = np.random.default_rng(w)
    img = rng.integers(0, 256, (97, w), dtype=np.uint8)
    ref = oracle.OrbExtractor(max_num_keypts=300, num_levels=3)
    ref.set_blur_kernel(variant)
    ref.extract(img, None)
    ext = orb_extractor(orb_params(max_num_keypts=300, num_levels=3))
    ext.set_blur_kernel(variant)
    assert ext.get_blur_kernel() == variant
    ext.extract(img, None)
    for lvl in range(3):
        rb = ref.blurred_level(lvl)
        assert rb is not None
        assert np.array_equal(ext.blurred_level(lvl), rb), "blurred level %d differs at width %d" % (lvl, w)


def test_blur_kernel_variants_end_to_end():
    """The normalised coefficient set (OpenCV >= 3.4.9 / 4.2.0) through the whole extraction: keypoints are
    the same as with the rounded set (the blur only feeds the descriptors), descriptors equal the oracle's for
    that set and differ from the other set's in a few bits; bright saturated areas stay at 255 in both."""
    img = synth.make_frame(752, 480, seed=77)
    img[:40, :60] = 255
    out = {}
    for variant in (0, 1):
        ref = oracle.OrbExtractor(max_num_keypts=1000)
        ref.set_blur_kernel(variant)
        rk, rd = ref.extract(img)
        ext = orb_extractor(orb_params(max_num_keypts=1000))
        ext.set_blur_kernel(variant)
        gk, gd = ext.extract(img)
        assert gk.tobytes() == rk.tobytes() and np.array_equal(gd, rd)
        assert np.array_equal(ext.blurred_level(0), ref.blurred_level(0)) and ext.blurred_level(0)[:30, :50].min() == 255
        out[variant] = (gk, gd)
    assert out[0][0].tobytes() == out[1][0].tobytes()
    differing = np.unpackbits(out[0][1] ^ out[1][1]).sum()
    assert 0 < differing < 0.05 * out[0][1].size * 8
    from openvslam_amd._lib import OvsError
    with pytest.raises(OvsError):
        orb_extractor(orb_params()).set_blur_kernel(7)


@pytest.mark.parametrize("levels,n,w,h", [(1, 3000, 1241, 376), (2, 2500, 640, 480), (8, 8000, 1241, 376),
                                          # example/tum_rgbd/TUM_RGBD_rgbd_4.yaml:33 (10000 keypoints at 640x480), which
                                          # is also the doubled initialiser quota of TUM_RGBD_mono_1.yaml:32
                                          # (2 x 5000, tracking_module.cc:29-36)
                                          (8, 10000, 640, 480)])
def test_extract_large_quota_per_level(levels, n, w, h):
    # per-level quotas beyond what the quadtree's node arrays hold in LDS (about 1130 keypoints): those
    # levels run the kernel instance that keeps the arrays in a global workspace; with eight levels and
    # 8000 keypoints the first three levels take that path and the rest the LDS one
    rng = np.random.default_rng(levels * 100 + n)
    img = rng.integers(0, 256, (h, w), dtype=np.uint8)
    gk, _ = _compare(img, max_num_keypts=n, num_levels=levels)
    assert len(gk) >= n // 2


def test_extract_large_quota_doubled_by_the_initialiser():
    # tracking_module.cc:29-36 builds the initialisation extractor with twice the configured keypoints:
    # one handle at 5000, a second at 2 x 5000, the same frame through both, then the quota changed on
    # a live handle (orb_extractor.cc:150-153 re-initialises)
    img = synth.make_frame(640, 480, seed=4242)
    _compare(img, max_num_keypts=5000)
    _compare(img, max_num_keypts=10000)
    ext = orb_extractor(orb_params(max_num_keypts=1000))
    ext.extract(img)
    ext.set_max_num_keypoints(10000)
    gk, gd = ext.extract(img)
    rk, rd = oracle.OrbExtractor(max_num_keypts=10000).extract(img)
    assert np.array_equal(gk.view(np.uint8), rk.view(np.uint8)) and np.array_equal(gd, rd)


def test_extract_large_quota_batch():
    # every (level, frame) workgroup of the workspace instance owns one slice of the node workspace
    import torch
    from openvslam_amd import _abi as A
    dev = torch.device("cuda", 0)
    w, h, n_frames, n_kp = 640, 480, 5, 10000
    rng = np.random.default_rng(5)
    frames = rng.integers(0, 256, (n_frames, h, w), dtype=np.uint8)
    frames[3] = synth.make_frame(w, h, seed=9)
    ext = orb_extractor(orb_params(max_num_keypts=n_kp), device=0)
    cap = ext.keypoint_capacity(w, h)
    kps = torch.zeros((n_frames, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.zeros((n_frames, cap, 32), dtype=torch.uint8, device=dev)
    counts = torch.zeros((n_frames,), dtype=torch.int32, device=dev)
    d_frames = torch.from_numpy(frames).to(dev)
    ext.extract_batch_dev(d_frames.data_ptr(), n_frames, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), cap,
                          torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    h_counts = counts.cpu().numpy()
    h_kps = kps.cpu().numpy().view(np.uint8).reshape(n_frames, cap, 28)
    h_desc = desc.cpu().numpy()
    ref = oracle.OrbExtractor(max_num_keypts=n_kp)
    for f in range(n_frames):
        k, d = ref.extract(frames[f])
        assert h_counts[f] == len(k) > 1000
        assert np.array_equal(h_kps[f, :len(k)].reshape(-1), k.view(np.uint8).reshape(-1)), "frame %d keypoints differ" % f
        assert np.array_equal(h_desc[f, :len(k)], d), "frame %d descriptors differ" % f


def test_extract_quota_beyond_the_limit_is_refused():
    with pytest.raises(Exception):
        ext = orb_extractor(orb_params(max_num_keypts=9000, num_levels=1))
        ext.extract(np.zeros((480, 640), np.uint8))


@pytest.mark.parametrize("kind", ["uniform", "salt_pepper", "checker"])
def test_extract_dense_corner_images(kind):
    # images in which most pixels pass the FAST rejection test: the per-cell survivor list runs
    # past the half the compacted lists need, and the dense fallback of the kernel is exercised
    rng = np.random.default_rng(11)
    if kind == "uniform":
        img = rng.integers(0, 256, (240, 320), dtype=np.uint8)
    elif kind == "salt_pepper":
        img = np.where(rng.random((240, 320)) < 0.5, 0, 255).astype(np.uint8)
    else:
        yy, xx = np.mgrid[0:240, 0:320]
        img = (((xx // 2 + yy // 2) & 1) * 200 + rng.integers(0, 40, (240, 320))).astype(np.uint8)
    _compare(img, max_num_keypts=1000)


def test_extract_flat_image_has_no_keypoints():
    img = np.full((480, 640), 77, dtype=np.uint8)
    gk, gd = _compare(img, max_num_keypts=500)
    assert len(gk) == 0 and gd.shape == (0, 32)


def test_extract_toy_samples():
    # test/openvslam/feature/orb_extractor.cc:24-76
    img = np.full((600, 600), 255, dtype=np.uint8)
    img[300:, 300:] = 0
    gk, _ = _compare(img)
    sf = orb_params.calc_scale_factors(8, 1.2)
    assert len(gk) > 0
    assert np.all(np.abs(gk["x"] - 300) <= 2.0 * sf[gk["octave"]])
    assert np.all(np.abs(gk["y"] - 300) <= 2.0 * sf[gk["octave"]])
    img = np.full((2000, 2000), 255, dtype=np.uint8)
    img[:1801, :1801] = 0
    gk, _ = _compare(img)
    assert len(gk) > 0
    assert np.all(np.abs(gk["x"] - 1800) <= 2.0 * sf[gk["octave"]])
    assert np.all(np.abs(gk["y"] - 1800) <= 2.0 * sf[gk["octave"]])


def test_extract_with_image_mask():
    # test/openvslam/feature/orb_extractor.cc:116-141: top and bottom quarter masked
    img = synth.make_frame(960, 480, seed=99)
    mask = np.ones_like(img)
    mask[:img.shape[0] // 4] = 0
    mask[3 * img.shape[0] // 4:img.shape[0] - 1] = 0
    gk, _ = _compare(img, mask)
    assert len(gk) > 0
    assert np.all(gk["y"] >= img.shape[0] // 4) and np.all(gk["y"] <= 3 * img.shape[0] // 4)


def test_extract_with_rectangle_masks():
    # the mask rectangles of the equirectangular tutorial configuration
    rects = [[0.0, 1.0, 0.0, 0.1], [0.0, 1.0, 0.84, 1.0], [0.0, 0.2, 0.7, 1.0], [0.8, 1.0, 0.7, 1.0]]
    img = synth.make_frame(960, 480, seed=5)
    gk, _ = _compare(img, mask_rects=rects)
    assert len(gk) > 0
    h, w = img.shape
    for x0, x1, y0, y1 in rects:
        inside = (gk["x"] >= x0 * w) & (gk["x"] <= x1 * w) & (gk["y"] >= y0 * h) & (gk["y"] <= y1 * h)
        assert not inside.any()


def test_extract_other_parameters():
    img = synth.make_frame(800, 600, seed=321)
    _compare(img, max_num_keypts=300, scale_factor=1.5, num_levels=4, ini_fast_thr=30, min_fast_thr=10)
    _compare(img, max_num_keypts=700, scale_factor=2.0, num_levels=3)
    _compare(img, max_num_keypts=100, num_levels=1)


def test_extract_portrait():
    img = synth.make_frame(376, 900, seed=8)
    _compare(img, max_num_keypts=1500)


def test_empty_image_leaves_outputs_untouched():
    ext = orb_extractor(orb_params())
    assert ext.extract(np.zeros((0, 0), dtype=np.uint8)) == (None, None)


def test_too_small_image_is_rejected():
    from openvslam_amd._lib import OvsError
    ext = orb_extractor(orb_params())
    with pytest.raises(OvsError):
        ext.extract(np.zeros((100, 100), dtype=np.uint8))


def test_extract_from_strided_host_buffers():
    """cv::Mat rows may be padded: the host entry point packs image and mask rows (any stride) before
    the transfer.  Same result as from a contiguous copy."""
    import ctypes as C
    from openvslam_amd import _abi as A
    from openvslam_amd import _lib
    w, h, stride, mstride = 333, 241, 333 + 19, 333 + 5
    img = synth.make_sequence(w, h, 1, seed=77)[0]
    mask = np.ones((h, w), np.uint8)
    mask[: h // 5] = 0
    big = np.full((h, stride), 123, np.uint8)
    big[:, :w] = img
    mbig = np.zeros((h, mstride), np.uint8)
    mbig[:, :w] = mask
    ext = orb_extractor(orb_params(max_num_keypts=700))
    want_k, want_d = ext.extract(img, mask)
    cap = ext.keypoint_capacity(w, h)
    kps = np.zeros(cap, dtype=A.KEYPOINT_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = C.c_int(0)
    _lib.check(_lib.lib().ovs_orb_extract(ext._h, A.ptr(big, A.u8p), w, h, stride, A.ptr(mbig, A.u8p), mstride,
                                          A.ptr(kps, A.kpp), A.ptr(desc, A.u8p), cap, C.byref(n)))
    assert n.value == len(want_k) > 100
    assert np.array_equal(kps[:n.value].view(np.uint8), want_k.view(np.uint8))
    assert np.array_equal(desc[:n.value], want_d)
    # and both equal the oracle
    rk, rd = oracle.OrbExtractor(max_num_keypts=700).extract(img, mask)
    assert np.array_equal(rk.view(np.uint8), want_k.view(np.uint8)) and np.array_equal(rd, want_d)


def test_repeated_single_frame_calls_on_one_handle():
    """One handle through a live session: ten different frames, then a new threshold, another image size,
    a caller mask and the first size again, then a new keypoint target.  Every result equals the oracle's
    for that frame and those parameters (staging buffers, plan and constants are reused or rebuilt as
    needed between the calls)."""
    def check(ext, img, mask=None, **params):
        ref = oracle.OrbExtractor(**params)
        rk, rd = ref.extract(img, mask)
        gk, gd = ext.extract(img, mask)
        assert len(gk) == len(rk) and len(rk) > 100
        assert np.array_equal(gk.view(np.uint8), rk.view(np.uint8)) and np.array_equal(gd, rd)
        assert np.array_equal(ext.image_pyramid(3), ref.pyramid_level(3))

    ext = orb_extractor(orb_params(max_num_keypts=800))
    frames = synth.make_sequence(640, 480, 10, seed=21, dx=7, dy=2)
    for f in frames:
        check(ext, f, max_num_keypts=800)
    ext.set_initial_fast_threshold(35)
    for f in frames[:4]:
        check(ext, f, max_num_keypts=800, ini_fast_thr=35)
    small = synth.make_sequence(400, 300, 4, seed=22, dx=3, dy=3)
    for f in small:
        check(ext, f, max_num_keypts=800, ini_fast_thr=35)
    mask = np.full((300, 400), 255, dtype=np.uint8)
    mask[:, :150] = 0
    check(ext, small[0], mask, max_num_keypts=800, ini_fast_thr=35)
    for f in small[:3]:
        check(ext, f, max_num_keypts=800, ini_fast_thr=35)
    ext.set_max_num_keypoints(500)
    for f in frames[:4]:
        check(ext, f, max_num_keypts=500, ini_fast_thr=35)


def test_row_stride_beyond_the_kernels_offset_arithmetic_is_refused():
    """Row offsets inside a frame are 24-bit x 24-bit / 32-bit products in the blur and descriptor kernels:
    a stride of 2^23 bytes or a frame spanning more than 2^31 bytes is refused, not mis-addressed."""
    import ctypes as C
    import torch
    from openvslam_amd import _lib
    ext = orb_extractor(orb_params(max_num_keypts=500))
    dev = torch.device("cuda", 0)
    img = torch.zeros((64, 64), dtype=torch.uint8, device=dev)
    out = torch.zeros(4096, dtype=torch.uint8, device=dev)
    L = _lib.lib()
    for stride in (1 << 23, (1 << 31) // 479 + 1):
        rc = L.ovs_orb_extract_batch_dev(ext._h, img.data_ptr(), 1, 640, 480, stride, stride * 480, None, 0, out.data_ptr(), out.data_ptr(),
                                         out.data_ptr(), 16, None)
        assert rc == _lib.OVS_ERR_UNSUPPORTED and b"stride" in L.ovs_last_error()

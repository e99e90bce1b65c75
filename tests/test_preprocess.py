"""Image preparation in front of the extractor: util::convert_to_grayscale, util::equalize_histogram
and the cv::remap of util::stereo_rectifier (src/openvslam/util/image_converter.cc,
stereo_rectifier.cc).  CPU part: known answers of the oracle restatement (the reference's tests hold
no pixels for these, so the oracle follows the published OpenCV 3.3.1 algorithms: parity unpinned).
GPU part: bit-exact parity of the HIP kernels with the oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle
from openvslam_amd import _abi as A

RGB, BGR = 1, 2


def _oracle_gray(img, order):
    img = np.ascontiguousarray(img)
    h, w, cn = img.shape
    out = np.zeros((h, w), np.uint8)
    oracle.check(oracle.lib().ovo_convert_to_grayscale(A.ptr(img, A.u8p), w, h, cn, w * cn, order, A.ptr(out, A.u8p), w))
    return out


def _oracle_equalize(img):
    img = np.ascontiguousarray(img)
    h, w = img.shape
    out = np.zeros((h, w), np.uint8)
    rc = oracle.lib().ovo_equalize_histogram(img.ctypes.data_as(C.c_void_p), w, h, w * img.itemsize, 8 * img.itemsize, A.ptr(out, A.u8p), w)
    return rc, out


def _oracle_remap(src, mx, my):
    src = np.ascontiguousarray(src)
    mx = np.ascontiguousarray(mx, np.float32)
    my = np.ascontiguousarray(my, np.float32)
    out = np.zeros(mx.shape, np.uint8)
    oracle.check(oracle.lib().ovo_remap_linear(A.ptr(src, A.u8p), src.shape[1], src.shape[0], src.shape[1], A.ptr(mx, A.f32p), A.ptr(my, A.f32p),
                                               mx.shape[1], mx.shape[0], A.ptr(out, A.u8p), mx.shape[1]))
    return out


def _rectify_maps(w, h, seed):
    """A smooth warp of the kind initUndistortRectifyMap produces (radial term + small rotation),
    reaching a few pixels outside of the source at the corners."""
    rng = np.random.default_rng(seed)
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    cx, cy, f = w / 2 + rng.uniform(-5, 5), h / 2 + rng.uniform(-5, 5), 0.9 * w
    x, y = (xs - cx) / f, (ys - cy) / f
    a = rng.uniform(-0.01, 0.01)
    xr, yr = np.cos(a) * x - np.sin(a) * y, np.sin(a) * x + np.cos(a) * y
    r2 = xr * xr + yr * yr
    k = 1 + rng.uniform(-0.3, -0.1) * r2 + rng.uniform(0.0, 0.1) * r2 * r2
    return (f * 1.08 * xr * k + cx).astype(np.float32), (f * 1.08 * yr * k + cy).astype(np.float32)


# ---- CPU: the oracle's known answers ------------------------------------------------------------------

def test_gray_known_answers():
    # (R 4899 + G 9617 + B 1868 + 8192) >> 14: the primaries, white, and a hand-computed colour
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [10, 200, 90], [0, 0, 0]]], np.uint8)
    assert _oracle_gray(px, RGB)[0].tolist() == [76, 150, 29, 255, (10 * 4899 + 200 * 9617 + 90 * 1868 + 8192) >> 14, 0]
    assert _oracle_gray(px, BGR)[0].tolist() == [29, 150, 76, 255, (90 * 4899 + 200 * 9617 + 10 * 1868 + 8192) >> 14, 0]
    # alpha is ignored
    rgba = np.concatenate([px, np.full((1, 6, 1), 77, np.uint8)], axis=2)
    assert np.array_equal(_oracle_gray(rgba, RGB), _oracle_gray(px, RGB))
    # equal channels stay (the coefficients sum to 2^14)
    ramp = np.repeat(np.arange(256, dtype=np.uint8)[None, :, None], 3, axis=2)
    assert np.array_equal(_oracle_gray(ramp, BGR)[0], np.arange(256))


def test_gray_color_order_gray_is_refused():
    px = np.zeros((2, 2, 3), np.uint8)
    out = np.zeros((2, 2), np.uint8)
    assert oracle.lib().ovo_convert_to_grayscale(A.ptr(px, A.u8p), 2, 2, 3, 6, 0, A.ptr(out, A.u8p), 2) == A.OVS_ERR_UNSUPPORTED


def test_equalize_known_answers():
    # 100 pixels 0..99: l = sorted[5] = 5, h = sorted[95] = 95 -> 255 * (v - 5) / 90, rounded half to even, saturated
    img = np.arange(100, dtype=np.uint8).reshape(10, 10)
    rc, out = _oracle_equalize(img)
    assert rc == 0
    a, b = np.float32(255.0 / 90), np.float32(-255.0 * 5 / 90)
    want = np.clip(np.rint(img.astype(np.float32) * a + b), 0, 255).astype(np.uint8)
    assert np.array_equal(out, want)
    assert out[0, 0] == 0 and out[9, 9] == 255 and out[6, 0] == round(255 * 55 / 90)
    # 16 bit in, 8 bit out
    img16 = (np.arange(100, dtype=np.uint16) * 300).reshape(10, 10)
    rc, out16 = _oracle_equalize(img16)
    assert rc == 0 and out16[0, 0] == 0 and out16[9, 9] == 255
    a, b = np.float32(255.0 / (90 * 300)), np.float32(-255.0 * 1500 / (90 * 300))
    assert np.array_equal(out16, np.clip(np.rint(img16.astype(np.float32) * a + b), 0, 255).astype(np.uint8))
    # a flat image divides by zero in the reference
    rc, _ = _oracle_equalize(np.full((8, 8), 9, np.uint8))
    assert rc == A.OVS_ERR_UNSUPPORTED


def test_remap_weight_table():
    """The 15-bit weights are (32 - fx)(32 - fy) * 32, ...; only the entry for fx = fy = 0 deviates
    (32767 and a stray 1 on the opposite tap), which never changes an 8-bit result."""
    tab = np.zeros(32 * 32 * 4, np.int16)
    oracle.lib().ovo_remap_linear_table(tab.ctypes.data_as(C.POINTER(C.c_int16)))
    tab = tab.reshape(32, 32, 4).astype(np.int64)
    fy, fx = np.meshgrid(np.arange(32), np.arange(32), indexing="ij")
    want = np.stack([(32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32], axis=-1)
    assert tab[0, 0].tolist() == [32767, 0, 0, 1]
    want[0, 0] = tab[0, 0]
    assert np.array_equal(tab, want)
    assert (tab.sum(axis=-1) == 32768).all()
    s00, s11 = np.meshgrid(np.arange(256), np.arange(256), indexing="ij")
    assert np.array_equal((s00 * 32767 + s11 + 16384) >> 15, s00)


def test_remap_known_answers():
    src = np.array([[10, 20, 30], [40, 50, 60], [70, 80, 90]], np.uint8)
    mx = np.array([[0.0, 0.5, 1.25, 2.0, 2.5, -0.5, 3.0, np.nan]], np.float32)
    my = np.array([[0.0, 0.5, 1.75, 2.0, 0.0, 0.0, 1.0, 0.0]], np.float32)
    out = _oracle_remap(src, mx, my)[0].tolist()
    # exact pixel; centre of four; general weights; last pixel; half outside right; half outside left; outside; NaN -> outside
    general = (50 * 24 * 8 * 32 + 60 * 8 * 8 * 32 + 80 * 24 * 24 * 32 + 90 * 8 * 24 * 32 + 16384) >> 15
    assert out == [10, 30, general, 90, 15, 5, 0, 0]
    # coordinates are rounded to 1/32 pixel, halves to even: 1/64 -> 0, 3/64 -> 2/32
    mx = np.array([[1 / 64, 3 / 64]], np.float32)
    assert _oracle_remap(src, mx, np.zeros_like(mx))[0].tolist() == [10, (10 * 30 * 32 * 32 + 20 * 2 * 32 * 32 + 16384) >> 15]


# ---- GPU parity -----------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("w,h,cn,order", [(1241, 376, 3, BGR), (640, 480, 3, RGB), (333, 17, 4, BGR), (5, 3, 4, RGB), (1, 1, 3, BGR), (1027, 9, 3, RGB)])
def test_gpu_gray_matches_oracle(w, h, cn, order):
    from openvslam_amd import util
    rng = np.random.default_rng(w * 7 + h)
    img = rng.integers(0, 256, (h, w, cn), dtype=np.uint8)
    assert np.array_equal(util.convert_to_grayscale(img, order), _oracle_gray(img, order))


@pytest.mark.gpu
def test_gpu_gray_batch_with_odd_strides():
    """Device batch entry point with row / frame strides and base addresses that are not multiples of four."""
    import torch
    from openvslam_amd import _lib
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(3)
    n, w, h, cn = 5, 317, 41, 3
    src_rs, src_fs, dst_rs, dst_fs = w * cn + 7, (w * cn + 7) * h + 13, w + 3, (w + 3) * h + 5
    src = rng.integers(0, 256, n * src_fs + 64, dtype=np.uint8)
    d_src = torch.from_numpy(src).to(dev)
    for src_off, dst_off in ((0, 0), (1, 2), (3, 1)):
        d_dst = torch.full((n * dst_fs + 64,), 171, dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib().ovs_convert_to_grayscale_dev(d_src.data_ptr() + src_off, n, w, h, cn, src_rs, src_fs, BGR,
                                                           d_dst.data_ptr() + dst_off, dst_rs, dst_fs, 0, torch.cuda.current_stream(dev).cuda_stream))
        torch.cuda.synchronize(dev)
        got = d_dst.cpu().numpy()
        want = np.full_like(got, 171)
        for f in range(n):
            frame = np.lib.stride_tricks.as_strided(src[src_off + f * src_fs:], (h, w, cn), (src_rs, cn, 1))
            g = _oracle_gray(np.ascontiguousarray(frame), BGR)
            for y in range(h):
                o = dst_off + f * dst_fs + y * dst_rs
                want[o:o + w] = g[y]
        assert np.array_equal(got, want)  # pixels right, padding untouched


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,dtype", [(752, 480, np.uint8), (1241, 376, np.uint8), (63, 5, np.uint8), (752, 480, np.uint16), (101, 7, np.uint16)])
def test_gpu_equalize_matches_oracle(w, h, dtype):
    from openvslam_amd import util
    rng = np.random.default_rng(w + h)
    top = 200 if dtype == np.uint8 else 30000
    img = np.clip(rng.normal(top / 2, top / 6, (h, w)), 0, np.iinfo(dtype).max).astype(dtype)
    rc, want = _oracle_equalize(img)
    assert rc == 0
    got = util.equalize_histogram(img)
    assert np.array_equal(got, want)
    assert got.min() == 0 and got.max() == 255


@pytest.mark.gpu
def test_gpu_equalize_flat_image_is_reported():
    from openvslam_amd import util
    from openvslam_amd._lib import OvsError
    with pytest.raises(OvsError) as e:
        util.equalize_histogram(np.full((16, 16), 5, np.uint8))
    assert e.value.code == A.OVS_ERR_UNSUPPORTED


@pytest.mark.gpu
def test_gpu_equalize_batch_in_place_with_odd_strides():
    import torch
    from openvslam_amd import _lib
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(8)
    n, w, h = 4, 211, 33
    rs, fs, off = w + 1, (w + 1) * h + 3, 1
    buf = rng.integers(20, 230, n * fs + 16, dtype=np.uint8)
    buf[off + 2 * fs: off + 3 * fs] = 40  # frame 2 is flat
    d = torch.from_numpy(buf).to(dev)
    flags = torch.full((n,), 9, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().ovs_equalize_histogram_dev(d.data_ptr() + off, n, w, h, rs, fs, flags.data_ptr(), 0, torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize(dev)
    got = d.cpu().numpy()
    want = buf.copy()
    for f in (0, 1, 3):
        frame = np.lib.stride_tricks.as_strided(buf[off + f * fs:], (h, w), (rs, 1))
        rc, eq = _oracle_equalize(np.ascontiguousarray(frame))
        assert rc == 0
        for y in range(h):
            o = off + f * fs + y * rs
            want[o:o + w] = eq[y]
    assert flags.cpu().tolist() == [0, 0, 1, 0]
    assert np.array_equal(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,seed", [(752, 480, 1), (1241, 376, 2), (37, 23, 3)])
def test_gpu_remap_matches_oracle(w, h, seed):
    from openvslam_amd import util
    rng = np.random.default_rng(seed)
    src = rng.integers(0, 256, (h, w), dtype=np.uint8)
    mx, my = _rectify_maps(w, h, seed)
    assert (mx < 0).any() or (my < 0).any() or (mx > w - 1).any() or (my > h - 1).any()  # the border is exercised
    want = _oracle_remap(src, mx, my)
    r = util.remap(mx, my)
    assert np.array_equal(r.apply(src), want)
    # integer coordinates, the borders and invalid coordinates
    mx2 = np.tile(np.arange(-2, w + 2, dtype=np.float32), (h + 4, 1))
    my2 = np.tile(np.arange(-2, h + 2, dtype=np.float32)[:, None], (1, w + 4))
    mx2[0, :3] = [np.nan, np.inf, -np.inf]
    mx2[1, :2] = [1e12, -1e12]
    assert np.array_equal(util.remap(mx2, my2).apply(src), _oracle_remap(src, mx2, my2))
    # a different source size than the map size (the map defines the output size)
    small = src[: h // 2, : w // 2]
    assert np.array_equal(r.apply(small), _oracle_remap(small, mx, my))


@pytest.mark.gpu
def test_gpu_stereo_rectifier_and_batch_remap():
    import torch
    from openvslam_amd import util
    dev = torch.device("cuda", 0)
    w, h, n = 320, 200, 3
    rng = np.random.default_rng(5)
    maps = _rectify_maps(w, h, 11) + _rectify_maps(w, h, 12)
    rect = util.stereo_rectifier(*maps)
    left = rng.integers(0, 256, (h, w), dtype=np.uint8)
    right = rng.integers(0, 256, (h, w), dtype=np.uint8)
    out_l, out_r = rect.rectify(left, right)
    assert np.array_equal(out_l, _oracle_remap(left, maps[0], maps[1]))
    assert np.array_equal(out_r, _oracle_remap(right, maps[2], maps[3]))
    frames = rng.integers(0, 256, (n, h, w), dtype=np.uint8)
    d_src = torch.from_numpy(frames).to(dev)
    d_dst = torch.zeros((n, h, w + 1), dtype=torch.uint8, device=dev)  # odd row stride
    rect._l.apply_batch_dev(d_src.data_ptr(), n, w, h, w, w * h, d_dst.data_ptr(), w + 1, (w + 1) * h, torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    got = d_dst.cpu().numpy()
    for f in range(n):
        assert np.array_equal(got[f, :, :w], _oracle_remap(frames[f], maps[0], maps[1]))
    assert (got[:, :, w] == 0).all()

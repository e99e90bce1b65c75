"""GPU parity of match::stereo::compute and robust::match_for_triangulation against the oracle."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
from openvslam_amd import _abi as A
from openvslam_amd import _lib, synth
from openvslam_amd.feature import orb_extractor, orb_params

pytestmark = pytest.mark.gpu

# EuRoC stereo configuration (example/euroc/EuRoC_stereo.yaml): fx 435.2, fx * baseline 47.906
FX, FXB = 435.2046959714599, 47.90639384423901


def _views(levels):
    arr = (A.ImageView * len(levels))()
    keep = []
    for i, im in enumerate(levels):
        im = np.ascontiguousarray(im)
        keep.append(im)
        arr[i].data = A.ptr(im, A.u8p)
        arr[i].width, arr[i].height, arr[i].stride = im.shape[1], im.shape[0], im.shape[1]
    return arr, keep


def _same_float(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("w,h,n,seed", [(752, 480, 1000, 3), (640, 480, 1500, 4)])
def test_stereo_compute_matches_oracle(w, h, n, seed):
    left, right, disps, band_h = synth.make_stereo_pair(w, h, seed=seed)
    ref_l, ref_r = oracle.OrbExtractor(max_num_keypts=n), oracle.OrbExtractor(max_num_keypts=n)
    kl, dl = ref_l.extract(left)
    kr, dr = ref_r.extract(right)
    lv = [ref_l.pyramid_level(i) for i in range(8)]
    rv = [ref_r.pyramid_level(i) for i in range(8)]
    la, keep_l = _views(lv)
    ra, keep_r = _views(rv)
    sf = oracle.scale_table("scale_factors", 8, 1.2)
    isf = oracle.scale_table("inv_scale_factors", 8, 1.2)
    true_baseline = np.float32(FXB / FX)
    res = []
    for lib_, prefix, extra in ((_lib.lib(), "ovs_", (0,)), (oracle.lib(), "ovo_", ())):
        xr = np.full(len(kl), 7.0, dtype=np.float32)
        dep = np.full(len(kl), 7.0, dtype=np.float32)
        rc = getattr(lib_, prefix + "stereo_compute")(la, ra, 8, A.ptr(kl, A.kpp), len(kl), A.ptr(kr, A.kpp), len(kr),
                                                      A.ptr(dl, A.u8p), A.ptr(dr, A.u8p), A.ptr(sf, A.f32p), A.ptr(isf, A.f32p),
                                                      np.float32(FXB), true_baseline, A.ptr(xr, A.f32p), A.ptr(dep, A.f32p), *extra)
        assert rc == 0
        res.append((xr, dep))
    assert _same_float(res[0][0], res[1][0]), "stereo_x_right differs"
    assert _same_float(res[0][1], res[1][1]), "depths differ"
    xr, dep = res[0]
    ok = xr >= 0
    assert ok.sum() > len(kl) // 5
    # the recovered disparities follow the banded ground truth
    band = np.minimum((kl["y"][ok] // band_h).astype(int), len(disps) - 1)
    err = np.abs((kl["x"][ok] - xr[ok]) - disps[band])
    assert np.median(err) < 1.0


def test_stereo_row_lists_larger_than_lds():
    # 4000 right keypoints on the coarse levels span 11..16 image rows each: their row lists (about
    # 50 k entries) do not fit in the LDS the row kernel builds them in, and it builds them in global
    # memory instead
    w, h = 752, 480
    left, right, _, _ = synth.make_stereo_pair(w, h, seed=9)
    ref_l, ref_r = oracle.OrbExtractor(max_num_keypts=1000), oracle.OrbExtractor(max_num_keypts=1000)
    kl, dl = ref_l.extract(left)
    kr, dr = ref_r.extract(right)
    keep = (kr["y"] > 20) & (kr["y"] < h - 20)
    kr, dr = kr[keep], dr[keep]
    kr = np.ascontiguousarray(np.tile(kr, 4))
    dr = np.ascontiguousarray(np.tile(dr, (4, 1)))
    kr["octave"] = 5 + (np.arange(len(kr)) % 3)
    la, keep_l = _views([ref_l.pyramid_level(i) for i in range(8)])
    ra, keep_r = _views([ref_r.pyramid_level(i) for i in range(8)])
    sf = oracle.scale_table("scale_factors", 8, 1.2)
    isf = oracle.scale_table("inv_scale_factors", 8, 1.2)
    res = []
    for lib_, prefix, extra in ((_lib.lib(), "ovs_", (0,)), (oracle.lib(), "ovo_", ())):
        xr = np.full(len(kl), 7.0, dtype=np.float32)
        dep = np.full(len(kl), 7.0, dtype=np.float32)
        rc = getattr(lib_, prefix + "stereo_compute")(la, ra, 8, A.ptr(kl, A.kpp), len(kl), A.ptr(kr, A.kpp), len(kr),
                                                      A.ptr(dl, A.u8p), A.ptr(dr, A.u8p), A.ptr(sf, A.f32p), A.ptr(isf, A.f32p),
                                                      np.float32(FXB), np.float32(FXB / FX), A.ptr(xr, A.f32p), A.ptr(dep, A.f32p), *extra)
        res.append((rc, xr, dep))
    assert res[0][0] == res[1][0] == 0
    assert _same_float(res[0][1], res[1][1]) and _same_float(res[0][2], res[1][2])
    assert (res[0][1] >= 0).sum() > 10


def test_stereo_compute_dev_matches_oracle():
    w, h, n = 752, 480, 1000
    dev = torch.device("cuda", 0)
    left, right, _, _ = synth.make_stereo_pair(w, h, seed=9)
    exts = [orb_extractor(orb_params(max_num_keypts=n), device=0) for _ in range(2)]
    cap = exts[0].keypoint_capacity(w, h)
    stream = torch.cuda.current_stream(dev).cuda_stream
    bufs = []
    for ext, img in zip(exts, (left, right)):
        d_img = torch.from_numpy(img).to(dev)
        kps = torch.zeros((1, cap, 7), dtype=torch.float32, device=dev)
        desc = torch.zeros((1, cap, 32), dtype=torch.uint8, device=dev)
        cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
        ext.extract_batch_dev(d_img.data_ptr(), 1, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap, stream)
        bufs.append((d_img, kps, desc, cnt))
    xr = torch.zeros((cap,), dtype=torch.float32, device=dev)
    dep = torch.zeros((cap,), dtype=torch.float32, device=dev)
    true_baseline = np.float32(FXB / FX)
    status1 = torch.full((1,), 77, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().ovs_stereo_compute_dev(exts[0]._h, 0, exts[1]._h, 0,
                                                 bufs[0][1].data_ptr(), bufs[0][2].data_ptr(), bufs[0][3].data_ptr(),
                                                 bufs[1][1].data_ptr(), bufs[1][2].data_ptr(), bufs[1][3].data_ptr(), cap,
                                                 np.float32(FXB), true_baseline, xr.data_ptr(), dep.data_ptr(), status1.data_ptr(), 0, stream))
    torch.cuda.synchronize(dev)
    assert int(status1.item()) == 0
    nl = int(bufs[0][3].item())
    ref_l, ref_r = oracle.OrbExtractor(max_num_keypts=n), oracle.OrbExtractor(max_num_keypts=n)
    kl, dl = ref_l.extract(left)
    kr, dr = ref_r.extract(right)
    assert nl == len(kl)
    la, keep_l = _views([ref_l.pyramid_level(i) for i in range(8)])
    ra, keep_r = _views([ref_r.pyramid_level(i) for i in range(8)])
    sf = oracle.scale_table("scale_factors", 8, 1.2)
    isf = oracle.scale_table("inv_scale_factors", 8, 1.2)
    rxr = np.zeros(len(kl), dtype=np.float32)
    rdep = np.zeros(len(kl), dtype=np.float32)
    oracle.check(oracle.lib().ovo_stereo_compute(la, ra, 8, A.ptr(kl, A.kpp), len(kl), A.ptr(kr, A.kpp), len(kr), A.ptr(dl, A.u8p), A.ptr(dr, A.u8p),
                                                 A.ptr(sf, A.f32p), A.ptr(isf, A.f32p), np.float32(FXB), true_baseline,
                                                 A.ptr(rxr, A.f32p), A.ptr(rdep, A.f32p)))
    assert _same_float(xr.cpu().numpy()[:nl], rxr)
    assert _same_float(dep.cpu().numpy()[:nl], rdep)
    assert (rxr >= 0).sum() > 100


def test_stereo_batch_equals_pair_by_pair_and_oracle():
    """One set of launches for a batch of pairs gives, pair by pair, what the single-pair entry
    point gives (first pair also checked against the CPU oracle)."""
    w, h, n, pairs = 752, 480, 1000, 5
    dev = torch.device("cuda", 0)
    imgs = [synth.make_stereo_pair(w, h, seed=20 + p)[:2] for p in range(pairs)]
    exts = [orb_extractor(orb_params(max_num_keypts=n), device=0) for _ in range(2)]
    cap = exts[0].keypoint_capacity(w, h)
    stream = torch.cuda.current_stream(dev).cuda_stream
    bufs = []
    for side, ext in enumerate(exts):
        d_img = torch.from_numpy(np.stack([imgs[p][side] for p in range(pairs)])).to(dev)
        kps = torch.zeros((pairs, cap, 7), dtype=torch.float32, device=dev)
        desc = torch.zeros((pairs, cap, 32), dtype=torch.uint8, device=dev)
        cnt = torch.zeros((pairs,), dtype=torch.int32, device=dev)
        ext.extract_batch_dev(d_img.data_ptr(), pairs, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap, stream)
        bufs.append((d_img, kps, desc, cnt))
    true_baseline = np.float32(FXB / FX)
    xr = torch.full((pairs, cap), 5.0, dtype=torch.float32, device=dev)
    dep = torch.full((pairs, cap), 5.0, dtype=torch.float32, device=dev)
    status = torch.full((1,), 77, dtype=torch.int32, device=dev)
    L = _lib.lib()
    _lib.check(L.ovs_stereo_compute_batch_dev(exts[0]._h, exts[1]._h, pairs, bufs[0][1].data_ptr(), bufs[0][2].data_ptr(), bufs[0][3].data_ptr(),
                                              bufs[1][1].data_ptr(), bufs[1][2].data_ptr(), bufs[1][3].data_ptr(), cap,
                                              np.float32(FXB), true_baseline, xr.data_ptr(), dep.data_ptr(), status.data_ptr(), 0, stream))
    torch.cuda.synchronize(dev)
    assert int(status.item()) == 0
    counts = bufs[0][3].cpu().numpy()
    for p in range(pairs):
        xr1 = torch.zeros((cap,), dtype=torch.float32, device=dev)
        dep1 = torch.zeros((cap,), dtype=torch.float32, device=dev)
        _lib.check(L.ovs_stereo_compute_dev(exts[0]._h, p, exts[1]._h, p, bufs[0][1][p].data_ptr(), bufs[0][2][p].data_ptr(), bufs[0][3][p:].data_ptr(),
                                            bufs[1][1][p].data_ptr(), bufs[1][2][p].data_ptr(), bufs[1][3][p:].data_ptr(), cap,
                                            np.float32(FXB), true_baseline, xr1.data_ptr(), dep1.data_ptr(), None, 0, stream))
        torch.cuda.synchronize(dev)
        nl = int(counts[p])
        assert nl > 500
        assert _same_float(xr[p, :nl].cpu().numpy(), xr1[:nl].cpu().numpy()), p
        assert _same_float(dep[p, :nl].cpu().numpy(), dep1[:nl].cpu().numpy()), p
        assert (xr1[:nl] >= 0).sum().item() > 100
        # entries past the keypoint count are not written
        assert bool((xr[p, nl:] == 5.0).all())
    # pair 0 against the oracle
    ref_l, ref_r = oracle.OrbExtractor(max_num_keypts=n), oracle.OrbExtractor(max_num_keypts=n)
    kl, dl = ref_l.extract(imgs[0][0])
    kr, dr = ref_r.extract(imgs[0][1])
    la, keep_l = _views([ref_l.pyramid_level(i) for i in range(8)])
    ra, keep_r = _views([ref_r.pyramid_level(i) for i in range(8)])
    sf = oracle.scale_table("scale_factors", 8, 1.2)
    isf = oracle.scale_table("inv_scale_factors", 8, 1.2)
    rxr = np.zeros(len(kl), dtype=np.float32)
    rdep = np.zeros(len(kl), dtype=np.float32)
    oracle.check(oracle.lib().ovo_stereo_compute(la, ra, 8, A.ptr(kl, A.kpp), len(kl), A.ptr(kr, A.kpp), len(kr), A.ptr(dl, A.u8p), A.ptr(dr, A.u8p),
                                                 A.ptr(sf, A.f32p), A.ptr(isf, A.f32p), np.float32(FXB), true_baseline,
                                                 A.ptr(rxr, A.f32p), A.ptr(rdep, A.f32p)))
    assert len(kl) == int(counts[0])
    assert _same_float(xr[0, :len(kl)].cpu().numpy(), rxr)
    assert _same_float(dep[0, :len(kl)].cpu().numpy(), rdep)


def test_stereo_resident_equals_host_pyramid_path():
    """Live path: pyramids stay in the extractor handles after the host-buffer extract; only keypoints
    and descriptors are sent.  Same result as the entry point that takes the pyramids from the host."""
    w, h, n = 752, 480, 1000
    left, right, _, _ = synth.make_stereo_pair(w, h, seed=31)
    exts = [orb_extractor(orb_params(max_num_keypts=n), device=0) for _ in range(2)]
    kl, dl = exts[0].extract(left)
    kr, dr = exts[1].extract(right)
    true_baseline = np.float32(FXB / FX)
    xr = np.full(len(kl), 7.0, dtype=np.float32)
    dep = np.full(len(kl), 7.0, dtype=np.float32)
    _lib.check(_lib.lib().ovs_stereo_compute_resident(exts[0]._h, exts[1]._h, A.ptr(kl, A.kpp), len(kl), A.ptr(kr, A.kpp), len(kr),
                                                      A.ptr(dl, A.u8p), A.ptr(dr, A.u8p), np.float32(FXB), true_baseline,
                                                      A.ptr(xr, A.f32p), A.ptr(dep, A.f32p), 0))
    ref_l, ref_r = oracle.OrbExtractor(max_num_keypts=n), oracle.OrbExtractor(max_num_keypts=n)
    okl, odl = ref_l.extract(left)
    okr, odr = ref_r.extract(right)
    assert np.array_equal(okl.view(np.uint8), kl.view(np.uint8)) and np.array_equal(odl, dl)
    la, keep_l = _views([ref_l.pyramid_level(i) for i in range(8)])
    ra, keep_r = _views([ref_r.pyramid_level(i) for i in range(8)])
    sf = oracle.scale_table("scale_factors", 8, 1.2)
    isf = oracle.scale_table("inv_scale_factors", 8, 1.2)
    rxr = np.zeros(len(kl), dtype=np.float32)
    rdep = np.zeros(len(kl), dtype=np.float32)
    oracle.check(oracle.lib().ovo_stereo_compute(la, ra, 8, A.ptr(okl, A.kpp), len(okl), A.ptr(okr, A.kpp), len(okr), A.ptr(odl, A.u8p), A.ptr(odr, A.u8p),
                                                 A.ptr(sf, A.f32p), A.ptr(isf, A.f32p), np.float32(FXB), true_baseline,
                                                 A.ptr(rxr, A.f32p), A.ptr(rdep, A.f32p)))
    assert _same_float(xr, rxr) and _same_float(dep, rdep)
    assert (xr >= 0).sum() > 100
    # no keypoints at all is fine
    _lib.check(_lib.lib().ovs_stereo_compute_resident(exts[0]._h, exts[1]._h, None, 0, None, 0, None, None, np.float32(FXB), true_baseline, None, None, 0))


def test_stereo_out_of_range_row_is_reported():
    img = np.zeros((64, 64), dtype=np.uint8)
    la, k1 = _views([img])
    ra, k2 = _views([img])
    kp = np.zeros(1, dtype=A.KEYPOINT_DTYPE)
    kp["x"], kp["y"] = 30, 63.5  # row list spans rows 61..66 -> outside of the 64-row image
    d = np.zeros((1, 32), dtype=np.uint8)
    sf = np.ones(1, dtype=np.float32)
    out = np.zeros(1, dtype=np.float32)
    for lib_, prefix, extra in ((_lib.lib(), "ovs_", (0,)), (oracle.lib(), "ovo_", ())):
        rc = getattr(lib_, prefix + "stereo_compute")(la, ra, 1, A.ptr(kp, A.kpp), 1, A.ptr(kp, A.kpp), 1, A.ptr(d, A.u8p), A.ptr(d, A.u8p),
                                                      A.ptr(sf, A.f32p), A.ptr(sf, A.f32p), 40.0, 0.1, A.ptr(out, A.f32p), A.ptr(out, A.f32p), *extra)
        assert rc == A.OVS_ERR_OUT_OF_RANGE


def test_device_stereo_reports_out_of_range_keypoints():
    """The single-pair device entry point hands the kernels' error word to the caller: a left keypoint
    whose row band leaves the image gives OVS_ERR_OUT_OF_RANGE in *d_status (and 0 without it)."""
    dev = torch.device("cuda", 0)
    w, h, n = 320, 240, 300
    frames = synth.make_sequence(w, h, 2, seed=12, dx=4, dy=0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    exts, bufs = [], []
    for f in range(2):
        ext = orb_extractor(orb_params(max_num_keypts=n), device=0)
        cap = ext.keypoint_capacity(w, h)
        d_img = torch.from_numpy(frames[f]).to(dev)
        kps = torch.zeros((1, cap, 7), dtype=torch.float32, device=dev)
        desc = torch.zeros((1, cap, 32), dtype=torch.uint8, device=dev)
        cnt = torch.zeros((1,), dtype=torch.int32, device=dev)
        ext.extract_batch_dev(d_img.data_ptr(), 1, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), cnt.data_ptr(), cap, stream)
        exts.append(ext)
        bufs.append((d_img, kps, desc, cnt))
    xr = torch.zeros((cap,), dtype=torch.float32, device=dev)
    dep = torch.zeros((cap,), dtype=torch.float32, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)

    def run(d_status):
        return _lib.lib().ovs_stereo_compute_dev(exts[0]._h, 0, exts[1]._h, 0, bufs[0][1].data_ptr(), bufs[0][2].data_ptr(), bufs[0][3].data_ptr(),
                                                 bufs[1][1].data_ptr(), bufs[1][2].data_ptr(), bufs[1][3].data_ptr(), cap,
                                                 np.float32(40.0), np.float32(0.1), xr.data_ptr(), dep.data_ptr(), d_status, 0, stream)

    _lib.check(run(status.data_ptr()))
    torch.cuda.synchronize(dev)
    assert int(status.item()) == 0
    # move the first right keypoint to the last image row: its row band (y +- 2 * scale) leaves the image
    bufs[1][1][0, 0, 1] = float(h) - 0.5
    _lib.check(run(status.data_ptr()))
    torch.cuda.synchronize(dev)
    assert int(status.item()) == A.OVS_ERR_OUT_OF_RANGE
    _lib.check(run(None))
    torch.cuda.synchronize(dev)


def _unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


@pytest.mark.parametrize("seed,n,words,check", [(1, 1500, 120, 1), (2, 800, 20, 0)])
def test_match_for_triangulation(seed, n, words, check):
    rng = np.random.default_rng(seed)
    # two views of the same random points: camera 2 = R * camera 1 + t
    pts1 = np.column_stack([rng.uniform(-4, 4, n), rng.uniform(-3, 3, n), rng.uniform(4, 20, n)])
    ang = 0.05
    R21 = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    t21 = np.array([0.4, 0.02, 0.05])
    pts2 = pts1 @ R21.T + t21
    order2 = rng.permutation(n)
    b1 = _unit(pts1)
    b2 = _unit(pts2[order2] + rng.normal(0, 0.002, (n, 3)))
    # E_12 with x1^T E_12 x2 = 0: x1 = R12 x2 + t12
    R12, t12 = R21.T, -R21.T @ t21
    tx = np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]])
    E12 = np.ascontiguousarray(tx @ R12)
    epipole_in_2 = _unit(t21[None, :])[0].copy()  # camera centre 1 seen from camera 2
    word = rng.integers(0, words, n)
    centres = rng.integers(0, 256, (words, 32), dtype=np.uint8)
    base = centres[word].copy()
    for i in range(n):
        for b in rng.choice(256, 40, replace=False):
            base[i, b >> 3] ^= np.uint8(1 << (b & 7))

    def flip(d, k):
        d = d.copy()
        for i in range(len(d)):
            for b in rng.choice(256, rng.integers(0, k + 1), replace=False):
                d[i, b >> 3] ^= np.uint8(1 << (b & 7))
        return d

    d1 = flip(base, 10)
    d2 = flip(base[order2], 10)
    k1 = np.zeros(n, dtype=A.KEYPOINT_DTYPE)
    k2 = np.zeros(n, dtype=A.KEYPOINT_DTYPE)
    k1["octave"] = rng.integers(0, 8, n)
    k1["angle"] = rng.uniform(0, 360, n)
    k2["angle"] = (k1["angle"][order2] + rng.normal(0, 6, n)) % 360
    v1, v2 = {}, {}
    for i in rng.permutation(n):
        v1.setdefault(int(word[i]), []).append(int(i))
    for j in rng.permutation(n):
        v2.setdefault(int(word[order2[j]]), []).append(int(j))
    fv1, fv2 = A.make_bow_feat_vec(v1), A.make_bow_feat_vec(v2)
    has1 = (rng.random(n) < 0.3).astype(np.uint8)
    has2 = (rng.random(n) < 0.3).astype(np.uint8)
    sxr1 = np.where(rng.random(n) < 0.2, 10.0, -1.0).astype(np.float32)
    sf = oracle.scale_table("scale_factors", 8, 1.2)
    b1c, b2c = np.ascontiguousarray(b1), np.ascontiguousarray(b2)
    res = []
    for lib_, prefix, extra in ((_lib.lib(), "ovs_", (0,)), (oracle.lib(), "ovo_", ())):
        pairs = np.zeros((n, 2), dtype=np.uint32)
        num = C.c_uint32(0)
        rc = getattr(lib_, prefix + "match_for_triangulation")(fv1.ref(), A.ptr(d1, A.u8p), A.ptr(k1, A.kpp), A.ptr(b1c, A.f64p), A.ptr(sxr1, A.f32p), A.ptr(has1, A.u8p), n,
                                                               fv2.ref(), A.ptr(d2, A.u8p), A.ptr(k2, A.kpp), A.ptr(b2c, A.f64p), None, A.ptr(has2, A.u8p), n,
                                                               A.ptr(E12, A.f64p), A.ptr(epipole_in_2, A.f64p), A.ptr(sf, A.f32p), 8, check,
                                                               A.ptr(pairs, A.u32p), C.byref(num), *extra)
        assert rc == 0
        res.append((pairs[:num.value].copy(), num.value))
    assert res[0][1] == res[1][1]
    assert np.array_equal(res[0][0], res[1][0])
    assert res[0][1] > n // 20


def test_match_for_triangulation_borderline_epipolar_decisions():
    """Pairs whose epipolar residual sits within a few hundred ulps of the threshold: the reference
    decides them through acos of its libm (robust.cc:324-334), the device compares cosines and hands
    such pairs to the host.  With E = identity, bearing_2 = (0, 0, 1) and bearing_1 = (., 0, c) the
    cosine of the residual is exactly c, so c = sin(threshold) + k ulp plants the case."""
    import math
    thr = 0.2 * math.pi / 180.0
    sf = oracle.scale_table("scale_factors", 8, 1.2)
    ks = list(range(-330, 331, 3))
    levels = [0, 1, 5, 7]
    n = len(ks) * len(levels)
    b1 = np.zeros((n, 3))
    b2 = np.zeros((n, 3))
    k1 = np.zeros(n, dtype=A.KEYPOINT_DTYPE)
    k2 = np.zeros(n, dtype=A.KEYPOINT_DTYPE)
    flips = 0
    i = 0
    for lv in levels:
        T = thr * float(sf[lv])
        s = math.sin(T)
        for k in ks:
            c = float(np.float64(s) + k * np.spacing(np.float64(s)))
            b1[i] = (math.sqrt(1.0 - c * c), 0.0, c)
            b2[i] = (0.0, 0.0, 1.0)
            k1["octave"][i] = lv
            flips += int((c < s) != ((math.pi / 2.0 - abs(math.acos(c))) < T))
            i += 1
    assert flips > 10, "the planted cosines no longer straddle the rounding of acos"
    rng = np.random.default_rng(5)
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    fv = A.make_bow_feat_vec({i: [i] for i in range(n)})
    E12 = np.ascontiguousarray(np.eye(3))
    epipole = np.array([1.0, 0.0, 0.0])
    res = []
    for lib_, prefix, extra in ((_lib.lib(), "ovs_", (0,)), (oracle.lib(), "ovo_", ())):
        pairs = np.zeros((n, 2), dtype=np.uint32)
        num = C.c_uint32(0)
        rc = getattr(lib_, prefix + "match_for_triangulation")(fv.ref(), A.ptr(d, A.u8p), A.ptr(k1, A.kpp), A.ptr(b1, A.f64p), None, None, n,
                                                               fv.ref(), A.ptr(d, A.u8p), A.ptr(k2, A.kpp), A.ptr(b2, A.f64p), None, None, n,
                                                               A.ptr(E12, A.f64p), A.ptr(epipole, A.f64p), A.ptr(sf, A.f32p), 8, 0,
                                                               A.ptr(pairs, A.u32p), C.byref(num), *extra)
        assert rc == 0
        res.append((pairs[:num.value].copy(), num.value))
    assert res[0][1] == res[1][1] and np.array_equal(res[0][0], res[1][0])
    assert 0 < res[0][1] < n


def test_match_for_triangulation_last_of_equals_and_exclusion():
    """robust.cc:102 keeps the LAST of equally distant candidates of a bucket, and a keypoint of
    keyframe 2 serves one keypoint of keyframe 1 only (robust.cc:88,131): large buckets of identical
    descriptors make both rules decide every match."""
    rng = np.random.default_rng(9)
    n, words = 600, 6
    word = rng.integers(0, words, n)
    centres = rng.integers(0, 256, (words, 32), dtype=np.uint8)
    d1 = centres[word].copy()
    d2 = centres[word].copy()
    # a few bits of difference in part of the rows, so that distance ties are common but not universal
    for i in rng.choice(n, n // 3, replace=False):
        d2[i, rng.integers(0, 32)] ^= np.uint8(1 << rng.integers(0, 8))
    pts = np.column_stack([rng.uniform(-4, 4, n), rng.uniform(-3, 3, n), rng.uniform(4, 20, n)])
    t21 = np.array([0.4, 0.0, 0.0])
    b1, b2 = _unit(pts), _unit(pts + t21)
    t12 = -t21
    E12 = np.ascontiguousarray(np.array([[0, -t12[2], t12[1]], [t12[2], 0, -t12[0]], [-t12[1], t12[0], 0]]))
    epipole = _unit(t21[None, :])[0].copy()
    k1 = np.zeros(n, dtype=A.KEYPOINT_DTYPE)
    k2 = np.zeros(n, dtype=A.KEYPOINT_DTYPE)
    k1["octave"] = 7  # the widest residual threshold: many candidates of a bucket are inliers
    v = {}
    for i in range(n):
        v.setdefault(int(word[i]), []).append(i)
    fv = A.make_bow_feat_vec(v)
    sf = oracle.scale_table("scale_factors", 8, 1.2)
    b1c, b2c = np.ascontiguousarray(b1), np.ascontiguousarray(b2)
    res = []
    for lib_, prefix, extra in ((_lib.lib(), "ovs_", (0,)), (oracle.lib(), "ovo_", ())):
        pairs = np.zeros((n, 2), dtype=np.uint32)
        num = C.c_uint32(0)
        rc = getattr(lib_, prefix + "match_for_triangulation")(fv.ref(), A.ptr(d1, A.u8p), A.ptr(k1, A.kpp), A.ptr(b1c, A.f64p), None, None, n,
                                                               fv.ref(), A.ptr(d2, A.u8p), A.ptr(k2, A.kpp), A.ptr(b2c, A.f64p), None, None, n,
                                                               A.ptr(E12, A.f64p), A.ptr(epipole, A.f64p), A.ptr(sf, A.f32p), 8, 0,
                                                               A.ptr(pairs, A.u32p), C.byref(num), *extra)
        assert rc == 0
        res.append((pairs[:num.value].copy(), num.value))
    assert res[0][1] == res[1][1] and np.array_equal(res[0][0], res[1][0])
    assert res[0][1] > n // 4

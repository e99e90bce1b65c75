"""First-principles checks of the CPU oracle's matchers on inputs whose answer is known by
construction, plus independent numpy restatements on small cases.  The reference's tests hold
vectors only for the descriptor distance and the angle checker (test/openvslam/match/); the
matcher outputs themselves are "parity unpinned" and rest on these checks."""
import ctypes as C

import numpy as np
import pytest

import oracle
from openvslam_amd import _abi as A

SF = np.array([1.2 ** i for i in range(8)], dtype=np.float32)
GRID = A.make_grid(0.0, 0.0, 640.0, 480.0)
POPCNT = np.array([bin(i).count("1") for i in range(256)], dtype=np.uint16)


def hamming_matrix(a, b):
    return POPCNT[a[:, None, :] ^ b[None, :, :]].sum(axis=2)


def spread_frame(rng, n, w=640.0, h=480.0):
    """Keypoints far apart from each other (> 45 px) so that every window holds at most one."""
    gx, gy = np.meshgrid(np.arange(30, w - 30, 48), np.arange(30, h - 30, 48))
    pts = np.stack([gx.ravel(), gy.ravel()], axis=1)
    pts = pts[rng.permutation(len(pts))[:n]]
    kps = np.zeros(len(pts), dtype=A.KEYPOINT_DTYPE)
    kps["x"], kps["y"] = pts[:, 0], pts[:, 1]
    kps["octave"] = rng.integers(0, 8, len(pts))
    kps["angle"] = rng.uniform(0, 360, len(pts))
    desc = rng.integers(0, 256, (len(pts), 32), dtype=np.uint8)
    return kps, desc


def call(name, *args, n_out):
    m = np.full(n_out, -7, dtype=np.int32)
    num = C.c_uint32(0)
    oracle.check(getattr(oracle.lib(), "ovo_" + name)(*args, A.ptr(m, A.i32p), C.byref(num)))
    return m, num.value


def test_current_and_last_frames_identity():
    """Every landmark re-projects onto its own keypoint with its own descriptor: all of them match,
    each to itself, and the rotation histogram keeps them (all deltas are 0)."""
    rng = np.random.default_rng(1)
    kps, desc = spread_frame(rng, 100)
    n = len(kps)
    view = A.make_frame_view(kps, desc, SF, GRID)
    xy = np.stack([kps["x"], kps["y"]], axis=1)
    q = A.make_proj_queries(xy, kps["octave"], desc, angle=kps["angle"])
    m, num = call("match_current_and_last_frames", view.ref(), q.ref(), None, np.float32(7.0), A.DIR_NONE, 8, np.float32(0.9), 1, n_out=n)
    assert num == n and np.array_equal(m, np.arange(n))


def test_current_and_last_frames_gates():
    rng = np.random.default_rng(2)
    kps, desc = spread_frame(rng, 60)
    n = len(kps)
    view = A.make_frame_view(kps, desc, SF, GRID)
    xy = np.stack([kps["x"], kps["y"]], axis=1).astype(np.float32)
    # invalid queries and occupied keypoints are skipped (projection.cc:103-104, 146-148)
    valid = np.ones(n, np.uint8)
    valid[::3] = 0
    occupied = np.zeros(n, np.uint8)
    occupied[1::3] = 1
    q = A.make_proj_queries(xy, kps["octave"], desc, valid=valid, angle=kps["angle"])
    m, num = call("match_current_and_last_frames", view.ref(), q.ref(), A.ptr(occupied, A.u8p), np.float32(7.0), A.DIR_NONE, 8, np.float32(0.9), 0, n_out=n)
    want = np.where((valid == 1) & (occupied == 0), np.arange(n), -1)
    assert np.array_equal(m, want) and num == int((want >= 0).sum())
    # a descriptor further than HAMMING_DIST_THR_HIGH (100) does not match (projection.cc:178)
    far = desc.copy()
    far[:, :16] ^= 0xFF  # 128 differing bits
    q = A.make_proj_queries(xy, kps["octave"], far, angle=kps["angle"])
    m, num = call("match_current_and_last_frames", view.ref(), q.ref(), None, np.float32(7.0), A.DIR_NONE, 8, np.float32(0.9), 0, n_out=n)
    assert num == 0 and np.all(m == -1)
    # a re-projection outside the image bounds is dropped (projection.cc:117-124)
    out = xy.copy()
    out[:, 0] += 10000
    q = A.make_proj_queries(out, kps["octave"], desc, angle=kps["angle"])
    m, num = call("match_current_and_last_frames", view.ref(), q.ref(), None, np.float32(7.0), A.DIR_NONE, 8, np.float32(0.9), 0, n_out=n)
    assert num == 0


def test_current_and_last_frames_rotation_check_clears_outliers():
    rng = np.random.default_rng(3)
    kps, desc = spread_frame(rng, 90)
    n = len(kps)
    view = A.make_frame_view(kps, desc, SF, GRID)
    xy = np.stack([kps["x"], kps["y"]], axis=1).astype(np.float32)
    ang = kps["angle"].copy()
    outl = np.arange(n) % 10 == 0
    ang[outl] = (ang[outl] + 180.0) % 360.0      # delta = 180 deg: a bin of its own
    q = A.make_proj_queries(xy, kps["octave"], desc, angle=ang)
    m, num = call("match_current_and_last_frames", view.ref(), q.ref(), None, np.float32(7.0), A.DIR_NONE, 8, np.float32(0.9), 1, n_out=n)
    # bins: 0 (inliers), 6 (outliers) -> top three keeps bin 6 only if it holds >= 10% of bin 0
    n_in, n_out_ = int((~outl).sum()), int(outl.sum())
    keep_outliers = n_out_ >= 0.1 * n_in
    want = np.where(outl & (not keep_outliers), -2, np.arange(n))
    assert np.array_equal(m, want) and num == int((want >= 0).sum())


def test_frame_and_landmarks_ratio_test():
    """Two keypoints on the same level in one window: accepted only when best <= ratio * second
    (projection.cc:64-69)."""
    kps = np.zeros(2, dtype=A.KEYPOINT_DTYPE)
    kps["x"], kps["y"], kps["octave"] = [100, 103], [100, 100], [2, 2]
    d0 = np.zeros(32, np.uint8)
    d1 = np.zeros(32, np.uint8)
    d1[0] = 0xFF          # 8 bits from d0
    desc = np.stack([d0, d1])
    view = A.make_frame_view(kps, desc, SF, GRID)
    qd = np.zeros((1, 32), np.uint8)
    qd[0, 1] = 0x3F       # 6 bits from d0, 14 from d1
    q = A.make_proj_queries(np.array([[101.0, 100.0]], np.float32), np.array([2], np.int32), qd)
    m, num = call("match_frame_and_landmarks", view.ref(), q.ref(), None, np.float32(5.0), np.float32(0.5), n_out=2)
    assert num == 1 and m.tolist() == [0, -1]          # 6 <= 0.5 * 14
    m, num = call("match_frame_and_landmarks", view.ref(), q.ref(), None, np.float32(5.0), np.float32(0.4), n_out=2)
    assert num == 0 and m.tolist() == [-1, -1]         # 6 > 0.4 * 14
    # on different levels the ratio test does not apply
    kps["octave"] = [2, 3]
    view = A.make_frame_view(kps, desc, SF, GRID)
    m, num = call("match_frame_and_landmarks", view.ref(), q.ref(), None, np.float32(5.0), np.float32(0.4), n_out=2)
    assert num == 1 and m.tolist() == [0, -1]


@pytest.mark.parametrize("seed,n1,n2", [(1, 60, 40), (2, 5, 1), (3, 64, 64), (4, 30, 90)])
def test_brute_force_match_vs_numpy(seed, n1, n2):
    """robust::brute_force_match (robust.cc:209-315) without the rotation check: for every keyframe
    keypoint (index 2) a strict-< scan over the frame keypoints (index 1) that are not matched yet,
    kept if best <= HAMMING_DIST_THR_LOW and best <= ratio * second; pairs come out as
    (idx_1, idx_2) ascending in idx_1."""
    rng = np.random.default_rng(seed)
    d1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    d2 = d1[rng.integers(0, n1, n2)].copy()
    for i in range(n2):
        for b in rng.choice(256, rng.integers(0, 40), replace=False):
            d2[i, b >> 3] ^= np.uint8(1 << (b & 7))
    D = hamming_matrix(d2, d1).astype(int)
    idx2_of_1 = {}
    for i2 in range(n2):
        best, second, b1 = 256, 256, -1
        for i1 in range(n1):
            if i1 in idx2_of_1:
                continue
            if D[i2, i1] < best:
                second, best, b1 = best, D[i2, i1], i1
            elif D[i2, i1] < second:
                second = D[i2, i1]
        if best > 50 or b1 < 0 or np.float32(0.8) * np.float32(second) < np.float32(best):
            continue
        idx2_of_1[b1] = i2
    want = sorted(idx2_of_1.items())
    a1 = np.zeros(n1, np.float32)
    a2 = np.zeros(n2, np.float32)
    m = np.zeros((max(n1, 1), 2), np.int32)
    num = C.c_uint32(0)
    oracle.check(oracle.lib().ovo_brute_force_match(A.ptr(d1, A.u8p), A.ptr(a1, A.f32p), n1, A.ptr(d2, A.u8p), A.ptr(a2, A.f32p),
                                                    None, n2, np.float32(0.8), 0, A.ptr(m, A.i32p), C.byref(num)))
    assert [tuple(r) for r in m[:num.value].tolist()] == want
    assert len(want) > 0


def test_knn2_vs_numpy():
    rng = np.random.default_rng(9)
    db = rng.integers(0, 256, (700, 32), dtype=np.uint8)
    db[650] = db[3]
    q = db[rng.integers(0, 700, 50)].copy()
    q[:, 5] ^= 0x0F
    q[0] = db[3]
    D = hamming_matrix(q, db).astype(np.int64)
    key = D * (1 << 20) + np.arange(700)[None, :]
    order = np.argsort(key, axis=1)[:, :2]
    out = np.zeros(len(q), dtype=A.KNN2_DTYPE)
    oracle.check(oracle.lib().ovo_knn2_search(A.ptr(db, A.u8p), 700, 1000, A.ptr(q, A.u8p), len(q), A.ptr(out, A.knn2p)))
    assert np.array_equal(out["row"], order + 1000)
    assert np.array_equal(out["dist"], np.take_along_axis(D, order, axis=1))
    assert out["row"][0].tolist() == [1003, 1650]


def test_stereo_recovers_a_constant_disparity():
    """Right image = left image shifted by 12 px: every left keypoint that finds its partner gets
    x_right = x - 12 within the sub-pixel fit and depth = fx*b / 12 (stereo.cc:20-111)."""
    from openvslam_amd import synth
    left = synth.make_frame(640, 480, seed=4)
    shift = 12
    right = np.roll(left, -shift, axis=1)
    ext_l, ext_r = oracle.OrbExtractor(max_num_keypts=800), oracle.OrbExtractor(max_num_keypts=800)
    kl, dl = ext_l.extract(left)
    pl = [left] + [ext_l.pyramid_level(i) for i in range(1, 8)]
    kr, dr = ext_r.extract(right)
    pr = [right] + [ext_r.pyramid_level(i) for i in range(1, 8)]

    def views(levels):
        arr = (A.ImageView * len(levels))()
        for i, im in enumerate(levels):
            arr[i].data = A.ptr(im, A.u8p)
            arr[i].width, arr[i].height, arr[i].stride = im.shape[1], im.shape[0], im.shape[1]
        return arr

    isf = (np.float32(1.0) / SF).astype(np.float32)
    xr = np.zeros(len(kl), np.float32)
    dep = np.zeros(len(kl), np.float32)
    fxb, base = np.float32(400.0), np.float32(0.5)
    oracle.check(oracle.lib().ovo_stereo_compute(views(pl), views(pr), 8, A.ptr(kl, A.kpp), len(kl), A.ptr(kr, A.kpp), len(kr),
                                                 A.ptr(dl, A.u8p), A.ptr(dr, A.u8p), A.ptr(SF, A.f32p), A.ptr(isf, A.f32p),
                                                 fxb, base, A.ptr(xr, A.f32p), A.ptr(dep, A.f32p)))
    ok = dep > 0
    assert ok.sum() > 0.3 * len(kl)
    disp = kl["x"][ok] - xr[ok]
    tol = 1.5 * SF[kl["octave"][ok]]
    assert np.mean(np.abs(disp - shift) <= tol) > 0.95
    assert np.allclose(dep[ok], fxb / disp, rtol=1e-5)
    assert np.all(xr[~ok] == -1) and np.all(dep[~ok] == -1)

"""GPU parity of the projection-window matchers against the CPU oracle (match indices bit-exact)."""
import ctypes as C

import numpy as np
import pytest

import oracle
from openvslam_amd import _abi as A
from openvslam_amd import _lib

pytestmark = pytest.mark.gpu


def random_frame(rng, n, w=1241.0, h=376.0, levels=8, stereo=False, cluster=False):
    kps = np.zeros(n, dtype=A.KEYPOINT_DTYPE)
    if cluster:
        cx, cy = rng.uniform(100, w - 100, 12), rng.uniform(50, h - 50, 12)
        k = rng.integers(0, 12, n)
        kps["x"] = np.clip(cx[k] + rng.normal(0, 15, n), 0, w - 1)
        kps["y"] = np.clip(cy[k] + rng.normal(0, 15, n), 0, h - 1)
    else:
        kps["x"] = rng.uniform(0, w, n)
        kps["y"] = rng.uniform(0, h, n)
    kps["octave"] = rng.integers(0, levels, n)
    kps["angle"] = rng.uniform(0, 360, n)
    desc = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    sxr = None
    if stereo:
        sxr = np.where(rng.random(n) < 0.6, kps["x"] - rng.uniform(1, 40, n), -1.0).astype(np.float32)
    return kps, desc, sxr


def make_queries(rng, kps, desc, sxr, nq, levels=8, noise_bits=20, jitter=6.0, frac_random=0.3):
    """Queries derived from frame keypoints (flipped descriptor bits, jittered positions) + random ones."""
    n = len(kps)
    src = rng.integers(0, n, nq)
    qd = desc[src].copy()
    flips = rng.integers(0, noise_bits + 1, nq)
    for i in range(nq):
        bits = rng.choice(256, flips[i], replace=False)
        for b in bits:
            qd[i, b >> 3] ^= np.uint8(1 << (b & 7))
    rnd = rng.random(nq) < frac_random
    qd[rnd] = rng.integers(0, 256, (int(rnd.sum()), 32), dtype=np.uint8)
    xy = np.stack([kps["x"][src] + rng.normal(0, jitter, nq), kps["y"][src] + rng.normal(0, jitter, nq)], axis=1).astype(np.float32)
    sl = np.clip(kps["octave"][src] + rng.integers(-1, 2, nq), 0, levels - 1).astype(np.int32)
    ang = ((kps["angle"][src] + rng.normal(0, 8, nq) + np.where(rng.random(nq) < 0.2, rng.uniform(0, 360, nq), 0)) % 360).astype(np.float32)
    valid = (rng.random(nq) < 0.9).astype(np.uint8)
    xr = None
    if sxr is not None:
        xr = (sxr[src] + rng.normal(0, 3, nq)).astype(np.float32)
    has_obs = (rng.random(nq) < 0.7).astype(np.uint8)
    return xy, sl, qd, valid, xr, ang, has_obs


SF = np.array([1.2 ** i for i in range(8)], dtype=np.float32)
GRID = A.make_grid(0.0, 0.0, 1241.0, 376.0)


def run_both(name, args_oracle, n_out, extra_outputs=True):
    G, O = _lib.lib(), oracle.lib()
    outs = []
    for lib_, prefix, extra in ((G, "ovs_", (0,)), (O, "ovo_", ())):
        m = np.full(n_out, -7, dtype=np.int32)
        num = C.c_uint32(12345)
        rc = getattr(lib_, prefix + name)(*args_oracle, A.ptr(m, A.i32p), C.byref(num), *extra)
        assert rc == 0, (prefix, rc)
        outs.append((m, num.value))
    assert outs[0][1] == outs[1][1], "match counts differ: %d vs %d" % (outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][0], outs[1][0]), "matched indices differ at %s" % np.nonzero(outs[0][0] != outs[1][0])[0][:10]
    return outs[0]


@pytest.mark.parametrize("seed,n,nq,direction,stereo,cluster,occ,check",
                         [(1, 2000, 2000, A.DIR_NONE, False, False, False, 1),
                          (2, 2000, 2500, A.DIR_FORWARD, True, False, True, 1),
                          (3, 1500, 2000, A.DIR_BACKWARD, True, True, True, 0),
                          (4, 3000, 3000, A.DIR_NONE, False, True, False, 1),
                          (5, 50, 400, A.DIR_NONE, False, True, False, 1),
                          # 146 KB of per-keypoint and per-query words: the most the workgroup-wide resolve kernel takes
                          (8, 5500, 2500, A.DIR_NONE, False, True, False, 1),
                          (6, 0, 10, A.DIR_NONE, False, False, False, 1),
                          (7, 300, 0, A.DIR_NONE, False, False, False, 1)])
def test_match_current_and_last_frames(seed, n, nq, direction, stereo, cluster, occ, check):
    rng = np.random.default_rng(seed)
    kps, desc, sxr = random_frame(rng, n, stereo=stereo, cluster=cluster)
    if n:
        xy, sl, qd, valid, xr, ang, has_obs = make_queries(rng, kps, desc, sxr, nq)
    else:
        xy = rng.uniform(0, 300, (nq, 2)).astype(np.float32)
        sl = rng.integers(0, 8, nq).astype(np.int32)
        qd = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        valid, xr, ang, has_obs = np.ones(nq, np.uint8), None, np.zeros(nq, np.float32), None
    view = A.make_frame_view(kps, desc, SF, GRID, sxr)
    q = A.make_proj_queries(xy, sl, qd, valid, xr, ang, has_obs)
    occupied = (rng.random(n) < 0.2).astype(np.uint8) if occ else None
    m, num = run_both("match_current_and_last_frames",
                      (view.ref(), q.ref(), A.ptr(occupied, A.u8p), 20.0, direction, 8, 0.9, check), n)
    if n >= 1500 and nq >= 2000:
        assert num > 200


@pytest.mark.parametrize("seed,n,nq,stereo,cluster", [(11, 2000, 4000, False, False), (12, 2500, 6000, True, True),
                                                      (13, 100, 1000, False, True),
                                                      # 136 KB of words: close to the limit of the workgroup-wide kernel for ratio tests
                                                      (14, 9000, 2000, False, True)])
def test_match_frame_and_landmarks(seed, n, nq, stereo, cluster):
    rng = np.random.default_rng(seed)
    kps, desc, sxr = random_frame(rng, n, stereo=stereo, cluster=cluster)
    xy, sl, qd, valid, xr, ang, has_obs = make_queries(rng, kps, desc, sxr, nq, jitter=2.0)
    view = A.make_frame_view(kps, desc, SF, GRID, sxr)
    q = A.make_proj_queries(xy, sl, qd, valid, xr, ang, has_obs)
    occupied = (rng.random(n) < 0.1).astype(np.uint8)
    for lowe in (0.8, 0.6):
        run_both("match_frame_and_landmarks", (view.ref(), q.ref(), A.ptr(occupied, A.u8p), 5.0, lowe), n)


@pytest.mark.parametrize("seed,n,nq,thr,check", [(21, 2000, 2000, 100, 1), (22, 2000, 1500, 50, 0), (23, 2000, 3000, 64, 1)])
def test_match_frame_and_keyframe(seed, n, nq, thr, check):
    rng = np.random.default_rng(seed)
    kps, desc, _ = random_frame(rng, n, cluster=True)
    xy, sl, qd, valid, xr, ang, _ = make_queries(rng, kps, desc, None, nq)
    view = A.make_frame_view(kps, desc, SF, GRID)
    q = A.make_proj_queries(xy, sl, qd, valid, None, ang)
    occupied = (rng.random(n) < 0.3).astype(np.uint8)
    run_both("match_frame_and_keyframe", (view.ref(), q.ref(), A.ptr(occupied, A.u8p), 10.0, thr, check), n)


@pytest.mark.parametrize("seed,n,nq", [(31, 2000, 2000), (32, 1200, 3000)])
def test_match_by_sim3_transform(seed, n, nq):
    rng = np.random.default_rng(seed)
    kps, desc, _ = random_frame(rng, n, cluster=True)
    xy, sl, qd, valid, xr, ang, _ = make_queries(rng, kps, desc, None, nq, noise_bits=10)
    view = A.make_frame_view(kps, desc, SF, GRID)
    q = A.make_proj_queries(xy, sl, qd, valid)
    occupied = (rng.random(n) < 0.3).astype(np.uint8)
    m, num = run_both("match_by_sim3_transform", (view.ref(), q.ref(), A.ptr(occupied, A.u8p), 7.5), n)
    # predicted level 0 rejects every candidate (unsigned `pred - 1`, projection.cc:390)
    assigned_q = m[m >= 0]
    assert np.all(sl[assigned_q] > 0)


def test_match_keyframes_mutually():
    rng = np.random.default_rng(41)
    n1, n2 = 1800, 2100
    k1, d1, _ = random_frame(rng, n1, cluster=True)
    k2, d2, _ = random_frame(rng, n2, cluster=True)
    G, O = _lib.lib(), oracle.lib()
    res = []
    for lib_, prefix, extra in ((G, "ovs_", (0,)), (O, "ovo_", ())):
        # keyframe 1 landmarks searched in keyframe 2 and vice versa
        xy, sl, qd, valid, _, _, _ = make_queries(np.random.default_rng(42), k2, d2, None, n1, noise_bits=10)
        v2 = A.make_frame_view(k2, d2, SF, GRID)
        q12 = A.make_proj_queries(xy, sl, qd, valid)
        b12 = np.full(n1, -9, dtype=np.int32)
        assert getattr(lib_, prefix + "match_keyframe_one_way")(v2.ref(), q12.ref(), 7.5, A.ptr(b12, A.i32p), *extra) == 0
        xy, sl, qd, valid, _, _, _ = make_queries(np.random.default_rng(43), k1, d1, None, n2, noise_bits=10)
        v1 = A.make_frame_view(k1, d1, SF, GRID)
        q21 = A.make_proj_queries(xy, sl, qd, valid)
        b21 = np.full(n2, -9, dtype=np.int32)
        assert getattr(lib_, prefix + "match_keyframe_one_way")(v1.ref(), q21.ref(), 7.5, A.ptr(b21, A.i32p), *extra) == 0
        out = np.full(n1, -9, dtype=np.int32)
        num = C.c_uint32(0)
        assert getattr(lib_, prefix + "cross_check")(A.ptr(b12, A.i32p), n1, A.ptr(b21, A.i32p), n2, A.ptr(out, A.i32p), C.byref(num)) == 0
        res.append((b12, b21, out, num.value))
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)


def test_greedy_conflicts_exhaust_candidate_lists():
    """Many queries compete for a handful of keypoints, so precomputed short lists run dry and the
    resolver has to rescan whole windows."""
    rng = np.random.default_rng(51)
    n = 40
    kps = np.zeros(n, dtype=A.KEYPOINT_DTYPE)
    kps["x"] = 600 + rng.uniform(-8, 8, n)
    kps["y"] = 200 + rng.uniform(-8, 8, n)
    kps["octave"] = 1
    kps["angle"] = rng.uniform(0, 360, n)
    base = rng.integers(0, 256, 32, dtype=np.uint8)
    desc = np.tile(base, (n, 1))
    for i in range(n):
        for b in rng.choice(256, 6, replace=False):
            desc[i, b >> 3] ^= np.uint8(1 << (b & 7))
    nq = 64
    xy = np.tile(np.array([[600.0, 200.0]], dtype=np.float32), (nq, 1))
    sl = np.ones(nq, dtype=np.int32)
    qd = np.tile(base, (nq, 1))
    view = A.make_frame_view(kps, desc, SF, GRID)
    q = A.make_proj_queries(xy, sl, qd, angle=np.zeros(nq, np.float32))
    m, num = run_both("match_current_and_last_frames", (view.ref(), q.ref(), None, 20.0, A.DIR_NONE, 8, 0.9, 0), n)
    assert num == n  # every keypoint ends up claimed, each by a different query
    run_both("match_frame_and_keyframe", (view.ref(), q.ref(), None, 20.0, 100, 0), n)
    run_both("match_frame_and_landmarks", (view.ref(), q.ref(), None, 20.0, 0.8), n)


def test_chain_of_forty_queries_that_push_each_other_on():
    """Forty keypoints on one spot, equally distant from 64 identical queries, octaves alternating so that
    the same-octave ratio test of match_frame_and_landmarks never applies (projection.cc:69-73): query k takes
    keypoint k, and from query 8 on every listed candidate is held, from query 16 on the next eight too, ...
    A resolver that keeps a bounded number of candidates per query has to fetch more, five times over."""
    n, nq = 40, 64
    rng = np.random.default_rng(52)
    kps = np.zeros(n, dtype=A.KEYPOINT_DTYPE)
    kps["x"], kps["y"] = 600.5, 200.5
    kps["octave"] = np.arange(n) % 2
    kps["angle"] = rng.uniform(0, 360, n)
    base = rng.integers(0, 256, 32, dtype=np.uint8)
    desc = np.tile(base, (n, 1))
    for i in range(n):
        for b in rng.choice(256, 6, replace=False):
            desc[i, b >> 3] ^= np.uint8(1 << (b & 7))
    xy = np.tile(np.array([[600.0, 200.0]], dtype=np.float32), (nq, 1))
    sl = np.ones(nq, dtype=np.int32)
    qd = np.tile(base, (nq, 1))
    view = A.make_frame_view(kps, desc, SF, GRID)
    q = A.make_proj_queries(xy, sl, qd, angle=np.zeros(nq, np.float32))
    m, num = run_both("match_frame_and_landmarks", (view.ref(), q.ref(), None, 20.0, 0.8), n)
    assert num == n and m.tolist() == list(range(n))
    m, num = run_both("match_current_and_last_frames", (view.ref(), q.ref(), None, 20.0, A.DIR_NONE, 8, 0.9, 0), n)
    assert num == n and m.tolist() == list(range(n))
    # every other query is a landmark without observations: it takes a keypoint without holding it
    has_obs = (np.arange(nq) % 2).astype(np.uint8)
    q2 = A.make_proj_queries(xy, sl, qd, angle=np.zeros(nq, np.float32), has_observation=has_obs)
    run_both("match_frame_and_landmarks", (view.ref(), q2.ref(), None, 20.0, 0.8), n)
    run_both("match_current_and_last_frames", (view.ref(), q2.ref(), None, 20.0, A.DIR_NONE, 8, 0.9, 0), n)


@pytest.mark.parametrize("kind", ["current_and_last", "frame_and_landmarks", "sim3"])
def test_searched_frame_beyond_the_lds_plan(kind):
    """More keypoints in the searched frame (24 000) than the resolve kernel can keep per-keypoint state
    for in LDS: its claim bookkeeping then lives in global memory and queries retire one per round."""
    rng = np.random.default_rng(31)
    n, nq, w, h = 24000, 9000, 4000.0, 3000.0
    kps, desc, sxr = random_frame(rng, n, w=w, h=h, cluster=True)
    xy, sl, qd, valid, xr, ang, has_obs = make_queries(rng, kps, desc, sxr, nq, jitter=3.0)
    grid = A.make_grid(0.0, 0.0, w, h)
    view = A.make_frame_view(kps, desc, SF, grid, sxr)
    q = A.make_proj_queries(xy, sl, qd, valid, xr, ang, has_obs)
    occupied = (rng.random(n) < 0.1).astype(np.uint8)
    if kind == "current_and_last":
        m, num = run_both("match_current_and_last_frames", (view.ref(), q.ref(), A.ptr(occupied, A.u8p), 20.0, A.DIR_NONE, 8, 0.9, 1), n)
    elif kind == "frame_and_landmarks":
        m, num = run_both("match_frame_and_landmarks", (view.ref(), q.ref(), A.ptr(occupied, A.u8p), 10.0, 0.8), n)
    else:
        m, num = run_both("match_by_sim3_transform", (view.ref(), q.ref(), A.ptr(occupied, A.u8p), 15.0), n)
    assert num > 500

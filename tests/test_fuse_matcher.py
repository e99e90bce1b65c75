"""The candidate search of match::fuse (src/openvslam/match/fuse.cc:67-102, 173-236): known
answers on the CPU oracle, GPU parity against it."""
import numpy as np
import pytest

import oracle
from openvslam_amd import _abi as A

SF = np.array([1.2 ** i for i in range(8)], dtype=np.float32)
ISIG = (1.0 / (SF * SF)).astype(np.float32)
GRID = A.make_grid(0.0, 0.0, 1241.0, 376.0)


def oracle_fuse(view, q, margin, rd=None, isig=None):
    best = np.full(q.struct.n, -9, np.int32)
    oracle.check(oracle.lib().ovo_fuse_best_candidates(view.ref(), q.ref(), A.ptr(rd, A.f64p), A.ptr(isig, A.f32p), np.float32(margin),
                                                       0 if rd is None else 1, A.ptr(best, A.i32p)))
    return best


def test_level_gate_is_signed_in_detect_and_unsigned_in_replace():
    """A landmark predicted at level 0 over a level-0 keypoint: detect_duplication compares ints
    (-1 <= 0 passes, fuse.cc:83-88), replace_duplication compares unsigned values, where
    `0 < pred - 1` wraps and rejects everything (fuse.cc:191-196)."""
    k = np.zeros(2, dtype=A.KEYPOINT_DTYPE)
    k["x"], k["y"], k["octave"] = [100, 300], [100, 100], [0, 1]
    d = np.zeros((2, 32), np.uint8)
    view = A.make_frame_view(k, d, SF, GRID, stereo_x_right=np.full(2, -1.0, np.float32))
    xy = np.array([[100.0, 100.0], [300.0, 100.0]], np.float32)
    q = A.make_proj_queries(xy, np.array([0, 1], np.int32), d.copy(), x_right=np.full(2, -1.0, np.float32))
    assert oracle_fuse(view, q, 3.0).tolist() == [0, 1]
    assert oracle_fuse(view, q, 3.0, xy.astype(np.float64), ISIG).tolist() == [-1, 1]


def test_chi_square_gate():
    """Two degrees of freedom: error^2 * inv_sigma^2 must not exceed 5.99146 (fuse.cc:213-221); with
    a stereo keypoint the x_right error joins and the bound is 7.81473 (fuse.cc:198-210)."""
    k = np.zeros(1, dtype=A.KEYPOINT_DTYPE)
    k["x"], k["y"], k["octave"] = 200.0, 100.0, 1
    d = np.zeros((1, 32), np.uint8)
    qd = d.copy()
    lvl = np.array([1], np.int32)
    for sxr, xr, dx, want in ((-1.0, -1.0, 2.9, 0), (-1.0, -1.0, 3.0, -1), (150.0, 150.0, 3.3, 0), (150.0, 153.0, 1.6, -1)):
        view = A.make_frame_view(k, d, SF, GRID, stereo_x_right=np.array([sxr], np.float32))
        rd = np.array([[200.0 + dx, 100.0]], np.float64)
        q = A.make_proj_queries(rd.astype(np.float32), lvl, qd, x_right=np.array([xr], np.float32))
        # sigma^2 at level 1 = 1.44: 2.9^2 / 1.44 = 5.84 passes, 3.0^2 / 1.44 = 6.25 fails;
        # 3.3^2 / 1.44 = 7.56 passes with three degrees, (1.6^2 + 3^2) / 1.44 = 8.03 fails
        assert oracle_fuse(view, q, 10.0, rd, ISIG).tolist() == [want], (sxr, xr, dx)


def random_case(seed, n, nq, stereo):
    rng = np.random.default_rng(seed)
    k = np.zeros(n, dtype=A.KEYPOINT_DTYPE)
    k["x"], k["y"] = rng.uniform(0, 1241, n), rng.uniform(0, 376, n)
    k["octave"] = rng.integers(0, 8, n)
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    sxr = np.where(rng.random(n) < (0.5 if stereo else 0.0), k["x"] - rng.uniform(1, 40, n), -1.0).astype(np.float32)
    src = rng.integers(0, n, nq)
    qd = d[src].copy()
    for i in range(nq):
        for b in rng.choice(256, rng.integers(0, 70), replace=False):
            qd[i, b >> 3] ^= np.uint8(1 << (b & 7))
    rd = np.stack([k["x"][src] + rng.normal(0, 2.0, nq), k["y"][src] + rng.normal(0, 2.0, nq)], axis=1).astype(np.float64)
    lvl = np.clip(k["octave"][src] + rng.integers(-1, 2, nq), 0, 7).astype(np.int32)
    xr = (sxr[src] + rng.normal(0, 1.5, nq)).astype(np.float32)
    valid = (rng.random(nq) < 0.9).astype(np.uint8)
    view = A.make_frame_view(k, d, SF, GRID, stereo_x_right=sxr)
    q = A.make_proj_queries(rd.astype(np.float32), lvl, qd, valid=valid, x_right=xr)
    return view, q, rd


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,nq,stereo,margin", [(1, 2000, 3000, True, 3.0), (2, 2500, 1000, False, 4.0), (3, 40, 200, True, 10.0), (4, 0, 5, False, 3.0)])
def test_gpu_matches_oracle(seed, n, nq, stereo, margin):
    from openvslam_amd import match
    view, q, rd = random_case(seed, max(n, 0), nq, stereo) if n else (None, None, None)
    if n == 0:
        k = np.zeros(0, dtype=A.KEYPOINT_DTYPE)
        view = A.make_frame_view(k, np.zeros((0, 32), np.uint8), SF, GRID)
        rd = np.zeros((nq, 2), np.float64) + 50.0
        q = A.make_proj_queries(rd.astype(np.float32), np.zeros(nq, np.int32), np.zeros((nq, 32), np.uint8), x_right=np.zeros(nq, np.float32))
    f = match.fuse()
    det = f.best_candidates(view, q, margin)
    assert np.array_equal(det, oracle_fuse(view, q, margin))
    rep = f.best_candidates(view, q, margin, rd, ISIG)
    assert np.array_equal(rep, oracle_fuse(view, q, margin, rd, ISIG))
    if n >= 2000:
        assert (det >= 0).sum() > nq // 10 and (rep >= 0).sum() > nq // 20 and not np.array_equal(det, rep)

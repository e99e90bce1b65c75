"""match::area::match_in_consistent_area (src/openvslam/match/area.cc:8-112): first-principles
checks of the oracle on the CPU, GPU parity against it on extracted frames."""
import ctypes as C

import numpy as np
import pytest

import oracle
from openvslam_amd import _abi as A

SF = np.array([1.2 ** i for i in range(8)], dtype=np.float32)


def oracle_area(k1, d1, view2, prev, margin, lowe, check):
    prev = prev.copy()
    m = np.full(len(k1), -7, np.int32)
    num = C.c_uint32(0)
    oracle.check(oracle.lib().ovo_match_in_consistent_area(A.ptr(k1, A.kpp), A.ptr(d1, A.u8p), len(k1), view2.ref(), A.ptr(prev, A.f32p),
                                                           margin, np.float32(lowe), check, A.ptr(m, A.i32p), C.byref(num)))
    return num.value, m, prev


def test_identity_and_level_gate():
    """A frame matched with itself around its own keypoints: every octave-0 keypoint matches itself,
    higher octaves never search (area.cc:24-26), the search centres move onto the matches."""
    rng = np.random.default_rng(3)
    n = 300
    k = np.zeros(n, dtype=A.KEYPOINT_DTYPE)
    k["x"], k["y"] = rng.uniform(10, 630, n), rng.uniform(10, 470, n)
    k["octave"] = rng.integers(0, 3, n)
    k["angle"] = rng.uniform(0, 360, n)
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    view = A.make_frame_view(k, d, SF, A.make_grid(0.0, 0.0, 640.0, 480.0))
    prev = np.stack([k["x"], k["y"]], axis=1).astype(np.float32) + np.float32(1.5)
    num, m, new_prev = oracle_area(k, d, view, prev, 30, 0.9, 1)
    lvl0 = k["octave"] == 0
    assert np.array_equal(m[lvl0], np.nonzero(lvl0)[0]) and np.all(m[~lvl0] == -1)
    assert num == int(lvl0.sum())
    assert np.array_equal(new_prev[lvl0, 0], k["x"][lvl0]) and np.array_equal(new_prev[~lvl0], prev[~lvl0])


def test_closer_keypoint_takes_over():
    """area.cc:46-49,75-85: a later keypoint that is strictly closer to a candidate replaces the
    earlier match; an equally distant one does not."""
    k1 = np.zeros(3, dtype=A.KEYPOINT_DTYPE)
    k1["x"], k1["y"] = 100.0, 100.0
    k2 = np.zeros(1, dtype=A.KEYPOINT_DTYPE)
    k2["x"], k2["y"] = 101.0, 100.0
    d2 = np.zeros((1, 32), np.uint8)
    d1 = np.zeros((3, 32), np.uint8)
    d1[0, 0] = 0x0F   # distance 4
    d1[1, 0] = 0x03   # distance 2: takes over
    d1[2, 0] = 0x30   # distance 2: not strictly closer
    view = A.make_frame_view(k2, d2, SF, A.make_grid(0.0, 0.0, 640.0, 480.0))
    prev = np.full((3, 2), 100.0, np.float32)
    num, m, _ = oracle_area(k1, d1, view, prev, 10, 0.9, 0)
    assert num == 1 and m.tolist() == [-1, 0, -1]


@pytest.mark.gpu
@pytest.mark.parametrize("check,lowe,margin", [(1, 0.9, 100), (0, 0.75, 30), (1, 1.0, 300)])
def test_gpu_matches_oracle_on_a_sequence(check, lowe, margin):
    from openvslam_amd import match, synth
    frames = synth.make_sequence(640, 480, 2, seed=12, dx=5, dy=2)
    ext = oracle.OrbExtractor(max_num_keypts=1500)
    k1, d1 = ext.extract(frames[0])
    k2, d2 = ext.extract(frames[1])
    view2 = A.make_frame_view(k2, d2, SF, A.make_grid(0.0, 0.0, 640.0, 480.0))
    prev = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32))
    want_num, want_m, want_prev = oracle_area(k1, d1, view2, prev, margin, lowe, check)
    got_prev = prev.copy()
    got_num, got_m = match.area(lowe, bool(check)).match_in_consistent_area(k1, d1, view2, got_prev, margin)
    assert got_num == want_num and np.array_equal(got_m, want_m)
    assert got_prev.tobytes() == want_prev.tobytes()
    assert got_num > 50


@pytest.mark.gpu
def test_gpu_closer_keypoint_takes_over():
    """The three-query case of test_closer_keypoint_takes_over through the device path."""
    from openvslam_amd import match
    k1 = np.zeros(3, dtype=A.KEYPOINT_DTYPE)
    k1["x"], k1["y"] = 100.0, 100.0
    k2 = np.zeros(1, dtype=A.KEYPOINT_DTYPE)
    k2["x"], k2["y"] = 101.0, 100.0
    d2 = np.zeros((1, 32), np.uint8)
    d1 = np.zeros((3, 32), np.uint8)
    d1[0, 0], d1[1, 0], d1[2, 0] = 0x0F, 0x03, 0x30
    view = A.make_frame_view(k2, d2, SF, A.make_grid(0.0, 0.0, 640.0, 480.0))
    prev = np.full((3, 2), 100.0, np.float32)
    num, m = match.area(0.9, False).match_in_consistent_area(k1, d1, view, prev, 10)
    assert num == 1 and m.tolist() == [-1, 1 - 1, -1]
    assert prev.tolist() == [[100.0, 100.0], [101.0, 100.0], [100.0, 100.0]]


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n1,n2,check,lowe,margin", [(1, 3000, 200, 1, 0.9, 60), (2, 1500, 40, 0, 1.0, 200), (3, 2500, 900, 1, 0.8, 25)])
def test_gpu_contested_keypoints(seed, n1, n2, check, lowe, margin):
    """Many keypoints of frame 1 compete for few keypoints of frame 2 (area.cc:46-49,75-85): matches
    change hands repeatedly, windows hold far more candidates than a query's candidate list, and the
    rotation histogram keeps the entries of matches that were taken over later (area.cc:87-91)."""
    from openvslam_amd import match
    rng = np.random.default_rng(seed)
    k2 = np.zeros(n2, dtype=A.KEYPOINT_DTYPE)
    k2["x"], k2["y"] = rng.uniform(200, 330, n2), rng.uniform(150, 260, n2)
    k2["angle"] = rng.uniform(0, 360, n2)
    k2["octave"] = rng.integers(0, 2, n2)
    d2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
    src = rng.integers(0, n2, n1)
    d1 = d2[src].copy()
    for i in range(n1):
        for b in rng.choice(256, rng.integers(0, 45), replace=False):
            d1[i, b >> 3] ^= np.uint8(1 << (b & 7))
    k1 = np.zeros(n1, dtype=A.KEYPOINT_DTYPE)
    k1["x"] = k2["x"][src] + rng.normal(0, 8, n1).astype(np.float32)
    k1["y"] = k2["y"][src] + rng.normal(0, 8, n1).astype(np.float32)
    k1["angle"] = (k2["angle"][src] + rng.normal(0, 25, n1)) % 360
    k1["octave"] = (rng.random(n1) < 0.1).astype(np.int32)
    view2 = A.make_frame_view(k2, d2, SF, A.make_grid(0.0, 0.0, 640.0, 480.0))
    prev = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32))
    want_num, want_m, want_prev = oracle_area(k1, d1, view2, prev, margin, lowe, check)
    got_prev = prev.copy()
    got_num, got_m = match.area(lowe, bool(check)).match_in_consistent_area(k1, d1, view2, got_prev, margin)
    assert got_num == want_num and np.array_equal(got_m, want_m)
    assert got_prev.tobytes() == want_prev.tobytes()
    assert 0 < got_num <= n2


@pytest.mark.gpu
def test_gpu_searched_frame_beyond_the_lds_plan():
    """More than about 18 000 keypoints in the searched frame: the take-over state of the resolve kernel
    no longer fits in LDS and lives in global memory, one query resolved per round.  The reference has no
    limit here (area.cc:8-112)."""
    from openvslam_amd import match
    rng = np.random.default_rng(9)
    n2, n1 = 20000, 2500
    k2 = np.zeros(n2, dtype=A.KEYPOINT_DTYPE)
    k2["x"], k2["y"] = rng.uniform(5, 635, n2), rng.uniform(5, 475, n2)
    k2["angle"] = rng.uniform(0, 360, n2)
    d2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
    src = rng.integers(0, 400, n1)  # many queries per target: matches change hands
    d1 = d2[src].copy()
    for i in range(n1):
        for b in rng.choice(256, rng.integers(0, 40), replace=False):
            d1[i, b >> 3] ^= np.uint8(1 << (b & 7))
    k1 = np.zeros(n1, dtype=A.KEYPOINT_DTYPE)
    k1["x"] = k2["x"][src] + rng.normal(0, 4, n1).astype(np.float32)
    k1["y"] = k2["y"][src] + rng.normal(0, 4, n1).astype(np.float32)
    k1["angle"] = (k2["angle"][src] + rng.normal(0, 20, n1)) % 360
    view2 = A.make_frame_view(k2, d2, SF, A.make_grid(0.0, 0.0, 640.0, 480.0))
    prev = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], axis=1).astype(np.float32))
    want_num, want_m, want_prev = oracle_area(k1, d1, view2, prev, 20, 0.9, 1)
    got_prev = prev.copy()
    got_num, got_m = match.area(0.9, True).match_in_consistent_area(k1, d1, view2, got_prev, 20)
    assert got_num == want_num and np.array_equal(got_m, want_m)
    assert got_prev.tobytes() == want_prev.tobytes()
    assert 100 < got_num <= 400

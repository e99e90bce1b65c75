"""The host-side mirrors of feature::orb_extractor and match::* (openvslam_amd.feature / .match)
drive the same C ABI the parity tests call directly; here they are checked end to end on a
two-frame sequence: extract, re-project, match, and compare with the oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle
from openvslam_amd import _abi as A
from openvslam_amd import match, synth
from openvslam_amd.feature import orb_extractor, orb_params

pytestmark = pytest.mark.gpu


def test_extractor_setters_reinitialize():
    ext = orb_extractor(orb_params())
    assert ext.get_max_num_keypoints() == 2000 and ext.get_num_scale_levels() == 8
    assert abs(ext.get_scale_factor() - 1.2) < 1e-6
    assert ext.get_initial_fast_threshold() == 20 and ext.get_minimum_fast_threshold() == 7
    per_level = ext.get_num_keypts_per_level()
    assert per_level.sum() == 2000 or abs(int(per_level.sum()) - 2000) <= 8
    ext.set_max_num_keypoints(1000)
    ext.set_num_scale_levels(6)
    ext.set_scale_factor(1.3)
    ext.set_initial_fast_threshold(30)
    ext.set_minimum_fast_threshold(9)
    assert len(ext.get_scale_factors()) == 6 and abs(ext.get_scale_factors()[5] - 1.3 ** 5) < 1e-4
    img = synth.make_frame(640, 480, seed=2)
    got_k, got_d = ext.extract(img)
    o = oracle.OrbExtractor(max_num_keypts=1000, num_levels=6, scale_factor=1.3, ini_fast_thr=30, min_fast_thr=9)
    want_k, want_d = o.extract(img)
    assert got_k.tobytes() == want_k.tobytes() and np.array_equal(got_d, want_d)
    assert np.array_equal(ext.get_num_keypts_per_level(), o.num_keypts_per_level())


def test_projection_and_robust_mirrors_on_a_sequence():
    frames = synth.make_sequence(640, 480, 2, seed=8, dx=4, dy=2)
    ext = orb_extractor(orb_params(max_num_keypts=1000))
    k0, d0 = ext.extract(frames[0])
    k1, d1 = ext.extract(frames[1])
    sf = ext.get_scale_factors()
    grid = A.make_grid(0.0, 0.0, 640.0, 480.0)
    view = A.make_frame_view(k1, d1, sf, grid)
    xy = np.stack([k0["x"] - 4.0, k0["y"] - 2.0], axis=1).astype(np.float32)
    q = A.make_proj_queries(xy, k0["octave"], d0, angle=k0["angle"])
    num, matched = match.projection(0.9, True).match_current_and_last_frames(view, q, 15.0)
    want = np.full(len(k1), -1, np.int32)
    wn = C.c_uint32()
    oracle.check(oracle.lib().ovo_match_current_and_last_frames(view.ref(), q.ref(), None, np.float32(15.0), A.DIR_NONE, 8, np.float32(0.9), 1,
                                                                A.ptr(want, A.i32p), C.byref(wn)))
    assert num == wn.value and np.array_equal(matched, want)
    assert num > 300
    pairs = match.robust(0.75, True).brute_force_match(d1, k1["angle"], d0, k0["angle"])
    wp = np.zeros((len(k1), 2), np.int32)
    a1, a0 = np.ascontiguousarray(k1["angle"]), np.ascontiguousarray(k0["angle"])  # kept alive across the call
    oracle.check(oracle.lib().ovo_brute_force_match(A.ptr(d1, A.u8p), A.ptr(a1, A.f32p), len(k1),
                                                    A.ptr(d0, A.u8p), A.ptr(a0, A.f32p), None, len(k0),
                                                    np.float32(0.75), 1, A.ptr(wp, A.i32p), C.byref(wn)))
    assert np.array_equal(pairs, wp[:wn.value])
    assert len(pairs) > 100


def test_stereo_mirror():
    left, right, _, _ = synth.make_stereo_pair(640, 480, seed=3)
    el, er = orb_extractor(orb_params(max_num_keypts=800)), orb_extractor(orb_params(max_num_keypts=800))
    kl, dl = el.extract(left)
    kr, dr = er.extract(right)
    pl = [left] + [el.image_pyramid(i) for i in range(1, 8)]
    pr = [right] + [er.image_pyramid(i) for i in range(1, 8)]
    sf, isf = el.get_scale_factors(), el.get_inv_scale_factors()
    xr, dep = match.stereo(pl, pr, kl, kr, dl, dr, sf, isf, 400.0, 0.5).compute()
    wx = np.zeros(len(kl), np.float32)
    wd = np.zeros(len(kl), np.float32)
    views = match.stereo._views
    oracle.check(oracle.lib().ovo_stereo_compute(views(pl), views(pr), 8, A.ptr(kl, A.kpp), len(kl), A.ptr(kr, A.kpp), len(kr),
                                                 A.ptr(dl, A.u8p), A.ptr(dr, A.u8p), A.ptr(sf, A.f32p), A.ptr(isf, A.f32p),
                                                 np.float32(400.0), np.float32(0.5), A.ptr(wx, A.f32p), A.ptr(wd, A.f32p)))
    assert xr.tobytes() == wx.tobytes() and dep.tobytes() == wd.tobytes()
    assert (dep > 0).sum() > 100

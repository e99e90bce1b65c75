"""Several host threads in the library at once, as the reference runs it: the tracking, mapping and
global optimisation threads all call the matchers (system.cc:126-127, mapping_module.cc:254,
module/loop_detector.cc:290) and a stereo frame extracts its two images on two threads, each on its own
extractor (data/frame.cc:64-67).  ctypes releases the GIL around every call, so the four threads below are
inside the library together; every result of every iteration must equal the oracle's, which is computed
beforehand on one thread."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import oracle
from openvslam_amd import _abi as A
from openvslam_amd import _lib, synth
from openvslam_amd.feature import orb_extractor, orb_params
from test_gpu_projection import GRID, SF, make_queries, random_frame
from test_gpu_robust_bow import clustered_sets, two_sets

pytestmark = pytest.mark.gpu

ROUNDS = int(os.environ.get("OVS_TEST_THREAD_ROUNDS", "20"))  # fewer when the library runs on the host emulator


def _oracle_call(name, args, n_out, shape=None):
    m = np.full(shape or n_out, -7, dtype=np.int32)
    num = C.c_uint32(12345)
    assert getattr(oracle.lib(), "ovo_" + name)(*args, A.ptr(m, A.i32p), C.byref(num)) == 0
    return m, num.value


def _device_call(name, args, n_out, shape=None):
    m = np.full(shape or n_out, -7, dtype=np.int32)
    num = C.c_uint32(12345)
    rc = getattr(_lib.lib(), "ovs_" + name)(*args, A.ptr(m, A.i32p), C.byref(num), 0)
    if rc != 0:
        raise RuntimeError("%s: rc %d: %s" % (name, rc, _lib.lib().ovs_last_error()))
    return m, num.value


def test_four_host_threads_share_the_library():
    rng = np.random.default_rng(2024)
    # --- the two extraction threads: a left and a right sequence, one extractor handle each
    left = synth.make_sequence(752, 480, 4, seed=31, dx=5, dy=1)
    right = synth.make_sequence(752, 480, 4, seed=32, dx=4, dy=2)
    ref = oracle.OrbExtractor(max_num_keypts=1000)
    want_left = [ref.extract(f) for f in left]
    want_right = [ref.extract(f) for f in right]
    ext_left = orb_extractor(orb_params(max_num_keypts=1000))
    ext_right = orb_extractor(orb_params(max_num_keypts=1000))

    # --- projection::match_frame_and_landmarks (tracking thread, tracking_module.cc:585-591)
    kps, desc, sxr = random_frame(rng, 2000, stereo=True, cluster=True)
    xy, sl, qd, valid, xr, ang, has_obs = make_queries(rng, kps, desc, sxr, 4000, jitter=2.0)
    view = A.make_frame_view(kps, desc, SF, GRID, sxr)
    q = A.make_proj_queries(xy, sl, qd, valid, xr, ang, has_obs)
    occupied = (rng.random(2000) < 0.1).astype(np.uint8)
    lm_args = (view.ref(), q.ref(), A.ptr(occupied, A.u8p), 5.0, 0.8)
    want_lm = _oracle_call("match_frame_and_landmarks", lm_args, 2000)

    # --- robust::brute_force_match (robust.cc:209-315) and bow_tree::match_keyframes (loop detector)
    d1, a1, d2, a2 = two_sets(rng, 1500, 1800)
    valid2 = (rng.random(1800) < 0.85).astype(np.uint8)
    bf_args = (A.ptr(d1, A.u8p), A.ptr(a1, A.f32p), 1500, A.ptr(d2, A.u8p), A.ptr(a2, A.f32p), A.ptr(valid2, A.u8p), 1800, 0.75, 1)
    want_bf = _oracle_call("brute_force_match", bf_args, 0, shape=(1500, 2))
    c1, c2, v1, v2 = clustered_sets(rng, 1200, 1400, 60)
    ca1 = rng.uniform(0, 360, 1200).astype(np.float32)
    ca2 = rng.uniform(0, 360, 1400).astype(np.float32)
    cv1 = (rng.random(1200) < 0.8).astype(np.uint8)
    cv2 = (rng.random(1400) < 0.8).astype(np.uint8)
    bow_args = (v1.ref(), A.ptr(c1, A.u8p), A.ptr(ca1, A.f32p), A.ptr(cv1, A.u8p), 1200,
                v2.ref(), A.ptr(c2, A.u8p), A.ptr(ca2, A.f32p), A.ptr(cv2, A.u8p), 1400, 0.75, 1)
    want_bow = _oracle_call("bow_match_keyframes", bow_args, 1200)
    assert want_lm[1] > 100 and want_bf[1] > 100 and want_bow[1] > 20

    start = threading.Barrier(4)
    errors = []

    def guarded(fn):
        def run():
            try:
                start.wait(timeout=60)
                for it in range(ROUNDS):
                    fn(it)
            except Exception as e:  # noqa: BLE001 -- reported by the main thread
                errors.append("%s: %r" % (fn.__name__, e))
        return run

    def extract_with(ext, frames, want):
        def body(it):
            k, d = ext.extract(frames[it % len(frames)])
            wk, wd = want[it % len(frames)]
            if len(k) != len(wk) or k.tobytes() != wk.tobytes() or not np.array_equal(d, wd):
                raise AssertionError("iteration %d: extraction differs from the oracle's" % it)
        return body

    def extract_left(it):
        extract_with(ext_left, left, want_left)(it)

    def extract_right(it):
        extract_with(ext_right, right, want_right)(it)

    def landmarks(it):
        m, n = _device_call("match_frame_and_landmarks", lm_args, 2000)
        if n != want_lm[1] or not np.array_equal(m, want_lm[0]):
            raise AssertionError("iteration %d: match_frame_and_landmarks differs" % it)

    def robust_and_bow(it):
        m, n = _device_call("brute_force_match", bf_args, 0, shape=(1500, 2))
        if n != want_bf[1] or not np.array_equal(m[:n], want_bf[0][:n]):
            raise AssertionError("iteration %d: brute_force_match differs" % it)
        m, n = _device_call("bow_match_keyframes", bow_args, 1200)
        if n != want_bow[1] or not np.array_equal(m, want_bow[0]):
            raise AssertionError("iteration %d: bow_match_keyframes differs" % it)

    threads = [threading.Thread(target=guarded(f), name=f.__name__) for f in (extract_left, extract_right, landmarks, robust_and_bow)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=100)
    assert not any(t.is_alive() for t in threads), "a thread did not finish"
    assert not errors, errors


def test_stereo_frame_on_two_threads_then_match():
    """data/frame.cc:64-81: the two extractions of a stereo frame run concurrently, each on its own
    extractor, then match::stereo reads both extractors' pyramids (here: where they are, on the device).
    Ten frames in a row; stereo_x_right and depths are compared as bit patterns."""
    from test_gpu_stereo import FX, FXB, _views
    w, h, n_kp = 752, 480, 1000
    pairs = [synth.make_stereo_pair(w, h, seed=50 + i)[:2] for i in range(3)]
    ext_l = orb_extractor(orb_params(max_num_keypts=n_kp))
    ext_r = orb_extractor(orb_params(max_num_keypts=n_kp))
    ref_l, ref_r = oracle.OrbExtractor(max_num_keypts=n_kp), oracle.OrbExtractor(max_num_keypts=n_kp)
    sf = oracle.scale_table("scale_factors", 8, 1.2)
    isf = oracle.scale_table("inv_scale_factors", 8, 1.2)
    true_baseline = np.float32(FXB / FX)
    want = []
    for il, ir in pairs:
        (kl, dl), (kr, dr) = ref_l.extract(il), ref_r.extract(ir)
        la, keep_l = _views([ref_l.pyramid_level(i) for i in range(8)])
        ra, keep_r = _views([ref_r.pyramid_level(i) for i in range(8)])
        xr = np.full(len(kl), 7.0, dtype=np.float32)
        dep = np.full(len(kl), 7.0, dtype=np.float32)
        oracle.check(oracle.lib().ovo_stereo_compute(la, ra, 8, A.ptr(kl, A.kpp), len(kl), A.ptr(kr, A.kpp), len(kr), A.ptr(dl, A.u8p), A.ptr(dr, A.u8p),
                                                     A.ptr(sf, A.f32p), A.ptr(isf, A.f32p), np.float32(FXB), true_baseline, A.ptr(xr, A.f32p), A.ptr(dep, A.f32p)))
        assert (xr >= 0).sum() > 100
        want.append((kl, dl, kr, dr, xr, dep))
    for it in range(10):
        il, ir = pairs[it % len(pairs)]
        out = {}

        def run(name, ext, img):
            out[name] = ext.extract(img)
        threads = [threading.Thread(target=run, args=("l", ext_l, il)), threading.Thread(target=run, args=("r", ext_r, ir))]
        for t in threads:
            t.start()
        for t in threads:
            t.join(60)
        kl, dl, kr, dr, wxr, wdep = want[it % len(pairs)]
        assert out["l"][0].tobytes() == kl.tobytes() and np.array_equal(out["l"][1], dl)
        assert out["r"][0].tobytes() == kr.tobytes() and np.array_equal(out["r"][1], dr)
        (gkl, gdl), (gkr, gdr) = out["l"], out["r"]
        xr = np.full(len(gkl), 7.0, dtype=np.float32)
        dep = np.full(len(gkl), 7.0, dtype=np.float32)
        _lib.check(_lib.lib().ovs_stereo_compute_resident(ext_l._h, ext_r._h, A.ptr(gkl, A.kpp), len(gkl), A.ptr(gkr, A.kpp), len(gkr),
                                                          A.ptr(gdl, A.u8p), A.ptr(gdr, A.u8p), np.float32(FXB), true_baseline,
                                                          A.ptr(xr, A.f32p), A.ptr(dep, A.f32p), 0))
        assert xr.view(np.uint32).tobytes() == wxr.view(np.uint32).tobytes(), "iteration %d: stereo_x_right differs" % it
        assert dep.view(np.uint32).tobytes() == wdep.view(np.uint32).tobytes(), "iteration %d: depths differ" % it

#!/usr/bin/env python3
"""Secondary measurements (not the headline line of bench.py): the other configurations of the
path at their full sizes, inputs resident in HBM, one JSON line each.

  stereo   752x480 pair, 1000 keypoints per eye: 2 x extract + match::stereo        (pairs/s)
  equirect 1920x960 frames, 4000 keypoints, rectangle masks: extract + match        (frames/s)
  knn      2000 queries against one shard of 1250 keyframes x 2000 descriptors      (distances/s)
  live     one frame at a time through the host-buffer entry points                  (ms per frame)
  pre      BGR -> gray, rectification remap, histogram stretch on 256 raw 1241x376 frames (frames/s)
  post     keypoint undistortion + bearings of 256 frames x 2250 keypoints               (keypoints/s)
  bow      BoW transform of 256 frames x 2000 descriptors, k = 10 / L = 6 vocabulary          (frames/s)
"""
import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from openvslam_amd import _abi as A  # noqa: E402
from openvslam_amd import _lib, match, synth  # noqa: E402
from openvslam_amd.feature import orb_extractor, orb_params  # noqa: E402


def timed(fn, steps, warmup):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def stereo(args):
    dev = torch.device("cuda", 0)
    w, h, n_kp, batch = 752, 480, 1000, 128
    fxb, fx = 47.90639384423901, 435.2046959714599
    pairs = [synth.make_stereo_pair(w, h, seed=100 + i) for i in range(4)]
    left = torch.from_numpy(np.stack([pairs[i % 4][0] for i in range(batch)])).to(dev)
    right = torch.from_numpy(np.stack([pairs[i % 4][1] for i in range(batch)])).to(dev)
    el, er = orb_extractor(orb_params(max_num_keypts=n_kp)), orb_extractor(orb_params(max_num_keypts=n_kp))
    cap = el.keypoint_capacity(w, h)
    bufs = [(torch.zeros((batch, cap, 7), dtype=torch.float32, device=dev), torch.zeros((batch, cap, 32), dtype=torch.uint8, device=dev),
             torch.zeros((batch,), dtype=torch.int32, device=dev)) for _ in range(2)]
    xr = torch.zeros((batch, cap), dtype=torch.float32, device=dev)
    dep = torch.zeros((batch, cap), dtype=torch.float32, device=dev)
    status = torch.zeros((1,), dtype=torch.int32, device=dev)
    L = _lib.lib()

    # the two extractions on two streams, as data::frame runs them on two threads (frame.cc:64-67);
    # the matcher waits for both
    right_stream = torch.cuda.Stream(dev)
    right_done = torch.cuda.Event()

    def step():
        main = torch.cuda.current_stream(dev)
        st = main.cuda_stream
        right_stream.wait_stream(main)  # the previous step's matcher still reads the right pyramid
        k, d, c = bufs[1]
        er.extract_batch_dev(right.data_ptr(), batch, w, h, w, w * h, k.data_ptr(), d.data_ptr(), c.data_ptr(), cap, right_stream.cuda_stream)
        right_done.record(right_stream)
        k, d, c = bufs[0]
        el.extract_batch_dev(left.data_ptr(), batch, w, h, w, w * h, k.data_ptr(), d.data_ptr(), c.data_ptr(), cap, st)
        main.wait_event(right_done)
        _lib.check(L.ovs_stereo_compute_batch_dev(el._h, er._h, batch, bufs[0][0].data_ptr(), bufs[0][1].data_ptr(), bufs[0][2].data_ptr(),
                                                  bufs[1][0].data_ptr(), bufs[1][1].data_ptr(), bufs[1][2].data_ptr(), cap,
                                                  np.float32(fxb), np.float32(fxb / fx), xr.data_ptr(), dep.data_ptr(), status.data_ptr(), 0, st))
    dt = timed(step, args.steps, args.warmup)
    valid = float((dep > 0).sum().item()) / batch
    return {"metric": "stereo pairs/s (2 x extract 752x480 1000 kp + match::stereo)", "value": batch / dt, "unit": "pairs/s",
            "ms_per_step": dt * 1e3, "pairs_per_step": batch, "mean_stereo_matches_per_pair": valid}


def equirect(args):
    dev = torch.device("cuda", 0)
    w, h, n_kp, batch = 1920, 960, 4000, 255  # + 1 halo frame = 256 frames per launch
    rects = [(0.0, 1.0, 0.0, 0.1), (0.0, 1.0, 0.84, 1.0), (0.0, 0.2, 0.7, 1.0), (0.8, 1.0, 0.7, 1.0)]
    frames = torch.from_numpy(synth.make_sequence(w, h, batch + 1, seed=77, dx=3, dy=1)).to(dev)
    ext = orb_extractor(orb_params(max_num_keypts=n_kp, mask_rects=rects))
    cap = ext.keypoint_capacity(w, h)
    n = batch + 1
    kps = torch.zeros((n, cap, 7), dtype=torch.float32, device=dev)
    desc = torch.zeros((n, cap, 32), dtype=torch.uint8, device=dev)
    counts = torch.zeros((n,), dtype=torch.int32, device=dev)
    reproj = torch.full((n, cap, 2), -1e38, dtype=torch.float32, device=dev)
    matched = torch.zeros((n, cap), dtype=torch.int32, device=dev)
    num = torch.zeros((n,), dtype=torch.int32, device=dev)
    shift = torch.tensor([3.0, 1.0], device=dev)
    grid = A.make_grid(0.0, 0.0, float(w), float(h))
    sf = ext.get_scale_factors()
    L = _lib.lib()

    def step():
        st = torch.cuda.current_stream(dev).cuda_stream
        ext.extract_batch_dev(frames.data_ptr(), n, w, h, w, w * h, kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), cap, st)
        torch.sub(kps[:-1, :, 0:2], shift, out=reproj[1:])
        _lib.check(L.ovs_match_batch_current_and_last_dev(kps.data_ptr(), desc.data_ptr(), counts.data_ptr(), n, cap, reproj.data_ptr(),
                                                          C.byref(grid), A.ptr(sf, A.f32p), 8, 20.0, 1, matched.data_ptr(), num.data_ptr(), 0, st))
    dt = timed(step, args.steps, args.warmup)
    return {"metric": "equirectangular 1920x960 frames/s (4000 kp, rectangle masks, extract + projection match)", "value": batch / dt,
            "unit": "frames/s", "ms_per_step": dt * 1e3, "frames_per_step": batch,
            "mean_keypoints_per_frame": float(counts.float().mean().item()), "mean_matches_per_frame": float(num[1:].float().mean().item())}


def knn(args):
    dev = torch.device("cuda", 0)
    rows, nq = 1250 * 2000, 2000
    gen = torch.Generator(device=dev)
    gen.manual_seed(5)
    db = torch.randint(0, 256, (rows, 32), dtype=torch.uint8, device=dev, generator=gen)
    q = torch.randint(0, 256, (nq, 32), dtype=torch.uint8, device=dev, generator=gen)
    out = torch.zeros(nq * 24, dtype=torch.uint8, device=dev)
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.ovs_desc_db_create(0, C.byref(h)))
    _lib.check(L.ovs_desc_db_attach_dev(h, db.data_ptr(), rows, 0))

    def step():
        _lib.check(L.ovs_desc_db_search_dev(h, q.data_ptr(), nq, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    dt = timed(step, args.steps, args.warmup)
    L.ovs_desc_db_destroy(h)
    dists = rows * nq
    return {"metric": "brute-force top-2 search, 2000 queries x 2.5 M rows (one of 8 shards)", "value": dists / dt, "unit": "distances/s",
            "ms_per_step": dt * 1e3, "database_gbs": rows * 32 / dt / 1e9, "popcount_word_ops_per_s": dists * 8 / dt}


def pre(args):
    """Image preparation in front of the extractor on a resident batch of raw frames: BGR -> gray,
    histogram stretch (in place), rectification remap.  Algorithmic bytes per pixel: 4, 3 and 8."""
    dev = torch.device("cuda", 0)
    from openvslam_amd import util
    w, h, batch = 1241, 376, 256
    gen = torch.Generator(device=dev)
    gen.manual_seed(1)
    bgr = torch.randint(0, 256, (batch, h, w, 3), dtype=torch.uint8, device=dev, generator=gen)
    gray = torch.zeros((batch, h, w), dtype=torch.uint8, device=dev)
    rect = torch.zeros((batch, h, w), dtype=torch.uint8, device=dev)
    flags = torch.zeros((batch,), dtype=torch.int32, device=dev)
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    x, y = (xs - w / 2) / (0.9 * w), (ys - h / 2) / (0.9 * w)
    k = 1 - 0.2 * (x * x + y * y)
    rm = util.remap((0.9 * w * 1.05 * x * k + w / 2).astype(np.float32), (0.9 * w * 1.05 * y * k + h / 2).astype(np.float32))
    L = _lib.lib()
    px = batch * w * h
    out = {}

    def run(name, fn, bytes_per_px):
        dt = timed(fn, args.steps, args.warmup)
        out[name] = {"ms": dt * 1e3, "frames_per_s": batch / dt, "algorithmic_gbs": px * bytes_per_px / dt / 1e9}

    st = torch.cuda.current_stream(dev).cuda_stream
    run("convert_to_grayscale", lambda: _lib.check(L.ovs_convert_to_grayscale_dev(bgr.data_ptr(), batch, w, h, 3, w * 3, w * h * 3, 2, gray.data_ptr(), w, w * h, 0, st)), 4)
    run("remap", lambda: rm.apply_batch_dev(gray.data_ptr(), batch, w, h, w, w * h, rect.data_ptr(), w, w * h, st), 8)
    run("equalize_histogram", lambda: _lib.check(L.ovs_equalize_histogram_dev(rect.data_ptr(), batch, w, h, w, w * h, flags.data_ptr(), 0, st)), 3)
    total_ms = sum(v["ms"] for v in out.values())
    return {"metric": "image preparation of 256 raw 1241x376 BGR frames (gray, rectify, equalize)", "value": batch / (total_ms * 1e-3), "unit": "frames/s",
            "ms_per_step": total_ms, "stages": out}


def post(args):
    """The step after the extractor on a resident batch: keypoint undistortion with the bearings
    written by the same launch (28 B in, 28 + 24 B out per keypoint), per camera model."""
    dev = torch.device("cuda", 0)
    from openvslam_amd import camera
    batch, cap, n_kp = 256, 2250, 2250
    rng = np.random.default_rng(1)
    kps = np.zeros(batch * cap, A.KEYPOINT_DTYPE)
    kps["x"] = rng.uniform(0, 752, len(kps)).astype(np.float32)
    kps["y"] = rng.uniform(0, 480, len(kps)).astype(np.float32)
    d_in = torch.from_numpy(kps.view(np.uint8)).to(dev)
    d_out = torch.zeros_like(d_in)
    d_bear = torch.zeros((batch, cap, 3), dtype=torch.float64, device=dev)
    cams = {"perspective (EuRoC, 20 iterations / 1e-6)": camera.perspective(458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0),
            "fisheye (TUM-VI)": camera.fisheye(190.97847715128717, 190.9733070521226, 254.93170605935475, 256.8974428996504,
                                               0.0034823894022493434, 0.0007150348452162257, -0.0020532361418706202, 0.00020293673591811182),
            "equirectangular (copy + sin / cos)": camera.equirectangular(1920, 960)}
    st = torch.cuda.current_stream(dev).cuda_stream
    out = {}
    reps = 50
    for name, cam in cams.items():
        def fn():
            cam.undistort_keypoints_dev(d_in.data_ptr(), None, batch, cap, d_out.data_ptr(), d_bear.data_ptr(), st)
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        out[name] = {"ms": ms, "keypoints_per_s": batch * n_kp / (ms * 1e-3), "algorithmic_gbs": batch * n_kp * 80 / (ms * 1e-3) / 1e9}
    first = next(iter(out.values()))
    return {"metric": "keypoint undistortion + bearings, 256 frames x 2250 keypoints resident", "value": first["keypoints_per_s"], "unit": "keypoints/s",
            "ms_per_step": first["ms"], "models": out}


def bow(args):
    """frame::compute_bow (data/frame.cc:162-170) on a resident batch: 256 frames x 2000 descriptors through a
    synthetic vocabulary of the reference's shape (k = 10, L = 6: 1.1 M nodes, levelsup 4), per-kernel times from
    the library's launch timing, and one frame through the host-buffer entry point; the oracle's transform of one
    frame on one core beside it."""
    dev = torch.device("cuda", 0)
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle
    from openvslam_amd import bow as ovs_bow
    from test_oracle_bow import noisy_leaf_descriptors, oracle_transform
    L = _lib.lib()
    voc = ovs_bow.synthetic_vocabulary(10, 6, seed=16)
    v = ovs_bow.bow_vocabulary(voc["k"], voc["L"], voc["node_descs"], voc["child_offsets"], voc["children"], voc["word_ids"], voc["weights"])
    n_frames, cap, n = 256, 2048, 2000
    one = noisy_leaf_descriptors(voc, n, seed=1)
    descs = np.zeros((n_frames, cap, 32), np.uint8)
    rng = np.random.default_rng(2)
    for f in range(n_frames):
        descs[f, :n] = one[rng.permutation(n)] ^ rng.integers(0, 2, (n, 32), dtype=np.uint8)
    d_descs = torch.from_numpy(descs).to(dev)
    d_counts = torch.full((n_frames,), n, dtype=torch.int32, device=dev)
    z = lambda shape, dt=torch.int32: torch.zeros(shape, dtype=dt, device=dev)  # noqa: E731
    word_of, node_of, bow_ids, fv_ids, fv_idx = z((n_frames, cap)), z((n_frames, cap)), z((n_frames, cap)), z((n_frames, cap)), z((n_frames, cap))
    bow_vals, fv_off, nw, nn = z((n_frames, cap), torch.float64), z((n_frames, cap + 1)), z(n_frames), z(n_frames)
    st = torch.cuda.current_stream(dev).cuda_stream

    def step():
        _lib.check(L.ovs_bow_transform_batch_dev(v._h, d_descs.data_ptr(), d_counts.data_ptr(), n_frames, cap, 4, word_of.data_ptr(), node_of.data_ptr(),
                                                 bow_ids.data_ptr(), bow_vals.data_ptr(), nw.data_ptr(), fv_ids.data_ptr(), fv_off.data_ptr(), fv_idx.data_ptr(),
                                                 nn.data_ptr(), st))
    dt = timed(step, args.steps, args.warmup)
    L.ovs_timing_reset()
    L.ovs_timing_enable(1)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    L.ovs_timing_enable(0)
    kernels = {k: ms / cnt for k, (ms, cnt) in _lib.timing_report().items()}
    t0 = time.perf_counter()
    for _ in range(20):
        v.transform(one, 4)
    live_ms = (time.perf_counter() - t0) / 20 * 1e3
    t0 = time.perf_counter()
    want = oracle_transform(voc, one, 4)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    (ids, vals), fv = v.transform(one, 4)
    same = np.array_equal(ids, want["word_ids"]) and vals.tobytes() == want["values"].tobytes() and np.array_equal(fv.keep[2], want["indices"])
    return {"metric": "BoW transform, 256 frames x 2000 descriptors resident, k = 10 / L = 6 vocabulary (1.1 M nodes), levelsup 4", "value": n_frames / dt,
            "unit": "frames/s", "ms_per_step": dt * 1e3, "kernel_ms_per_launch": kernels, "one_frame_host_buffers_ms": live_ms,
            "oracle_one_frame_one_core_ms": cpu_ms, "bit_exact_vs_oracle": bool(same), "words_per_frame": int(len(ids)), "nodes_per_frame": int(fv.struct.num_nodes)}


def live(args):
    """The live path: one frame at a time through the host-buffer entry points (image up, keypoints and
    descriptors down, synchronous), as the reference's tracking thread calls extract()."""
    w, h = 1241, 376
    frames = synth.make_sequence(w, h, 8, seed=5, dx=3, dy=1)
    ext = orb_extractor(orb_params(max_num_keypts=2000))
    for f in frames[:3]:
        ext.extract(f)
    t0 = time.perf_counter()
    n = 0
    for _ in range(args.steps * 4):
        for f in frames:
            ext.extract(f)
            n += 1
    mono_ms = (time.perf_counter() - t0) / n * 1e3
    # projection::match_current_and_last_frames of two consecutive frames, host buffers in and out
    k0, d0 = ext.extract(frames[0])
    k1, d1 = ext.extract(frames[1])
    view = A.make_frame_view(k1, d1, ext.get_scale_factors(), A.make_grid(0.0, 0.0, float(w), float(h)))
    xy = np.stack([k0["x"] - 3.0, k0["y"] - 1.0], axis=1).astype(np.float32)
    q = A.make_proj_queries(xy, k0["octave"], d0, angle=k0["angle"])
    matcher = match.projection(0.9, True)
    for _ in range(3):
        matcher.match_current_and_last_frames(view, q, 20.0)
    t0 = time.perf_counter()
    for _ in range(args.steps * 10):
        num_matches, _m = matcher.match_current_and_last_frames(view, q, 20.0)
    match_ms = (time.perf_counter() - t0) / (args.steps * 10) * 1e3
    sw, sh = 752, 480
    left, right = synth.make_stereo_pair(sw, sh, seed=100)[:2]
    el, er = orb_extractor(orb_params(max_num_keypts=1000)), orb_extractor(orb_params(max_num_keypts=1000))
    L = _lib.lib()
    fxb, fx = 47.90639384423901, 435.2046959714599

    def stereo_frame():
        kl, dl = el.extract(left)
        kr, dr = er.extract(right)
        xr = np.empty(len(kl), np.float32)
        dep = np.empty(len(kl), np.float32)
        _lib.check(L.ovs_stereo_compute_resident(el._h, er._h, A.ptr(kl, A.kpp), len(kl), A.ptr(kr, A.kpp), len(kr), A.ptr(dl, A.u8p), A.ptr(dr, A.u8p),
                                                 np.float32(fxb), np.float32(fxb / fx), A.ptr(xr, A.f32p), A.ptr(dep, A.f32p), 0))
    for _ in range(3):
        stereo_frame()
    t0 = time.perf_counter()
    for _ in range(args.steps * 10):
        stereo_frame()
    stereo_ms = (time.perf_counter() - t0) / (args.steps * 10) * 1e3

    # the same with the right image extracted on a second host thread, as data/frame.cc:64-67 does it
    import queue
    import threading
    jobs, done = queue.Queue(), queue.Queue()

    def right_worker():
        while True:
            if jobs.get() is None:
                return
            done.put(er.extract(right))
    worker = threading.Thread(target=right_worker, daemon=True)
    worker.start()

    def stereo_frame_two_threads():
        jobs.put(1)
        kl, dl = el.extract(left)
        kr, dr = done.get()
        xr = np.empty(len(kl), np.float32)
        dep = np.empty(len(kl), np.float32)
        _lib.check(L.ovs_stereo_compute_resident(el._h, er._h, A.ptr(kl, A.kpp), len(kl), A.ptr(kr, A.kpp), len(kr), A.ptr(dl, A.u8p), A.ptr(dr, A.u8p),
                                                 np.float32(fxb), np.float32(fxb / fx), A.ptr(xr, A.f32p), A.ptr(dep, A.f32p), 0))
    for _ in range(3):
        stereo_frame_two_threads()
    t0 = time.perf_counter()
    for _ in range(args.steps * 10):
        stereo_frame_two_threads()
    stereo2_ms = (time.perf_counter() - t0) / (args.steps * 10) * 1e3
    jobs.put(None)
    worker.join()
    return {"metric": "live single-frame latency, host buffers in and out", "value": 1e3 / mono_ms, "unit": "frames/s",
            "mono_1241x376_2000kp_ms_per_frame": mono_ms, "stereo_752x480_1000kp_ms_per_pair": stereo_ms, "stereo_pair_right_image_on_a_second_host_thread_ms": stereo2_ms,
            "projection_match_ms": match_ms, "projection_queries": int(len(k0)), "projection_matches": int(num_matches)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", choices=["stereo", "equirect", "knn", "pre", "post", "live", "bow", "all"])
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()
    for name in (["stereo", "equirect", "knn", "pre", "post", "live", "bow"] if args.workload == "all" else [args.workload]):
        print(json.dumps(dict(workload=name, **globals()[name](args))))


if __name__ == "__main__":
    main()

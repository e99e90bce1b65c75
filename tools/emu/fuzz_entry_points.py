#!/usr/bin/env python3
"""Hostile inputs for the host-buffer entry points of the library built for the host (tools/emu): keypoints and
reprojections outside the image, NaN / infinite / huge coordinates and margins, octaves and scale levels outside the
pyramid, index lists that point past their arrays, empty and tiny inputs.  Every array the library receives or
allocates as device memory sits in guarded address space, so an access through an unchecked index ends the process
with the emulator's report; the seed is printed before each call.  Return codes are not judged -- an entry point may
refuse an input (OVS_ERR_*) or compute on it; what it must not do is touch memory that is not its own (on the device
that is a GPU memory fault).

    fuzz_entry_points.py [--seeds A:B] [--only NAME[,NAME...]]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]

import host_lib  # noqa: E402

L = host_lib.install()

from openvslam_amd import _abi as A  # noqa: E402

SF = np.array([1.2 ** i for i in range(8)], dtype=np.float32)
ISF = (1.0 / SF).astype(np.float32)
W, H = 1241.0, 376.0
WILD_F = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 3.4e38, -3.4e38, 1e9, -1e9, 2.0 ** 31, -2.0 ** 31, 2.0 ** 24], dtype=np.float32)
WILD_I = np.array([-1, -2, 8, 9, 15, 16, 63, 64, 100, 1 << 20, (1 << 31) - 1, -(1 << 31)], dtype=np.int64)


def wild_floats(rng, base, frac):
    out = np.asarray(base, dtype=np.float32).copy()
    m = rng.random(out.shape) < frac
    out[m] = rng.choice(WILD_F, int(m.sum()))
    return out


def wild_ints(rng, base, frac):
    out = np.asarray(base, dtype=np.int64).copy()
    m = rng.random(out.shape) < frac
    out[m] = rng.choice(WILD_I, int(m.sum()))
    return out.astype(np.int32)


def frame(rng, n, frac, stereo=None):
    k = np.zeros(n, dtype=A.KEYPOINT_DTYPE)
    spread = rng.choice([1.0, 1.0, 3.0])  # some frames have many keypoints outside the image
    k["x"] = wild_floats(rng, rng.uniform(-(spread - 1) * W, spread * W, n), frac)
    k["y"] = wild_floats(rng, rng.uniform(-(spread - 1) * H, spread * H, n), frac)
    k["octave"] = wild_ints(rng, rng.integers(0, 8, n), frac)
    k["angle"] = wild_floats(rng, rng.uniform(0, 360, n), frac)
    k["size"] = 31.0
    d = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    if rng.random() < 0.3 and n:
        d[:] = d[0]  # everything equally distant: long ties
    sxr = None
    if stereo if stereo is not None else rng.random() < 0.5:
        sxr = wild_floats(rng, np.where(rng.random(n) < 0.6, k["x"] - rng.uniform(1, 40, n), -1.0), frac)
    return k, d, sxr


def grid_for(rng):
    r = rng.random()
    if r < 0.7:
        return A.make_grid(0.0, 0.0, W, H)
    if r < 0.85:
        return A.make_grid(-50.0, -30.0, W + 50, H + 30, 96, 48)
    return A.make_grid(0.0, 0.0, W, H, int(rng.integers(1, 128)), int(rng.integers(1, 128)))


def queries(rng, k, d, nq, frac):
    n = len(k)
    src = rng.integers(0, max(n, 1), nq)
    if n:
        xy = np.stack([k["x"][src], k["y"][src]], axis=1) + rng.normal(0, 5, (nq, 2))
        qd = d[src].copy()
        sl = k["octave"][src].astype(np.int64) + rng.integers(-1, 2, nq)
    else:
        xy = rng.uniform(0, 400, (nq, 2))
        qd = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
        sl = rng.integers(0, 8, nq)
    xy = wild_floats(rng, np.nan_to_num(xy, nan=10.0, posinf=1e6, neginf=-1e6), frac)
    sl = wild_ints(rng, np.clip(sl, -3, 12), frac)
    ang = wild_floats(rng, rng.uniform(0, 360, nq), frac)
    valid = (rng.random(nq) < 0.9).astype(np.uint8)
    xr = wild_floats(rng, xy[:, 0] - 10, frac) if rng.random() < 0.5 else None
    has_obs = (rng.random(nq) < 0.7).astype(np.uint8) if rng.random() < 0.7 else None
    return A.make_proj_queries(xy, sl, qd, valid, xr, ang, has_obs), xy


def margin(rng):
    return float(rng.choice([5.0, 20.0, 100.0, 1e4, 1e9, 0.0, -5.0, np.nan, np.inf, 3e38]))


def sizes(rng):
    n = int(rng.choice([0, 1, 2, 63, 64, 65, 300, 1500]))
    nq = int(rng.choice([0, 1, 2, 63, 64, 65, 300, 1200]))
    return n, nq


def out_i32(n):
    return np.full(max(n, 1), -7, dtype=np.int32)


def fuzz_projection(rng):
    n, nq = sizes(rng)
    frac = float(rng.choice([0.0, 0.02, 0.3]))
    k, d, sxr = frame(rng, n, frac)
    view = A.make_frame_view(k, d, SF, grid_for(rng), sxr)
    q, _ = queries(rng, k, d, nq, frac)
    occ = (rng.random(n) < 0.2).astype(np.uint8) if rng.random() < 0.5 else None
    m, num = out_i32(n), C.c_uint32(0)
    which = int(rng.integers(0, 5))
    if which == 0:
        return L.ovs_match_current_and_last_frames(view.ref(), q.ref(), A.ptr(occ, A.u8p), margin(rng), int(rng.integers(0, 3)), int(rng.choice([8, 0, -1, 100])),
                                                   0.9, int(rng.integers(0, 2)), A.ptr(m, A.i32p), C.byref(num), 0)
    if which == 1:
        return L.ovs_match_frame_and_landmarks(view.ref(), q.ref(), A.ptr(occ, A.u8p), margin(rng), float(rng.choice([0.8, 0.6, 1.0, np.nan])),
                                               A.ptr(m, A.i32p), C.byref(num), 0)
    if which == 2:
        return L.ovs_match_frame_and_keyframe(view.ref(), q.ref(), A.ptr(occ, A.u8p), margin(rng), int(rng.choice([50, 100, 0, 256, 4000000000])),
                                              int(rng.integers(0, 2)), A.ptr(m, A.i32p), C.byref(num), 0)
    if which == 3:
        return L.ovs_match_by_sim3_transform(view.ref(), q.ref(), A.ptr(occ, A.u8p), margin(rng), A.ptr(m, A.i32p), C.byref(num), 0)
    b = out_i32(nq)
    return L.ovs_match_keyframe_one_way(view.ref(), q.ref(), margin(rng), A.ptr(b, A.i32p), 0)


def fuzz_fuse(rng):
    n, nq = sizes(rng)
    frac = float(rng.choice([0.0, 0.02, 0.3]))
    k, d, sxr = frame(rng, n, frac, stereo=True)
    view = A.make_frame_view(k, d, SF, grid_for(rng), sxr)
    q, xy = queries(rng, k, d, nq, frac)
    best = out_i32(nq)
    replace = int(rng.integers(0, 2))
    rd = xy.astype(np.float64) if replace else None
    isig = (1.0 / (SF * SF)).astype(np.float32) if replace else None
    return L.ovs_fuse_best_candidates(view.ref(), q.ref(), A.ptr(rd, A.f64p), A.ptr(isig, A.f32p), margin(rng), replace, A.ptr(best, A.i32p), 0)


def fuzz_area(rng):
    n2, n1 = sizes(rng)
    frac = float(rng.choice([0.0, 0.02, 0.3]))
    k2, d2, _ = frame(rng, n2, frac, stereo=False)
    k1, d1, _ = frame(rng, n1, frac, stereo=False)
    view = A.make_frame_view(k2, d2, SF, grid_for(rng))
    prev = wild_floats(rng, rng.uniform(0, W, (max(n1, 1), 2)), frac)
    m, num = out_i32(n1), C.c_uint32(0)
    return L.ovs_match_in_consistent_area(A.ptr(k1, A.kpp), A.ptr(d1, A.u8p), n1, view.ref(), A.ptr(prev, A.f32p), int(rng.choice([30, 100, 0, -1, 100000, 2 ** 31 - 1])),
                                          float(rng.choice([0.9, 1.0, 0.0, np.nan])), int(rng.integers(0, 2)), A.ptr(m, A.i32p), C.byref(num), 0)


def fuzz_cells(rng):
    n, nq = sizes(rng)
    frac = float(rng.choice([0.0, 0.05, 0.3]))
    k, d, _ = frame(rng, n, frac, stereo=False)
    view = A.make_frame_view(k, d, SF, grid_for(rng))
    ref = wild_floats(rng, rng.uniform(-100, W + 100, (max(nq, 1), 2)), frac)
    mar = wild_floats(rng, rng.uniform(0, 200, max(nq, 1)), frac)
    lo = wild_ints(rng, rng.integers(-1, 8, max(nq, 1)), frac)
    hi = wild_ints(rng, rng.integers(-1, 8, max(nq, 1)), frac)
    cap = int(rng.choice([0, 1, 100, 100000]))
    out = np.zeros(max(cap, 1), np.uint32)
    offs = np.zeros(nq + 1, np.uint32)
    return L.ovs_get_keypoints_in_cell(view.ref(), nq, A.ptr(ref, A.f32p), A.ptr(mar, A.f32p), A.ptr(lo, A.i32p), A.ptr(hi, A.i32p), A.ptr(out, A.u32p), cap,
                                       A.ptr(offs, A.u32p), 0)


def bad_feat_vec(rng, n, frac):
    """A feature vector whose lists may name keypoints past the frame; offsets stay monotone (the caller's
    container guarantees that much: they are the sizes of its vectors)."""
    nodes = int(rng.choice([0, 1, 5, 60]))
    ids = np.sort(rng.choice(1000, nodes, replace=False)).astype(np.uint32)
    counts = rng.integers(0, 2 * max(n, 1) // max(nodes, 1) + 2, nodes)
    offsets = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint32)
    total = int(offsets[-1])
    idx = rng.integers(0, max(n, 1), total).astype(np.int64)
    m = rng.random(total) < frac
    idx[m] = rng.choice([n, n + 1, 2 * n + 5, 1 << 20, (1 << 32) - 1], int(m.sum()))
    idx = idx.astype(np.uint32)
    v = A.BowFeatVec()
    v.num_nodes = nodes
    v.node_ids, v.offsets, v.indices = A.ptr(ids, A.u32p), A.ptr(offsets, A.u32p), A.ptr(idx if total else np.zeros(1, np.uint32), A.u32p)
    return A.Holder(v, ids, offsets, idx)


def fuzz_robust(rng):
    n1, n2 = sizes(rng)
    frac = float(rng.choice([0.0, 0.02, 0.2]))
    d1 = rng.integers(0, 256, (max(n1, 1), 32), dtype=np.uint8)
    d2 = rng.integers(0, 256, (max(n2, 1), 32), dtype=np.uint8)
    if rng.random() < 0.3:
        d2[:] = d1[0]
    a1 = wild_floats(rng, rng.uniform(0, 360, max(n1, 1)), frac)
    a2 = wild_floats(rng, rng.uniform(0, 360, max(n2, 1)), frac)
    v1 = (rng.random(max(n1, 1)) < 0.8).astype(np.uint8)
    v2 = (rng.random(max(n2, 1)) < 0.8).astype(np.uint8)
    lowe = float(rng.choice([0.75, 0.9, 1.0, 0.0, np.nan]))
    check = int(rng.integers(0, 2))
    num = C.c_uint32(0)
    which = int(rng.integers(0, 3))
    if which == 0:
        m = out_i32(n1)
        return L.ovs_brute_force_match(A.ptr(d1, A.u8p), A.ptr(a1, A.f32p), n1, A.ptr(d2, A.u8p), A.ptr(a2, A.f32p), A.ptr(v2, A.u8p), n2, lowe, check,
                                       A.ptr(m, A.i32p), C.byref(num), 0)
    f1, f2 = bad_feat_vec(rng, n1, frac), bad_feat_vec(rng, n2, frac)
    if which == 1:
        m = out_i32(n2)
        return L.ovs_bow_match_frame_and_keyframe(f1.ref(), A.ptr(d1, A.u8p), A.ptr(a1, A.f32p), A.ptr(v1, A.u8p), n1, f2.ref(), A.ptr(d2, A.u8p),
                                                  A.ptr(a2, A.f32p), n2, lowe, check, A.ptr(m, A.i32p), C.byref(num), 0)
    m = out_i32(n1)
    return L.ovs_bow_match_keyframes(f1.ref(), A.ptr(d1, A.u8p), A.ptr(a1, A.f32p), A.ptr(v1, A.u8p), n1, f2.ref(), A.ptr(d2, A.u8p), A.ptr(a2, A.f32p),
                                     A.ptr(v2, A.u8p), n2, lowe, check, A.ptr(m, A.i32p), C.byref(num), 0)


def fuzz_landmarks(rng):
    nl = int(rng.choice([0, 1, 7, 200]))
    sizes_ = rng.integers(0, int(rng.choice([3, 40, 600])), nl)
    offsets = np.concatenate([[0], np.cumsum(sizes_)]).astype(np.uint32)
    descs = rng.integers(0, 256, (max(int(offsets[-1]), 1), 32), dtype=np.uint8)
    best = out_i32(nl)
    return L.ovs_landmark_compute_descriptors(A.ptr(descs, A.u8p), A.ptr(offsets, A.u32p), nl, A.ptr(best, A.i32p), 0)


def image_view(img):
    v = A.ImageView()
    v.data = A.ptr(img, A.u8p)
    v.width, v.height, v.stride = img.shape[1], img.shape[0], img.strides[0]
    return v


def fuzz_stereo(rng):
    w, h = int(rng.choice([64, 200, 752])), int(rng.choice([48, 160, 480]))
    levels = int(rng.choice([1, 3, 8]))
    pyr_l, pyr_r, keep = (A.ImageView * levels)(), (A.ImageView * levels)(), []
    for lv in range(levels):
        lw, lh = max(int(round(w * ISF[lv])), 1), max(int(round(h * ISF[lv])), 1)
        a = rng.integers(0, 256, (lh, lw), dtype=np.uint8)
        b = rng.integers(0, 256, (lh, lw), dtype=np.uint8)
        keep += [a, b]
        pyr_l[lv], pyr_r[lv] = image_view(a), image_view(b)
    frac = float(rng.choice([0.0, 0.05, 0.3]))
    nl, nr = int(rng.choice([0, 1, 64, 400])), int(rng.choice([0, 1, 64, 400]))

    def kps(n):
        k = np.zeros(max(n, 1), dtype=A.KEYPOINT_DTYPE)
        k["x"] = wild_floats(rng, rng.uniform(-20, w + 20, max(n, 1)), frac)
        k["y"] = wild_floats(rng, rng.uniform(-20, h + 20, max(n, 1)), frac)
        k["octave"] = wild_ints(rng, rng.integers(0, levels, max(n, 1)), frac)
        return k
    kl, kr = kps(nl), kps(nr)
    dl = rng.integers(0, 256, (max(nl, 1), 32), dtype=np.uint8)
    dr = rng.integers(0, 256, (max(nr, 1), 32), dtype=np.uint8)
    sxr, depth = np.zeros(max(nl, 1), np.float32), np.zeros(max(nl, 1), np.float32)
    return L.ovs_stereo_compute(pyr_l, pyr_r, levels, A.ptr(kl, A.kpp), nl, A.ptr(kr, A.kpp), nr, A.ptr(dl, A.u8p), A.ptr(dr, A.u8p), A.ptr(SF, A.f32p),
                                A.ptr(ISF, A.f32p), float(rng.choice([40.0, 0.0, np.nan, 1e9])), float(rng.choice([0.1, 0.0, np.nan])), A.ptr(sxr, A.f32p),
                                A.ptr(depth, A.f32p), 0)


def fuzz_extract(rng):
    w, h = int(rng.choice([38, 40, 64, 100, 253, 320, 400])), int(rng.choice([38, 40, 64, 100, 240, 300]))
    if rng.random() < 0.15:
        w, h = int(rng.choice([1, 31, 37])), int(rng.choice([1, 31, 37]))  # below the 19-pixel borders: refused
    kind = rng.integers(0, 4)
    img = [rng.integers(0, 256, (h, w), dtype=np.uint8), np.full((h, w), 128, np.uint8),
           ((np.indices((h, w)).sum(0) % 2) * 255).astype(np.uint8), rng.choice([0, 255], (h, w)).astype(np.uint8)][kind]
    rects = []
    for _ in range(int(rng.integers(0, 3))):
        if rng.random() < 0.7:
            x0, x1 = sorted(rng.choice([-1.0, 0.0, 0.2, 0.5, 0.9, 1.0, 2.0], 2, replace=False))
            y0, y1 = sorted(rng.choice([-1.0, 0.0, 0.2, 0.5, 0.9, 1.0, 2.0], 2, replace=False))
            rects.append((float(x0), float(x1), float(y0), float(y1)))
        else:
            rects.append(tuple(float(v) for v in rng.choice([-1.0, 0.0, 0.5, 1.0, np.nan, np.inf], 4)))
    scale = float(rng.choice([1.2, 1.01, 1.5, 2.0, 3.0]))
    levels = int(rng.choice([1, 2, 3, 5, 8, 16]))
    if rng.random() < 0.85:
        # mostly pyramids whose smallest level still has an interior (more than 38 pixels per side)
        fit = 1
        while fit < 16 and round(min(w, h) / scale ** fit) > 38:
            fit += 1
        levels = int(rng.integers(1, fit + 1))
    p = A.make_orb_params(int(rng.choice([1, 50, 500, 5000])), scale, levels, int(rng.choice([20, 7, 255, 1])), int(rng.choice([7, 1, 20])), rects)
    hnd = C.c_void_p()
    rc = L.ovs_orb_create(p.ref(), 0, C.byref(hnd))
    if rc != 0:
        return rc
    cap = L.ovs_orb_keypoint_capacity(hnd, w, h)
    if cap < 0:
        L.ovs_orb_destroy(hnd)
        return cap
    kp = np.zeros(max(cap, 1), dtype=A.KEYPOINT_DTYPE)
    desc = np.zeros((max(cap, 1), 32), np.uint8)
    n = C.c_int(0)
    mask = None
    if rng.random() < 0.3:
        mask = (rng.random((h, w)) < 0.7).astype(np.uint8) * 255
    if rng.random() < 0.5:
        L.ovs_orb_set_blur_kernel(hnd, 1)
    rc = L.ovs_orb_extract(hnd, A.ptr(img, A.u8p), w, h, img.strides[0], A.ptr(mask, A.u8p), w if mask is not None else 0, A.ptr(kp, A.kpp), A.ptr(desc, A.u8p),
                           cap, C.byref(n))
    for _ in range(int(rng.integers(0, 3))):
        # more frames on the same handle: the same size, or another one (the plan and the buffers are rebuilt)
        if rc != 0:
            break
        if rng.random() < 0.6:
            w, h = int(rng.choice([64, 100, 253, 320, 400, 640])), int(rng.choice([64, 100, 240, 300, 480]))
        cap2 = L.ovs_orb_keypoint_capacity(hnd, w, h)
        if cap2 <= 0:
            break
        kp2 = np.zeros(cap2, dtype=A.KEYPOINT_DTYPE)
        desc2 = np.zeros((cap2, 32), np.uint8)
        img2 = rng.integers(0, 256, (h, w), dtype=np.uint8)
        rc = L.ovs_orb_extract(hnd, A.ptr(img2, A.u8p), w, h, img2.strides[0], A.ptr(None, A.u8p), 0, A.ptr(kp2, A.kpp), A.ptr(desc2, A.u8p), cap2, C.byref(n))
    L.ovs_orb_destroy(hnd)
    return rc


def fuzz_bow(rng):
    from openvslam_amd import bow
    k, depth = int(rng.choice([2, 3, 10, 37])), int(rng.choice([1, 2, 3]))
    voc = bow.synthetic_vocabulary(k, depth, seed=int(rng.integers(0, 1000)), early_leaf_fraction=float(rng.choice([0.0, 0.3])),
                                   shuffle_ids=bool(rng.integers(0, 2)), stop_fraction=float(rng.choice([0.0, 0.5, 1.0])))
    voc = {key: (v.copy() if isinstance(v, np.ndarray) else v) for key, v in voc.items()}
    nodes = len(voc["child_offsets"]) - 1
    damage = int(rng.integers(0, 7))
    if damage == 1 and len(voc["children"]):
        voc["children"][rng.integers(0, len(voc["children"]))] = rng.choice([0, nodes, nodes + 7, (1 << 32) - 1, 1 << 31])
    elif damage == 2:
        voc["child_offsets"][rng.integers(0, nodes + 1)] = rng.choice([0, len(voc["children"]) + 5, (1 << 32) - 1])
    elif damage == 3:
        voc["weights"][rng.integers(0, nodes)] = rng.choice([np.nan, np.inf, -1.0, 1e308])
    elif damage == 4 and len(voc["children"]) > 1:
        voc["children"][1] = voc["children"][0]
    view = A.make_bow_vocab_view(voc["k"], voc["L"], voc["node_descs"], voc["child_offsets"], voc["children"], voc["word_ids"], voc["weights"],
                                 int(rng.choice([0, 1, 2, 3, 9, -1])) if damage == 5 else int(rng.integers(0, 4)), int(rng.integers(0, 6)))
    h = C.c_void_p()
    rc = L.ovs_bow_vocab_create(view.ref(), 0, C.byref(h))
    if rc != 0:
        return rc
    n = int(rng.choice([0, 1, 15, 16, 17, 500]))
    d = rng.integers(0, 256, (max(n, 1), 32), dtype=np.uint8)
    word_of, node_of = np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.uint32)
    ids, vals = np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.float64)
    nids, offs, idx = np.zeros(n + 1, np.uint32), np.zeros(n + 2, np.uint32), np.zeros(n + 1, np.uint32)
    nw, nn = C.c_int32(0), C.c_int32(0)
    rc = L.ovs_bow_transform(h, A.ptr(d, A.u8p), n, int(rng.choice([0, 1, 4, 100, -1])), A.ptr(word_of, A.u32p), A.ptr(node_of, A.u32p), A.ptr(ids, A.u32p),
                             A.ptr(vals, A.f64p), C.byref(nw), A.ptr(nids, A.u32p), A.ptr(offs, A.u32p), A.ptr(idx, A.u32p), C.byref(nn))
    L.ovs_bow_vocab_destroy(h)
    return rc


def fuzz_triangulation(rng):
    n1, n2 = sizes(rng)
    frac = float(rng.choice([0.0, 0.02, 0.2]))
    k1, d1, s1 = frame(rng, max(n1, 1), frac)
    k2, d2, s2 = frame(rng, max(n2, 1), frac)

    def bearings(n):
        b = rng.normal(0, 1, (max(n, 1), 3))
        b /= np.linalg.norm(b, axis=1, keepdims=True)
        b = b.astype(np.float64)
        m = rng.random(b.shape) < frac
        b[m] = rng.choice([np.nan, np.inf, 0.0, 1e300], int(m.sum()))
        return b
    b1, b2 = bearings(n1), bearings(n2)
    l1 = (rng.random(max(n1, 1)) < 0.3).astype(np.uint8)
    l2 = (rng.random(max(n2, 1)) < 0.3).astype(np.uint8)
    f1, f2 = bad_feat_vec(rng, n1, frac), bad_feat_vec(rng, n2, frac)
    E = rng.normal(0, 1, 9)
    if rng.random() < 0.2:
        E[:] = rng.choice([0.0, np.nan, np.inf])
    ep = rng.normal(0, 1, 3)
    levels = int(rng.choice([8, 1, 0, 64]))
    pairs = np.zeros((max(n1, 1), 2), np.uint32)
    num = C.c_uint32(0)
    return L.ovs_match_for_triangulation(f1.ref(), A.ptr(d1, A.u8p), A.ptr(k1, A.kpp), A.ptr(b1, A.f64p), A.ptr(s1, A.f32p), A.ptr(l1, A.u8p), n1,
                                         f2.ref(), A.ptr(d2, A.u8p), A.ptr(k2, A.kpp), A.ptr(b2, A.f64p), A.ptr(s2, A.f32p), A.ptr(l2, A.u8p), n2,
                                         A.ptr(E, A.f64p), A.ptr(ep, A.f64p), A.ptr(SF, A.f32p), levels, int(rng.integers(0, 2)), A.ptr(pairs, A.u32p),
                                         C.byref(num), 0)


L._cdll.ovs_emu_guarded_alloc.restype = C.c_void_p
L._cdll.ovs_emu_guarded_alloc.argtypes = [C.c_size_t, C.c_int]
L._cdll.ovs_emu_guarded_free.argtypes = [C.c_void_p]


class Guarded:
    """Copies of numpy arrays in guarded blocks (what hipMalloc returns on this build): `dev(a)` gives an array
    of the same shape and dtype whose memory ends at inaccessible address space; all are freed by close()."""

    def __init__(self):
        self.ptrs = []

    def dev(self, a):
        a = np.ascontiguousarray(a)
        nbytes = max(a.nbytes, 1)
        p = L._cdll.ovs_emu_guarded_alloc(nbytes, 0xDB)
        assert p, "cannot reserve a guarded block"
        self.ptrs.append(p)
        buf = (C.c_char * nbytes).from_address(p)
        out = np.frombuffer(buf, dtype=a.dtype, count=a.size).reshape(a.shape)
        out[...] = a
        return out

    def close(self):
        for p in self.ptrs:
            L._cdll.ovs_emu_guarded_free(p)
        self.ptrs = []


def wild_counts(rng, n_frames, cap, frac):
    c = rng.integers(0, cap + 1, n_frames).astype(np.int64)
    m = rng.random(n_frames) < frac
    c[m] = rng.choice([-1, -1000, cap + 1, 2 * cap, 1 << 30, -(1 << 31)], int(m.sum()))
    return c.astype(np.int32)


def fuzz_batch_match_dev(rng):
    """ovs_match_batch_current_and_last_dev on arrays that did not come from the extractor: counts outside [0, cap],
    keypoints and reprojections anywhere.  (Device pointers are host pointers on this build.)"""
    n_frames, cap = int(rng.choice([1, 2, 3, 9])), int(rng.choice([1, 64, 300, 1100]))
    frac = float(rng.choice([0.0, 0.05, 0.3]))
    k, d, _ = frame(rng, n_frames * cap, frac, stereo=False)
    counts = wild_counts(rng, n_frames, cap, frac)
    reproj = wild_floats(rng, rng.uniform(-50, W + 50, (n_frames, cap, 2)), frac)
    reproj[rng.random((n_frames, cap)) < 0.1, 0] = -3e38
    G = Guarded()
    k, d, counts, reproj = G.dev(k), G.dev(d), G.dev(counts), G.dev(reproj)
    matched = G.dev(np.full((n_frames, cap), -7, np.int32))
    num = G.dev(np.zeros(n_frames, np.uint32))
    g = grid_for(rng)
    levels = int(rng.choice([8, 8, 1, 3]))
    rc = L.ovs_match_batch_current_and_last_dev(k.ctypes.data, d.ctypes.data, counts.ctypes.data, n_frames, cap, reproj.ctypes.data, C.byref(g),
                                                A.ptr(SF, A.f32p), levels, margin(rng), int(rng.integers(0, 2)), matched.ctypes.data, num.ctypes.data, 0, None)
    G.close()
    return rc


def fuzz_pack_dev(rng):
    n_frames, cap = int(rng.choice([1, 2, 17, 300])), int(rng.choice([1, 7, 64, 500]))
    counts = wild_counts(rng, n_frames, cap, float(rng.choice([0.0, 0.3])))
    kps = np.zeros(n_frames * cap, dtype=A.KEYPOINT_DTYPE)
    desc = rng.integers(0, 256, (n_frames * cap, 32), dtype=np.uint8)
    matched = rng.integers(-1, cap, n_frames * cap).astype(np.int32)
    G = Guarded()
    counts, kps, desc, matched = G.dev(counts), G.dev(kps), G.dev(desc), G.dev(matched)
    offs = G.dev(np.zeros(n_frames + 1, np.int32))
    pk, pd, pm = G.dev(np.zeros_like(kps)), G.dev(np.zeros_like(desc)), G.dev(np.zeros_like(matched))
    which = int(rng.integers(0, 4))
    rc = L.ovs_pack_batch_results_dev(counts.ctypes.data, n_frames, cap, kps.ctypes.data if which != 1 else None, desc.ctypes.data if which != 2 else None,
                                      matched.ctypes.data if which != 3 else None, offs.ctypes.data, pk.ctypes.data, pd.ctypes.data, pm.ctypes.data, 0, None)
    G.close()
    return rc


def fuzz_extract_stereo_dev(rng):
    """Two batches through ovs_orb_extract_batch_dev, then ovs_stereo_compute_dev on keypoint arrays that were
    overwritten with hostile values, and the undistortion / bearing entry points on the same arrays."""
    w, h = int(rng.choice([64, 160, 320])), int(rng.choice([48, 120, 240]))
    n_frames = int(rng.choice([1, 2, 3]))
    levels = int(rng.choice([1, 2, 3]))
    p = A.make_orb_params(int(rng.choice([50, 400])), 1.2, levels)
    hl, hr = C.c_void_p(), C.c_void_p()
    if L.ovs_orb_create(p.ref(), 0, C.byref(hl)) != 0 or L.ovs_orb_create(p.ref(), 0, C.byref(hr)) != 0:
        return -1
    cap = L.ovs_orb_keypoint_capacity(hl, w, h)
    rc = cap if cap < 0 else 0
    G = Guarded()
    if cap > 0:
        imgs = [G.dev(rng.integers(0, 256, (n_frames, h, w), dtype=np.uint8)) for _ in range(2)]
        out = []
        for hnd, im in zip((hl, hr), imgs):
            kps = G.dev(np.zeros(n_frames * cap, dtype=A.KEYPOINT_DTYPE))
            desc = G.dev(np.zeros((n_frames * cap, 32), np.uint8))
            counts = G.dev(np.zeros(n_frames, np.int32))
            rc = L.ovs_orb_extract_batch_dev(hnd, im.ctypes.data, n_frames, w, h, w, w * h, None, 0, kps.ctypes.data, desc.ctypes.data, counts.ctypes.data, cap, None)
            out.append((kps, desc, counts))
        if rc == 0:
            frac = float(rng.choice([0.0, 0.05, 0.3]))
            for kps, desc, counts in out:
                m = rng.random(len(kps)) < frac
                kps["x"][m] = rng.choice(WILD_F, int(m.sum()))
                kps["y"][m] = rng.choice(WILD_F, int(m.sum()))
                kps["octave"][m] = rng.choice(WILD_I, int(m.sum())).astype(np.int32)
                counts[:] = np.where(rng.random(n_frames) < frac, wild_counts(rng, n_frames, cap, 1.0), counts)
            (kl, dl, nl), (kr, dr, nr) = out
            xr, dep = G.dev(np.zeros(cap, np.float32)), G.dev(np.zeros(cap, np.float32))
            status = G.dev(np.zeros(1, np.int32))
            f = int(rng.integers(0, n_frames))
            rc = L.ovs_stereo_compute_dev(hl, f, hr, f, kl[f * cap:].ctypes.data, dl[f * cap:].ctypes.data, nl[f:].ctypes.data, kr[f * cap:].ctypes.data,
                                          dr[f * cap:].ctypes.data, nr[f:].ctypes.data, cap, 40.0, 0.1, xr.ctypes.data, dep.ctypes.data, status.ctypes.data, 0, None)
            cam = A.Camera()
            cam.model, cam.cols, cam.rows = int(rng.integers(0, 3)), w, h
            cam.fx, cam.fy, cam.cx, cam.cy = 300.0, 300.0, w / 2.0, h / 2.0
            for i, v in enumerate((-0.28, 0.07, 0.0002, 0.00002, 0.0)):
                cam.dist[i] = v
            und = G.dev(np.zeros_like(kl))
            bear = G.dev(np.zeros((n_frames * cap, 3), np.float64))
            L.ovs_undistort_keypoints_dev(C.byref(cam), kl.ctypes.data, nl.ctypes.data, n_frames, cap, und.ctypes.data, bear.ctypes.data, 0, None)
            L.ovs_convert_keypoints_to_bearings_dev(C.byref(cam), kl.ctypes.data, nl.ctypes.data, n_frames, cap, bear.ctypes.data, 0, None)
    L.ovs_orb_destroy(hl)
    L.ovs_orb_destroy(hr)
    G.close()
    return rc


def fuzz_database_dev(rng):
    rows = int(rng.choice([0, 1, 63, 64, 65, 4096, 4097, 9000]))
    nq = int(rng.choice([0, 1, 63, 64, 65, 300]))
    G = Guarded()
    db = G.dev(rng.integers(0, int(rng.choice([2, 256])), (max(rows, 1), 32), dtype=np.uint8))
    q = G.dev(rng.integers(0, int(rng.choice([2, 256])), (max(nq, 1), 32), dtype=np.uint8))
    h = C.c_void_p()
    if L.ovs_desc_db_create(0, C.byref(h)) != 0:
        return -1
    first = int(rng.choice([0, 5, 1 << 40]))
    rc = L.ovs_desc_db_attach_dev(h, db.ctypes.data, rows, first)
    out = G.dev(np.zeros(max(nq, 1), dtype=A.KNN2_DTYPE))
    if rc == 0:
        fn = L.ovs_desc_db_search_noalloc_dev if rng.random() < 0.5 else L.ovs_desc_db_search_dev
        rc = fn(h, q.ctypes.data, nq, out.ctypes.data, None)
        parts = int(rng.choice([1, 2, 5]))
        gathered = np.zeros(parts * max(nq, 1), dtype=A.KNN2_DTYPE)
        gathered["dist"] = rng.integers(0, 257, gathered["dist"].shape)
        gathered["row"] = rng.integers(-1, 1 << 40, gathered["row"].shape)
        gathered, merged = G.dev(gathered), G.dev(np.zeros(max(nq, 1), dtype=A.KNN2_DTYPE))
        L.ovs_knn2_merge_dev(gathered.ctypes.data, parts, nq, merged.ctypes.data, None)
    L.ovs_desc_db_destroy(h)
    G.close()
    return rc


TARGETS = {"batch_match_dev": fuzz_batch_match_dev, "pack_dev": fuzz_pack_dev, "extract_stereo_dev": fuzz_extract_stereo_dev, "database_dev": fuzz_database_dev,
           "triangulation": fuzz_triangulation, "bow": fuzz_bow, "projection": fuzz_projection, "fuse": fuzz_fuse, "area": fuzz_area, "cells": fuzz_cells, "robust": fuzz_robust, "landmarks": fuzz_landmarks,
           "stereo": fuzz_stereo, "extract": fuzz_extract}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="0:200")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    lo, hi = (int(v) for v in a.seeds.split(":"))
    names = [n for n in (a.only.split(",") if a.only else TARGETS)]
    codes = {}
    for seed in range(lo, hi):
        for name in names:
            print("seed %d %s" % (seed, name), flush=True)
            rng = np.random.default_rng([seed, sorted(TARGETS).index(name)])
            rc = TARGETS[name](rng)
            codes[(name, rc)] = codes.get((name, rc), 0) + 1
    for (name, rc), c in sorted(codes.items()):
        print("%-12s rc %3d x %d" % (name, rc, c))
    print("OK")


if __name__ == "__main__":
    sys.exit(main())

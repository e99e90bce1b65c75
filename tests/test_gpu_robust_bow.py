"""GPU parity of robust::brute_force_match and the bow_tree matchers against the CPU oracle."""
import ctypes as C

import numpy as np
import pytest

import oracle
from openvslam_amd import _abi as A
from openvslam_amd import _lib

pytestmark = pytest.mark.gpu


def noisy_copies(rng, desc, src, max_flips):
    out = desc[src].copy()
    for i in range(len(src)):
        for b in rng.choice(256, rng.integers(0, max_flips + 1), replace=False):
            out[i, b >> 3] ^= np.uint8(1 << (b & 7))
    return out


def two_sets(rng, n1, n2, shared=0.6, max_flips=24, dup=0.15):
    """Two descriptor sets with a shared part (noisy copies) and near-duplicates inside set 1 so that
    the ratio test and the greedy exclusion both matter."""
    d1 = rng.integers(0, 256, (n1, 32), dtype=np.uint8)
    ndup = int(dup * n1)
    if ndup and n1 > 1:
        src = rng.integers(0, n1, ndup)
        d1[rng.integers(0, n1, ndup)] = noisy_copies(rng, d1, src, 10)
    d2 = rng.integers(0, 256, (n2, 32), dtype=np.uint8)
    k = int(shared * min(n1, n2))
    if k:
        idx2 = rng.choice(n2, k, replace=False)
        d2[idx2] = noisy_copies(rng, d1, rng.integers(0, n1, k), max_flips)
    a1 = rng.uniform(0, 360, n1).astype(np.float32)
    a2 = rng.uniform(0, 360, n2).astype(np.float32)
    return d1, a1, d2, a2


@pytest.mark.parametrize("seed,n1,n2,lowe,check", [(1, 2000, 2000, 0.75, 1), (2, 1500, 2200, 0.9, 0), (3, 64, 3000, 0.75, 1),
                                                   (4, 3000, 40, 0.6, 1), (5, 0, 50, 0.75, 1), (6, 50, 0, 0.75, 1)])
def test_brute_force_match(seed, n1, n2, lowe, check):
    rng = np.random.default_rng(seed)
    d1, a1, d2, a2 = two_sets(rng, n1, n2)
    valid2 = (rng.random(n2) < 0.85).astype(np.uint8)
    res = []
    for lib_, prefix, extra in ((_lib.lib(), "ovs_", (0,)), (oracle.lib(), "ovo_", ())):
        m = np.full((max(n1, 1), 2), -5, dtype=np.int32)
        num = C.c_uint32(999)
        rc = getattr(lib_, prefix + "brute_force_match")(A.ptr(d1, A.u8p), A.ptr(a1, A.f32p), n1, A.ptr(d2, A.u8p), A.ptr(a2, A.f32p),
                                                         A.ptr(valid2, A.u8p), n2, lowe, check, A.ptr(m, A.i32p), C.byref(num), *extra)
        assert rc == 0
        res.append((m[:num.value].copy(), num.value))
    assert res[0][1] == res[1][1]
    assert np.array_equal(res[0][0], res[1][0])
    if n1 >= 1500 and n2 >= 1500:
        assert res[0][1] > 100


def random_feat_vec(rng, n, n_nodes, node_id_space):
    ids = np.sort(rng.choice(node_id_space, n_nodes, replace=False))
    assign = rng.integers(0, n_nodes, n)
    perm = rng.permutation(n)
    d = {}
    for i in perm:
        d.setdefault(int(ids[assign[i]]), []).append(int(i))
    return A.make_bow_feat_vec(d)


@pytest.mark.parametrize("seed,n_kf,n_frm,nodes,check", [(11, 2000, 2000, 400, 1), (12, 1200, 2500, 150, 0), (13, 300, 300, 10, 1)])
def test_bow_match_frame_and_keyframe(seed, n_kf, n_frm, nodes, check):
    rng = np.random.default_rng(seed)
    dk, ak, df, af = two_sets(rng, n_kf, n_frm)
    vk = random_feat_vec(rng, n_kf, nodes, 3 * nodes)
    vf = random_feat_vec(rng, n_frm, nodes, 3 * nodes)
    # make the node assignment of shared descriptors mostly agree so that buckets contain true matches
    valid = (rng.random(n_kf) < 0.8).astype(np.uint8)
    res = []
    for lib_, prefix, extra in ((_lib.lib(), "ovs_", (0,)), (oracle.lib(), "ovo_", ())):
        m = np.full(n_frm, -5, dtype=np.int32)
        num = C.c_uint32(999)
        rc = getattr(lib_, prefix + "bow_match_frame_and_keyframe")(vk.ref(), A.ptr(dk, A.u8p), A.ptr(ak, A.f32p), A.ptr(valid, A.u8p), n_kf,
                                                                    vf.ref(), A.ptr(df, A.u8p), A.ptr(af, A.f32p), n_frm,
                                                                    0.75, check, A.ptr(m, A.i32p), C.byref(num), *extra)
        assert rc == 0
        res.append((m, num.value))
    assert res[0][1] == res[1][1]
    assert np.array_equal(res[0][0], res[1][0])


def clustered_sets(rng, n1, n2, n_words):
    """Descriptors drawn around `n_words` centres; the feature vector bucket of a keypoint is its centre."""
    centres = rng.integers(0, 256, (n_words, 32), dtype=np.uint8)
    w1, w2 = rng.integers(0, n_words, n1), rng.integers(0, n_words, n2)
    d1 = noisy_copies(rng, centres, w1, 16)
    d2 = noisy_copies(rng, centres, w2, 16)
    v1, v2 = {}, {}
    for i in rng.permutation(n1):
        v1.setdefault(int(w1[i]) * 7 + 3, []).append(int(i))
    for i in rng.permutation(n2):
        v2.setdefault(int(w2[i]) * 7 + 3, []).append(int(i))
    return d1, d2, A.make_bow_feat_vec(v1), A.make_bow_feat_vec(v2)


@pytest.mark.parametrize("seed,n1,n2,words,check", [(21, 2000, 2000, 300, 1), (22, 1000, 2400, 40, 1), (23, 800, 700, 8, 0)])
def test_bow_matchers_on_clustered_buckets(seed, n1, n2, words, check):
    rng = np.random.default_rng(seed)
    d1, d2, v1, v2 = clustered_sets(rng, n1, n2, words)
    a1 = rng.uniform(0, 360, n1).astype(np.float32)
    a2 = ((a1[rng.integers(0, n1, n2)] + rng.normal(0, 10, n2)) % 360).astype(np.float32)
    valid1 = (rng.random(n1) < 0.8).astype(np.uint8)
    valid2 = (rng.random(n2) < 0.8).astype(np.uint8)
    for name in ("frame_kf", "keyframes"):
        res = []
        for lib_, prefix, extra in ((_lib.lib(), "ovs_", (0,)), (oracle.lib(), "ovo_", ())):
            num = C.c_uint32(999)
            if name == "frame_kf":
                m = np.full(n2, -5, dtype=np.int32)
                rc = getattr(lib_, prefix + "bow_match_frame_and_keyframe")(v1.ref(), A.ptr(d1, A.u8p), A.ptr(a1, A.f32p), A.ptr(valid1, A.u8p), n1,
                                                                            v2.ref(), A.ptr(d2, A.u8p), A.ptr(a2, A.f32p), n2,
                                                                            0.75, check, A.ptr(m, A.i32p), C.byref(num), *extra)
            else:
                m = np.full(n1, -5, dtype=np.int32)
                rc = getattr(lib_, prefix + "bow_match_keyframes")(v1.ref(), A.ptr(d1, A.u8p), A.ptr(a1, A.f32p), A.ptr(valid1, A.u8p), n1,
                                                                   v2.ref(), A.ptr(d2, A.u8p), A.ptr(a2, A.f32p), A.ptr(valid2, A.u8p), n2,
                                                                   0.75, check, A.ptr(m, A.i32p), C.byref(num), *extra)
            assert rc == 0
            res.append((m, num.value))
        assert res[0][1] == res[1][1], name
        assert np.array_equal(res[0][0], res[1][0]), name
        if words >= 40:
            assert res[0][1] > 20, name


def test_bow_out_of_range_index():
    d = np.zeros((4, 32), dtype=np.uint8)
    a = np.zeros(4, dtype=np.float32)
    v_ok = A.make_bow_feat_vec({1: [0, 1], 2: [2, 3]})
    m = np.zeros(4, dtype=np.int32)
    num = C.c_uint32(0)
    # 9: past the four keypoints; 2^31 and 2^32 - 1: negative when read as a signed int
    for bad in (9, 4, 1 << 31, (1 << 32) - 1):
        v_bad = A.make_bow_feat_vec({1: [0, bad]})
        for lib_, prefix, extra in ((_lib.lib(), "ovs_", (0,)), (oracle.lib(), "ovo_", ())):
            for first, second in ((v_bad, v_ok), (v_ok, v_bad)):
                rc = getattr(lib_, prefix + "bow_match_frame_and_keyframe")(first.ref(), A.ptr(d, A.u8p), A.ptr(a, A.f32p), None, 4,
                                                                            second.ref(), A.ptr(d, A.u8p), A.ptr(a, A.f32p), 4,
                                                                            0.75, 0, A.ptr(m, A.i32p), C.byref(num), *extra)
                assert rc == A.OVS_ERR_OUT_OF_RANGE, (prefix, bad)

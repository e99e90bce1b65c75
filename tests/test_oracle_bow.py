"""The oracle's restatement of DBoW2's TemplatedVocabulary::transform (oracle/ovo_bow.cc) against first
principles.  PARITY UNPINNED: DBoW2 and a vocabulary file are not in the reference tree and no reference test
holds a BowVector or FeatureVector (the call sites are src/openvslam/data/frame.cc:162-170 and
data/keyframe.cc:181-189); what can be checked is that the oracle does what DBoW2's published algorithm says:
greedy descent by Hamming distance with the first of equally distant children winning, the node at level
L - levelsup as the feature-vector key, stopped words dropped, weights added per descriptor in descriptor
order, the normalisation of the scoring type.  A pure-Python restatement with dicts and Python floats (IEEE
doubles) is the second opinion."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle
from openvslam_amd import _abi as A
from openvslam_amd import bow


def oracle_transform(voc, descs, levelsup, weighting=A.BOW_TF_IDF, scoring=A.BOW_L1_NORM):
    view = A.make_bow_vocab_view(voc["k"], voc["L"], voc["node_descs"], voc["child_offsets"], voc["children"], voc["word_ids"], voc["weights"],
                                 weighting, scoring)
    d = np.ascontiguousarray(descs, dtype=np.uint8).reshape(-1, 32)
    n = len(d)
    word_of, node_of = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    ids, vals = np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.float64)
    node_ids, offsets, indices = np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.uint32), np.zeros(n + 1, np.uint32)
    nw, nn = C.c_int32(0), C.c_int32(0)
    oracle.check(oracle.lib().ovo_bow_transform(view.ref(), A.ptr(d, A.u8p), n, levelsup, A.ptr(word_of, A.u32p), A.ptr(node_of, A.u32p),
                                                A.ptr(ids, A.u32p), A.ptr(vals, A.f64p), C.byref(nw), A.ptr(node_ids, A.u32p),
                                                A.ptr(offsets, A.u32p), A.ptr(indices, A.u32p), C.byref(nn)))
    return dict(word_of=word_of, node_of=node_of, word_ids=ids[:nw.value].copy(), values=vals[:nw.value].copy(),
                node_ids=node_ids[:nn.value].copy(), offsets=offsets[:nn.value + 1].copy(), indices=indices[:offsets[nn.value]].copy())


def python_transform(voc, descs, levelsup, weighting, scoring):
    co, ch, nd = voc["child_offsets"], voc["children"], voc["node_descs"]
    v, fv = {}, {}
    tf = weighting in (A.BOW_TF, A.BOW_TF_IDF)
    for i, f in enumerate(np.asarray(descs, np.uint8).reshape(-1, 32)):
        node, level, nid, nid_level = 0, 0, None, voc["L"] - levelsup
        if nid_level <= 0:
            nid = 0
        while co[node + 1] > co[node]:
            level += 1
            kids = ch[co[node]:co[node + 1]]
            dist = [int(np.unpackbits(f ^ nd[c]).sum()) for c in kids]
            node = int(kids[dist.index(min(dist))])  # list.index: the first minimum
            if level == nid_level:
                nid = node
        if nid is None:
            nid = node
        w = float(voc["weights"][node])
        if w > 0:
            word = int(voc["word_ids"][node])
            if tf:
                v[word] = v[word] + w if word in v else w
            else:
                v.setdefault(word, w)
            fv.setdefault(nid, []).append(i)
    words = sorted(v)
    if scoring == A.BOW_DOT_PRODUCT:
        if tf and v:
            nd_ = float(len(v))
            for k in words:
                v[k] /= nd_
    else:
        norm = 0.0
        for k in words:
            norm += v[k] * v[k] if scoring == A.BOW_L2_NORM else abs(v[k])
        if scoring == A.BOW_L2_NORM:
            norm = math.sqrt(norm)
        if norm > 0.0:
            for k in words:
                v[k] /= norm
    return words, [v[k] for k in words], fv


def noisy_leaf_descriptors(voc, n, seed, max_flips=12):
    """Descriptors near randomly chosen leaves (so the descent is not a coin toss) plus uniform ones."""
    rng = np.random.default_rng(seed)
    leaves = np.nonzero(voc["child_offsets"][1:] == voc["child_offsets"][:-1])[0]
    d = voc["node_descs"][rng.choice(leaves, n)].copy()
    for i in range(n):
        for b in rng.choice(256, rng.integers(0, max_flips + 1), replace=False):
            d[i, b >> 3] ^= np.uint8(1 << (b & 7))
    d[::7] = rng.integers(0, 256, (len(d[::7]), 32), dtype=np.uint8)
    return d


@pytest.mark.parametrize("k,L,levelsup,weighting,scoring,kw", [
    (4, 3, 2, A.BOW_TF_IDF, A.BOW_L1_NORM, {}),
    (3, 4, 4, A.BOW_TF_IDF, A.BOW_L1_NORM, {"early_leaf_fraction": 0.3}),   # nid level 0: every descriptor under the root
    (5, 2, 1, A.BOW_TF, A.BOW_L2_NORM, {"shuffle_ids": True}),
    (4, 3, 1, A.BOW_IDF, A.BOW_DOT_PRODUCT, {}),
    (4, 3, 2, A.BOW_TF_IDF, A.BOW_DOT_PRODUCT, {"stop_fraction": 0.3}),
    (3, 4, 3, A.BOW_BINARY, A.BOW_CHI_SQUARE, {"tie_fraction": 0.5}),
    (3, 3, 7, A.BOW_TF_IDF, A.BOW_KL, {"early_leaf_fraction": 0.4, "shuffle_ids": True}),
])
def test_oracle_equals_python_restatement(k, L, levelsup, weighting, scoring, kw):
    voc = bow.synthetic_vocabulary(k, L, seed=k * 10 + L, **kw)
    descs = noisy_leaf_descriptors(voc, 300, seed=L)
    got = oracle_transform(voc, descs, levelsup, weighting, scoring)
    words, values, fv = python_transform(voc, descs, levelsup, weighting, scoring)
    assert got["word_ids"].tolist() == words
    assert got["values"].tobytes() == np.array(values, np.float64).tobytes()  # the same doubles, bit for bit
    assert got["node_ids"].tolist() == sorted(fv)
    for j, nid in enumerate(sorted(fv)):
        assert got["indices"][got["offsets"][j]:got["offsets"][j + 1]].tolist() == fv[nid]
    assert len(words) >= 2


def test_first_of_equally_distant_children_wins_and_leaf_is_the_word():
    # root -> 3 children; children 1 and 2 (ids 2, 3) have the same descriptor: a query equal to it goes to id 2
    nd = np.zeros((4, 32), np.uint8)
    nd[1] = 0xFF
    nd[2] = nd[3] = 0x0F
    voc = dict(k=3, L=1, node_descs=nd, child_offsets=np.array([0, 3, 3, 3, 3], np.uint32), children=np.array([1, 2, 3], np.uint32),
               word_ids=np.array([0, 10, 11, 12], np.uint32), weights=np.array([0, 1.0, 2.0, 4.0]))
    q = np.stack([np.full(32, 0x0F, np.uint8), np.full(32, 0xFF, np.uint8), np.full(32, 0x0F, np.uint8)])
    got = oracle_transform(voc, q, levelsup=0, weighting=A.BOW_TF_IDF, scoring=A.BOW_L1_NORM)
    assert got["word_of"].tolist() == [11, 10, 11]
    assert got["word_ids"].tolist() == [10, 11]
    assert got["values"].tolist() == [1.0 / 5.0, 4.0 / 5.0]  # word 11: 2 + 2, L1 norm 5
    # levelsup 0: the node of a descriptor is its leaf
    assert got["node_ids"].tolist() == [1, 2] and got["indices"].tolist() == [1, 0, 2]
    # one level up: the root
    up = oracle_transform(voc, q, levelsup=1)
    assert up["node_ids"].tolist() == [0] and up["indices"].tolist() == [0, 1, 2]


def test_stopped_words_leave_no_trace():
    voc = bow.synthetic_vocabulary(3, 2, seed=5, stop_fraction=0.0)
    descs = noisy_leaf_descriptors(voc, 80, seed=1)
    ref = oracle_transform(voc, descs, 1)
    stop_word = int(ref["word_ids"][0])
    voc["weights"][np.nonzero((voc["word_ids"] == stop_word) & (voc["child_offsets"][1:] == voc["child_offsets"][:-1]))[0]] = 0.0
    got = oracle_transform(voc, descs, 1)
    assert stop_word not in got["word_ids"].tolist()
    gone = np.nonzero(ref["word_of"] == stop_word)[0]
    assert len(gone) > 0 and (got["word_of"][gone] == 0xFFFFFFFF).all()
    assert not set(gone.tolist()) & set(got["indices"].tolist())
    assert abs(got["values"].sum() - 1.0) < 1e-12  # L1


def test_empty_vocabulary_and_no_descriptors():
    voc = dict(k=10, L=6, node_descs=np.zeros((1, 32), np.uint8), child_offsets=np.array([0, 0], np.uint32), children=np.zeros(0, np.uint32),
               word_ids=np.zeros(1, np.uint32), weights=np.zeros(1))
    got = oracle_transform(voc, np.zeros((5, 32), np.uint8), 4)
    assert len(got["word_ids"]) == 0 and len(got["node_ids"]) == 0
    voc = bow.synthetic_vocabulary(3, 2, seed=5)
    got = oracle_transform(voc, np.zeros((0, 32), np.uint8), 4)
    assert len(got["word_ids"]) == 0 and len(got["node_ids"]) == 0 and got["offsets"].tolist() == [0]


def test_library_refuses_arrays_that_are_not_a_tree():
    """ovs_bow_vocab_create validates on the host before it touches a device."""
    from openvslam_amd import _lib
    L = _lib.lib()
    good = bow.synthetic_vocabulary(3, 2, seed=1)

    def create(**change):
        voc = dict(good)
        voc.update(change)
        view = A.make_bow_vocab_view(voc["k"], voc["L"], voc["node_descs"], voc["child_offsets"], voc["children"], voc["word_ids"], voc["weights"],
                                     voc.get("weighting", 0), voc.get("scoring", 0))
        h = C.c_void_p()
        rc = L.ovs_bow_vocab_create(view.ref(), 0, C.byref(h))
        if rc == 0:
            L.ovs_bow_vocab_destroy(h)
        return rc, L.ovs_last_error()

    twice = good["children"].copy()
    twice[1] = twice[0]
    assert create(children=twice)[0] == A.OVS_ERR_INVALID_ARGUMENT
    to_root = good["children"].copy()
    to_root[2] = 0
    assert create(children=to_root)[0] == A.OVS_ERR_INVALID_ARGUMENT
    cyc = good["children"].copy()  # node 4 (child of 1) becomes a child of itself's subtree: unreachable cycle
    co = good["child_offsets"].copy()
    # move the first child of the root under node 4 and node 4 ... simplest cycle: swap so that 1's list names 1
    cyc[int(co[1])] = 1
    cyc[0] = 4
    rc, msg = create(children=cyc)
    assert rc == A.OVS_ERR_INVALID_ARGUMENT and (b"reach" in msg or b"twice" in msg)
    bad_off = co.copy()
    bad_off[2], bad_off[3] = bad_off[3], bad_off[2] - 1
    assert create(child_offsets=bad_off)[0] == A.OVS_ERR_INVALID_ARGUMENT
    assert create(weighting=7)[0] == A.OVS_ERR_INVALID_ARGUMENT
    assert create(scoring=-1)[0] == A.OVS_ERR_INVALID_ARGUMENT

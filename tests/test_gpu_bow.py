"""GPU parity of the BoW transform (frame::compute_bow, src/openvslam/data/frame.cc:162-170) against the oracle's
restatement of DBoW2's TemplatedVocabulary::transform: word and node of every descriptor, the BowVector (doubles
compared as bit patterns) and the FeatureVector, bit for bit.  PARITY UNPINNED against the reference (DBoW2 and
the vocabulary file are not in its tree); the vocabularies here are seeded synthetic trees."""
import ctypes as C

import numpy as np
import pytest

import oracle
from openvslam_amd import _abi as A
from openvslam_amd import _lib, bow, synth
from openvslam_amd.feature import orb_extractor, orb_params
from test_oracle_bow import noisy_leaf_descriptors, oracle_transform

pytestmark = pytest.mark.gpu


def make_vocab(voc, weighting=A.BOW_TF_IDF, scoring=A.BOW_L1_NORM):
    return bow.bow_vocabulary(voc["k"], voc["L"], voc["node_descs"], voc["child_offsets"], voc["children"], voc["word_ids"], voc["weights"],
                              weighting, scoring)


def compare(voc, descs, levelsup, weighting=A.BOW_TF_IDF, scoring=A.BOW_L1_NORM):
    want = oracle_transform(voc, descs, levelsup, weighting, scoring)
    v = make_vocab(voc, weighting, scoring)
    (ids, vals), fv, (word_of, node_of) = v.transform(descs, levelsup, per_feature=True)
    assert np.array_equal(word_of, want["word_of"]), "word of a descriptor differs"
    assert np.array_equal(node_of, want["node_of"]), "node of a descriptor differs"
    assert np.array_equal(ids, want["word_ids"])
    assert vals.tobytes() == want["values"].tobytes(), "BowVector values differ in some bit"
    node_ids, offsets, indices = fv.keep
    assert np.array_equal(node_ids, want["node_ids"]) and np.array_equal(offsets, want["offsets"]) and np.array_equal(indices, want["indices"])
    return want


@pytest.mark.parametrize("k,L,n", [(10, 6, 2000), (8, 3, 2000), (10, 6, 2243), (8, 3, 5000)])
def test_transform_matches_oracle(k, L, n):
    # the reference's vocabulary shape (k = 10, L = 6, levelsup = 4: orb_vocab.dbow2) and a small one; 5 % of
    # the children are copies of an earlier sibling, so equal distances occur on every level
    voc = bow.synthetic_vocabulary(k, L, seed=k + L)
    descs = noisy_leaf_descriptors(voc, n, seed=n)
    want = compare(voc, descs, 4)
    assert len(want["word_ids"]) > min(n, k ** L) // 4
    if L == 6:
        assert 50 <= len(want["node_ids"]) <= 100  # level 2 of a k = 10 tree


@pytest.mark.parametrize("weighting,scoring", [(A.BOW_TF, A.BOW_L2_NORM), (A.BOW_TF_IDF, A.BOW_DOT_PRODUCT), (A.BOW_IDF, A.BOW_L1_NORM),
                                               (A.BOW_BINARY, A.BOW_DOT_PRODUCT), (A.BOW_TF_IDF, A.BOW_BHATTACHARYYA)])
def test_weighting_and_scoring_types(weighting, scoring):
    voc = bow.synthetic_vocabulary(6, 3, seed=3, stop_fraction=0.1)
    # few words for many descriptors: long chains of w + w + ... per word
    descs = noisy_leaf_descriptors(voc, 3000, seed=4, max_flips=4)
    compare(voc, descs, 1, weighting, scoring)


@pytest.mark.parametrize("levelsup", [0, 2, 3, 9])
def test_irregular_trees_and_every_node_level(levelsup):
    # leaves above the last level, node ids in random order (children lists that are not id ranges), heavy ties
    voc = bow.synthetic_vocabulary(5, 4, seed=11, early_leaf_fraction=0.3, shuffle_ids=True, tie_fraction=0.3)
    descs = noisy_leaf_descriptors(voc, 1500, seed=12)
    compare(voc, descs, levelsup)


def test_wide_nodes_and_small_inputs():
    # more children than the 16 lanes that share a descriptor; 1, 15, 16, 17 descriptors; none
    voc = bow.synthetic_vocabulary(37, 2, seed=21, tie_fraction=0.2)
    for n in (1, 15, 16, 17, 300):
        compare(voc, noisy_leaf_descriptors(voc, n, seed=n), 1)
    v = make_vocab(voc)
    (ids, vals), fv = v.transform(np.zeros((0, 32), np.uint8))
    assert len(ids) == 0 and fv.struct.num_nodes == 0


def test_batch_on_the_device_with_ragged_counts():
    import torch
    dev = torch.device("cuda", 0)
    voc = bow.synthetic_vocabulary(10, 4, seed=31)
    v = make_vocab(voc)
    counts = np.array([1800, 0, 2300, 17, 1024], np.int32)
    n_frames, cap = len(counts), 2304
    descs = np.zeros((n_frames, cap, 32), np.uint8)
    for f, c in enumerate(counts):
        descs[f, :c] = noisy_leaf_descriptors(voc, int(c), seed=40 + f) if c else 0
    d_descs = torch.from_numpy(descs).to(dev)
    d_counts = torch.from_numpy(counts).to(dev)
    word_of = torch.zeros((n_frames, cap), dtype=torch.int32, device=dev)
    node_of = torch.zeros((n_frames, cap), dtype=torch.int32, device=dev)
    bow_ids = torch.zeros((n_frames, cap), dtype=torch.int32, device=dev)
    bow_vals = torch.zeros((n_frames, cap), dtype=torch.float64, device=dev)
    fv_ids = torch.zeros((n_frames, cap), dtype=torch.int32, device=dev)
    fv_off = torch.zeros((n_frames, cap + 1), dtype=torch.int32, device=dev)
    fv_idx = torch.zeros((n_frames, cap), dtype=torch.int32, device=dev)
    nw = torch.zeros(n_frames, dtype=torch.int32, device=dev)
    nn = torch.zeros(n_frames, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(2):  # the second call reuses the scratch arena of the first
        _lib.check(_lib.lib().ovs_bow_transform_batch_dev(v._h, d_descs.data_ptr(), d_counts.data_ptr(), n_frames, cap, 2, word_of.data_ptr(),
                                                          node_of.data_ptr(), bow_ids.data_ptr(), bow_vals.data_ptr(), nw.data_ptr(), fv_ids.data_ptr(),
                                                          fv_off.data_ptr(), fv_idx.data_ptr(), nn.data_ptr(), stream))
    torch.cuda.synchronize(dev)
    h_nw, h_nn = nw.cpu().numpy(), nn.cpu().numpy()
    for f, c in enumerate(counts):
        want = oracle_transform(voc, descs[f, :c], 2)
        assert h_nw[f] == len(want["word_ids"]) and h_nn[f] == len(want["node_ids"])
        assert np.array_equal(word_of[f, :c].cpu().numpy().view(np.uint32), want["word_of"])
        assert np.array_equal(node_of[f, :c].cpu().numpy().view(np.uint32), want["node_of"])
        assert np.array_equal(bow_ids[f, :h_nw[f]].cpu().numpy().view(np.uint32), want["word_ids"])
        assert bow_vals[f, :h_nw[f]].cpu().numpy().tobytes() == want["values"].tobytes()
        assert np.array_equal(fv_ids[f, :h_nn[f]].cpu().numpy().view(np.uint32), want["node_ids"])
        assert np.array_equal(fv_off[f, :h_nn[f] + 1].cpu().numpy().view(np.uint32), want["offsets"])
        assert np.array_equal(fv_idx[f, :len(want["indices"])].cpu().numpy().view(np.uint32), want["indices"])


def test_extract_transform_match_chain():
    """data/frame.cc:162-170 then match/bow_tree.cc:16-122 with no host BoW pass in between: two frames of a
    sequence are extracted, transformed and matched through the library; the oracle does the same chain."""
    frames = synth.make_sequence(640, 480, 2, seed=61, dx=4, dy=1)
    ext = orb_extractor(orb_params(max_num_keypts=1000))
    (k1, d1), (k2, d2) = ext.extract(frames[0]), ext.extract(frames[1])
    # a vocabulary grown around the first frame's descriptors, so that buckets hold true matches
    voc = bow.synthetic_vocabulary(10, 3, seed=62)
    leaves = np.nonzero(voc["child_offsets"][1:] == voc["child_offsets"][:-1])[0]
    v = make_vocab(voc)
    (_, _), fv1 = v.transform(d1, 1)
    (_, _), fv2 = v.transform(d2, 1)
    ref = oracle.OrbExtractor(max_num_keypts=1000)
    (rk1, rd1), (rk2, rd2) = ref.extract(frames[0]), ref.extract(frames[1])
    w1, w2 = oracle_transform(voc, rd1, 1), oracle_transform(voc, rd2, 1)
    for fv, w in ((fv1, w1), (fv2, w2)):
        assert np.array_equal(fv.keep[0], w["node_ids"]) and np.array_equal(fv.keep[1], w["offsets"]) and np.array_equal(fv.keep[2], w["indices"])

    def holder(w):
        f = A.BowFeatVec()
        f.num_nodes = len(w["node_ids"])
        f.node_ids, f.offsets, f.indices = A.ptr(w["node_ids"], A.u32p), A.ptr(w["offsets"], A.u32p), A.ptr(w["indices"], A.u32p)
        return A.Holder(f, w)
    valid = np.ones(len(k1), np.uint8)
    res = []
    for lib_, prefix, extra, a, b, da, db, ka, kb in ((_lib.lib(), "ovs_", (0,), fv1, fv2, d1, d2, k1, k2),
                                                      (oracle.lib(), "ovo_", (), holder(w1), holder(w2), rd1, rd2, rk1, rk2)):
        m = np.full(len(kb), -5, dtype=np.int32)
        num = C.c_uint32(999)
        a1, a2 = np.ascontiguousarray(ka["angle"]), np.ascontiguousarray(kb["angle"])
        rc = getattr(lib_, prefix + "bow_match_frame_and_keyframe")(a.ref(), A.ptr(da, A.u8p), A.ptr(a1, A.f32p), A.ptr(valid, A.u8p), len(ka),
                                                                    b.ref(), A.ptr(db, A.u8p), A.ptr(a2, A.f32p), len(kb),
                                                                    0.75, 1, A.ptr(m, A.i32p), C.byref(num), *extra)
        assert rc == 0
        res.append((m, num.value))
    assert res[0][1] == res[1][1] and np.array_equal(res[0][0], res[1][0])
    assert res[0][1] > 50 and len(leaves) == 1000

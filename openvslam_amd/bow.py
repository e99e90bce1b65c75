"""Host-side mirror of data::bow_vocabulary (src/openvslam/data/bow_vocabulary.h:16, the DBoW2 type) as
frame::compute_bow uses it (src/openvslam/data/frame.cc:162-170): transform(descriptors, levelsup) gives the
BowVector and the FeatureVector, computed on the device.  The vocabulary comes in as arrays; reading a
vocabulary file stays with the caller."""
import ctypes as C

import numpy as np

from . import _abi as A
from ._lib import check, lib


class bow_vocabulary:
    def __init__(self, k, L, node_descs, child_offsets, children, word_ids, weights, weighting=A.BOW_TF_IDF, scoring=A.BOW_L1_NORM,
                 device=0):
        self.view = A.make_bow_vocab_view(k, L, node_descs, child_offsets, children, word_ids, weights, weighting, scoring)
        self._h = C.c_void_p()
        check(lib().ovs_bow_vocab_create(self.view.ref(), device, C.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                lib().ovs_bow_vocab_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    def transform(self, descriptors, levelsup=4, per_feature=False):
        """-> (bow_vec, bow_feat_vec): bow_vec = (word ids ascending, values as float64), bow_feat_vec = the
        flattened FeatureVector (an _abi.Holder of BowFeatVec) the bow_tree matchers take.  per_feature adds
        (word id, node id) of every descriptor."""
        d = np.ascontiguousarray(descriptors, dtype=np.uint8).reshape(-1, 32)
        n = len(d)
        word_of, node_of = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        ids, vals = np.zeros(max(n, 1), np.uint32), np.zeros(max(n, 1), np.float64)
        node_ids, offsets, indices = np.zeros(max(n, 1), np.uint32), np.zeros(n + 1, np.uint32), np.zeros(max(n, 1), np.uint32)
        nw, nn = C.c_int32(0), C.c_int32(0)
        check(lib().ovs_bow_transform(self._h, A.ptr(d, A.u8p), n, levelsup, A.ptr(word_of, A.u32p), A.ptr(node_of, A.u32p),
                                      A.ptr(ids, A.u32p), A.ptr(vals, A.f64p), C.byref(nw), A.ptr(node_ids, A.u32p), A.ptr(offsets, A.u32p),
                                      A.ptr(indices, A.u32p), C.byref(nn)))
        node_ids = np.ascontiguousarray(node_ids[:nn.value])
        offsets = np.ascontiguousarray(offsets[:nn.value + 1])
        indices = np.ascontiguousarray(indices[:int(offsets[-1]) if nn.value else 0])
        fv = A.BowFeatVec()
        fv.num_nodes = nn.value
        fv.node_ids, fv.offsets, fv.indices = A.ptr(node_ids, A.u32p), A.ptr(offsets, A.u32p), A.ptr(indices, A.u32p)
        out = (ids[:nw.value].copy(), vals[:nw.value].copy()), A.Holder(fv, node_ids, offsets, indices)
        return out + ((word_of, node_of),) if per_feature else out


def synthetic_vocabulary(k, L, seed=0, tie_fraction=0.05, stop_fraction=0.02, early_leaf_fraction=0.0, shuffle_ids=False):
    """A seeded random vocabulary tree with branching factor k and depth L, as the arrays bow_vocabulary
    takes.  The descriptor of a child is its parent's with random bits flipped (a quarter of them on the
    first two levels, then an eighth, then a sixteenth), so a descriptor near a leaf descends towards it; a
    fraction of the children are exact copies of an earlier sibling (equal distances: the first one must
    win); a fraction of the words have weight 0 (stopped); a fraction of the inner nodes stay leaves (words
    above the last level).  Nodes are numbered level by level, the children of a node consecutively, unless
    shuffle_ids permutes the ids of all nodes but the root.  Word ids number the leaves in node order."""
    rng = np.random.default_rng(seed)
    descs = [rng.integers(0, 256, (1, 32), dtype=np.uint8)]
    parent_of = [np.zeros(1, np.int64)]
    level_first, level_parents = 0, np.zeros(1, np.int64)  # ids of the nodes that get children at this level
    n = 1
    for level in range(1, L + 1):
        par = level_parents
        if level > 1 and early_leaf_fraction > 0:
            par = par[rng.random(len(par)) >= early_leaf_fraction]
        if len(par) == 0:
            break
        all_descs = np.concatenate(descs)
        base = np.repeat(all_descs[par], k, axis=0)
        m = 2 if level <= 2 else (3 if level <= 4 else 4)
        flip = rng.integers(0, 256, base.shape, dtype=np.uint8)
        for _ in range(m - 1):
            flip &= rng.integers(0, 256, base.shape, dtype=np.uint8)
        child = base ^ flip
        child = child.reshape(len(par), k, 32)
        for c in range(1, k):
            tie = rng.random(len(par)) < tie_fraction
            src = rng.integers(0, c, len(par))
            child[tie, c] = child[tie, src[tie]]
        child = child.reshape(-1, 32)
        descs.append(child)
        parent_of.append(np.repeat(par, k))
        level_parents = n + np.arange(len(child), dtype=np.int64)
        n += len(child)
    node_descs = np.concatenate(descs)
    parent = np.concatenate(parent_of)
    parent[0] = -1
    ids = np.arange(n, dtype=np.int64)
    if shuffle_ids:
        perm = np.concatenate([[0], 1 + rng.permutation(n - 1)])  # new id of old node
        inv = np.empty(n, np.int64)
        inv[perm] = ids
        node_descs = node_descs[inv]
        new_parent = np.full(n, -1, np.int64)
        new_parent[perm[1:]] = perm[parent[1:]]
        # children keep their creation order inside a family
        order = 1 + np.argsort(parent[1:], kind="stable")           # old ids grouped by old parent, creation order
        children = perm[order]
        counts = np.bincount(perm[parent[1:]], minlength=n)
        # the groups must follow the NEW parent order
        starts_old = np.zeros(n + 1, np.int64)
        np.cumsum(np.bincount(parent[1:], minlength=n), out=starts_old[1:])
        child_offsets = np.zeros(n + 1, np.int64)
        np.cumsum(counts, out=child_offsets[1:])
        out_children = np.zeros(n - 1, np.int64)
        for old in np.nonzero(np.bincount(parent[1:], minlength=n))[0]:
            new = perm[old]
            out_children[child_offsets[new]:child_offsets[new + 1]] = children[starts_old[old]:starts_old[old + 1]]
        children = out_children
    else:
        counts = np.bincount(parent[1:], minlength=n)
        child_offsets = np.zeros(n + 1, np.int64)
        np.cumsum(counts, out=child_offsets[1:])
        children = ids[1:]  # level by level, families consecutive and in parent order
    is_leaf = (child_offsets[1:] - child_offsets[:-1]) == 0
    word_ids = np.zeros(n, np.uint32)
    word_ids[is_leaf] = np.arange(int(is_leaf.sum()), dtype=np.uint32)
    weights = np.where(rng.random(n) < stop_fraction, 0.0, np.log(rng.uniform(1.5, 4000.0, n)))
    weights[~is_leaf] = 0.0
    return dict(k=k, L=L, node_descs=node_descs, child_offsets=child_offsets.astype(np.uint32), children=children.astype(np.uint32),
                word_ids=word_ids, weights=weights.astype(np.float64))

"""GPU parity of the descriptor-distance entry points, the grid window query and the sharded
top-2 database search (bit-exact against the CPU oracle)."""
import ctypes as C

import numpy as np
import pytest

import oracle
from openvslam_amd import _abi as A
from openvslam_amd import _lib, match

pytestmark = pytest.mark.gpu


def test_hamming_known_answers_on_device():
    # test/openvslam/match/base.cc of the reference: equal -> 0, complement -> 256, half -> 128
    z = np.zeros(32, np.uint8)
    f = np.full(32, 0xFF, np.uint8)
    half = np.array([0x0F] * 32, np.uint8)
    assert match.compute_descriptor_distance_32(z, z) == 0
    assert match.compute_descriptor_distance_32(z, f) == 256
    assert match.compute_descriptor_distance_32(f, half) == 128
    assert match.compute_descriptor_distance_32(z, half) == 128


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 4097, 100000])
def test_hamming_pairs(n):
    rng = np.random.default_rng(n)
    a = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    got = np.zeros(n, np.uint32)
    want = np.zeros(n, np.uint32)
    _lib.check(_lib.lib().ovs_hamming_pairs(A.ptr(a, A.u8p), A.ptr(b, A.u8p), n, A.ptr(got, A.u32p), 0))
    oracle.check(oracle.lib().ovo_hamming_pairs(A.ptr(a, A.u8p), A.ptr(b, A.u8p), n, A.ptr(want, A.u32p)))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("na,nb", [(1, 1), (7, 300), (257, 129), (2000, 2100)])
def test_hamming_matrix(na, nb):
    rng = np.random.default_rng(na * 7 + nb)
    a = rng.integers(0, 256, (na, 32), dtype=np.uint8)
    b = rng.integers(0, 256, (nb, 32), dtype=np.uint8)
    got = match.descriptor_distance_matrix(a, b)
    want = np.zeros((na, nb), np.uint16)
    oracle.check(oracle.lib().ovo_hamming_matrix(A.ptr(a, A.u8p), na, A.ptr(b, A.u8p), nb, A.ptr(want, A.u16p)))
    assert np.array_equal(got, want)


def test_unaligned_descriptor_buffers():
    """Caller buffers only need byte alignment on the host side (they are staged)."""
    rng = np.random.default_rng(3)
    raw = rng.integers(0, 256, 32 * 100 + 3, dtype=np.uint8)
    a = raw[3:].reshape(100, 32)
    b = np.ascontiguousarray(a[::-1])
    got = np.zeros(100, np.uint32)
    want = np.zeros(100, np.uint32)
    _lib.check(_lib.lib().ovs_hamming_pairs(A.ptr(a, A.u8p), A.ptr(b, A.u8p), 100, A.ptr(got, A.u32p), 0))
    oracle.check(oracle.lib().ovo_hamming_pairs(A.ptr(np.ascontiguousarray(a), A.u8p), A.ptr(b, A.u8p), 100, A.ptr(want, A.u32p)))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("seed,n,nq", [(1, 2000, 500), (2, 30, 64), (3, 0, 5), (4, 5000, 1)])
def test_get_keypoints_in_cell(seed, n, nq):
    rng = np.random.default_rng(seed)
    kps = np.zeros(n, dtype=A.KEYPOINT_DTYPE)
    kps["x"] = rng.uniform(0, 1241, n)
    kps["y"] = rng.uniform(0, 376, n)
    kps["octave"] = rng.integers(0, 8, n)
    desc = np.zeros((n, 32), np.uint8)
    sf = np.array([1.2 ** i for i in range(8)], np.float32)
    view = A.make_frame_view(kps, desc, sf, A.make_grid(0.0, 0.0, 1241.0, 376.0))
    xy = np.stack([rng.uniform(-50, 1300, nq), rng.uniform(-50, 420, nq)], axis=1).astype(np.float32)
    margin = rng.uniform(1, 80, nq).astype(np.float32)
    lo = rng.integers(-1, 5, nq).astype(np.int32)
    hi = np.where(rng.random(nq) < 0.3, -1, lo + rng.integers(0, 4, nq)).astype(np.int32)
    cap = 400000
    res = []
    for lib_, pre, extra in ((_lib.lib(), "ovs_", (0,)), (oracle.lib(), "ovo_", ())):
        idx = np.zeros(cap, np.uint32)
        off = np.zeros(nq + 1, np.uint32)
        rc = getattr(lib_, pre + "get_keypoints_in_cell")(view.ref(), nq, A.ptr(xy, A.f32p), A.ptr(margin, A.f32p), A.ptr(lo, A.i32p),
                                                        A.ptr(hi, A.i32p), A.ptr(idx, A.u32p), cap, A.ptr(off, A.u32p), *extra)
        assert rc == 0
        res.append((off, idx[:off[-1]].copy()))
    assert np.array_equal(res[0][0], res[1][0])
    assert np.array_equal(res[0][1], res[1][1])
    if n >= 2000 and nq >= 500:
        assert res[0][0][-1] > 1000


@pytest.mark.parametrize("cols,rows,n", [(128, 128, 3000), (100, 90, 20000), (16, 12, 3000)])
def test_get_keypoints_in_cell_other_grids(cols, rows, n):
    # grids other than the reference's 64 x 48: the largest supported one (16384 cells: 128 KB of cell
    # counters in LDS), one whose counters alone pass 64 KB with more keypoints than the LDS index array
    # holds, and a coarse one
    rng = np.random.default_rng(cols * 1000 + rows)
    nq = 300
    kps = np.zeros(n, dtype=A.KEYPOINT_DTYPE)
    kps["x"] = rng.uniform(0, 1920, n)
    kps["y"] = rng.uniform(0, 960, n)
    kps["octave"] = rng.integers(0, 8, n)
    desc = np.zeros((n, 32), np.uint8)
    sf = np.array([1.2 ** i for i in range(8)], np.float32)
    view = A.make_frame_view(kps, desc, sf, A.make_grid(0.0, 0.0, 1920.0, 960.0, cols, rows))
    xy = np.stack([rng.uniform(-50, 2000, nq), rng.uniform(-50, 1000, nq)], axis=1).astype(np.float32)
    margin = rng.uniform(1, 60, nq).astype(np.float32)
    lo = np.full(nq, -1, np.int32)
    hi = np.full(nq, -1, np.int32)
    cap = 2000000
    res = []
    for lib_, pre, extra in ((_lib.lib(), "ovs_", (0,)), (oracle.lib(), "ovo_", ())):
        idx = np.zeros(cap, np.uint32)
        off = np.zeros(nq + 1, np.uint32)
        rc = getattr(lib_, pre + "get_keypoints_in_cell")(view.ref(), nq, A.ptr(xy, A.f32p), A.ptr(margin, A.f32p), A.ptr(lo, A.i32p),
                                                        A.ptr(hi, A.i32p), A.ptr(idx, A.u32p), cap, A.ptr(off, A.u32p), *extra)
        assert rc == 0
        res.append((off, idx[:off[-1]].copy()))
    assert np.array_equal(res[0][0], res[1][0])
    assert np.array_equal(res[0][1], res[1][1])
    assert res[0][0][-1] > 100


def test_grid_with_too_many_cells_is_refused():
    kps = np.zeros(10, dtype=A.KEYPOINT_DTYPE)
    view = A.make_frame_view(kps, np.zeros((10, 32), np.uint8), np.ones(8, np.float32), A.make_grid(0.0, 0.0, 640.0, 480.0, 129, 128))
    xy = np.array([[100.0, 100.0]], np.float32)
    margin = np.array([5.0], np.float32)
    lo = np.array([-1], np.int32)
    idx = np.zeros(10, np.uint32)
    off = np.zeros(2, np.uint32)
    rc = _lib.lib().ovs_get_keypoints_in_cell(view.ref(), 1, A.ptr(xy, A.f32p), A.ptr(margin, A.f32p), A.ptr(lo, A.i32p), A.ptr(lo, A.i32p),
                                              A.ptr(idx, A.u32p), 10, A.ptr(off, A.u32p), 0)
    assert rc != 0


def test_get_keypoints_in_cell_capacity_error():
    n = 500
    kps = np.zeros(n, dtype=A.KEYPOINT_DTYPE)
    kps["x"], kps["y"] = 100.0, 100.0
    view = A.make_frame_view(kps, np.zeros((n, 32), np.uint8), np.ones(8, np.float32), A.make_grid(0.0, 0.0, 640.0, 480.0))
    xy = np.array([[100.0, 100.0]], np.float32)
    margin = np.array([5.0], np.float32)
    lo = np.array([-1], np.int32)
    idx = np.zeros(10, np.uint32)
    off = np.zeros(2, np.uint32)
    rc = _lib.lib().ovs_get_keypoints_in_cell(view.ref(), 1, A.ptr(xy, A.f32p), A.ptr(margin, A.f32p), A.ptr(lo, A.i32p), A.ptr(lo, A.i32p),
                                              A.ptr(idx, A.u32p), 10, A.ptr(off, A.u32p), 0)
    assert rc == A.OVS_ERR_CAPACITY


def db_search(db, first, q):
    h = C.c_void_p()
    L = _lib.lib()
    _lib.check(L.ovs_desc_db_create(0, C.byref(h)))
    try:
        _lib.check(L.ovs_desc_db_upload(h, A.ptr(db, A.u8p), len(db), first))
        out = np.zeros(len(q), dtype=A.KNN2_DTYPE)
        _lib.check(L.ovs_desc_db_search(h, A.ptr(q, A.u8p), len(q), A.ptr(out, A.knn2p)))
    finally:
        L.ovs_desc_db_destroy(h)
    return out


def oracle_search(db, first, q):
    out = np.zeros(len(q), dtype=A.KNN2_DTYPE)
    oracle.check(oracle.lib().ovo_knn2_search(A.ptr(db, A.u8p), len(db), first, A.ptr(q, A.u8p), len(q), A.ptr(out, A.knn2p)))
    return out


@pytest.mark.parametrize("seed,rows,nq", [(1, 100000, 300), (2, 4096, 64), (3, 4097, 65), (4, 1, 10), (5, 0, 10), (6, 50000, 1)])
def test_desc_db_search(seed, rows, nq):
    rng = np.random.default_rng(seed)
    # few distinct byte values -> many equal distances, so the lowest-row rule is exercised
    db = rng.integers(0, 4, (rows, 32), dtype=np.uint8)
    q = rng.integers(0, 4, (nq, 32), dtype=np.uint8)
    if rows > 10:
        db[rows - 1] = db[0]
        q[0] = db[0]
    got = db_search(db, 12345, q)
    want = oracle_search(db, 12345, q)
    assert got.tobytes() == want.tobytes()


def test_sharded_search_equals_whole_database():
    """Four shards searched one after another on this GPU, merged on the host and on the device,
    equal the scan of the whole database."""
    import torch
    from openvslam_amd import dist as ovd
    rng = np.random.default_rng(11)
    db = rng.integers(0, 8, (40001, 32), dtype=np.uint8)
    q = rng.integers(0, 8, (513, 32), dtype=np.uint8)
    lists = []
    for r in range(4):
        first, count = ovd.shard_rows(len(db), 4, r)
        lists.append(db_search(np.ascontiguousarray(db[first:first + count]), first, q))
    lists = np.stack(lists)
    want = oracle_search(db, 0, q)
    assert ovd.merge_topk_host(lists, 4, len(q)).tobytes() == want.tobytes()
    d_lists = torch.from_numpy(lists.view(np.uint8).reshape(-1)).cuda()
    d_out = torch.zeros(len(q) * 24, dtype=torch.uint8, device="cuda")
    _lib.check(_lib.lib().ovs_knn2_merge_dev(d_lists.data_ptr(), 4, len(q), d_out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert d_out.cpu().numpy().tobytes() == want.tobytes()


def test_full_size_database_shards_find_planted_neighbours():
    """The relocalisation configuration at full size: 2000 queries against 10 000 keyframes x 2000
    descriptors (640 MB) held as 8 shards of 1250 keyframes; 25 % of the queries are planted
    near-duplicates (<= 30 flipped bits) of known database rows.  Checked through properties (the
    planted row is the reported nearest row at the planted distance; the merged result is
    independent of the shard order) and, for a sample of queries, against the oracle's scan of
    the whole database."""
    import torch
    from openvslam_amd import dist as ovd
    rng = np.random.default_rng(2024)
    n_shards, rows_per_shard, nq = 8, 1250 * 2000, 2000
    n_rows = n_shards * rows_per_shard
    q = rng.integers(0, 256, (nq, 32), dtype=np.uint8)
    planted = rng.choice(nq, nq // 4, replace=False)
    planted_rows = rng.integers(0, n_rows, len(planted))
    flips = rng.integers(0, 31, len(planted))
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.ovs_desc_db_create(0, C.byref(h)))
    lists = np.zeros((n_shards, nq), dtype=A.KNN2_DTYPE)
    sample = np.concatenate([planted[:6], np.setdiff1d(np.arange(nq), planted)[:6]])
    want_sample = np.zeros((n_shards, len(sample)), dtype=A.KNN2_DTYPE)
    qs = np.ascontiguousarray(q[sample])
    try:
        for s in range(n_shards):
            # shard s is generated from its own seed; planted rows are overwritten after generation
            shard = np.random.default_rng(1000 + s).integers(0, 256, (rows_per_shard, 32), dtype=np.uint8)
            first = s * rows_per_shard
            for qi, row, nf in zip(planted, planted_rows, flips):
                if first <= row < first + rows_per_shard:
                    d = shard[row - first].copy()
                    for b in np.random.default_rng(int(row)).choice(256, nf, replace=False):
                        d[b >> 3] ^= np.uint8(1 << (b & 7))
                    q[qi] = d
            del shard
        # second pass: search every shard with the final queries (regenerated from the same seeds)
        qs = np.ascontiguousarray(q[sample])
        for s in range(n_shards):
            shard = np.random.default_rng(1000 + s).integers(0, 256, (rows_per_shard, 32), dtype=np.uint8)
            first = s * rows_per_shard
            _lib.check(L.ovs_desc_db_upload(h, A.ptr(shard, A.u8p), rows_per_shard, first))
            _lib.check(L.ovs_desc_db_search(h, A.ptr(q, A.u8p), nq, A.ptr(lists[s], A.knn2p)))
            oracle.check(oracle.lib().ovo_knn2_search(A.ptr(shard, A.u8p), rows_per_shard, first, A.ptr(qs, A.u8p), len(sample),
                                                      A.ptr(want_sample[s], A.knn2p)))
            del shard
    finally:
        L.ovs_desc_db_destroy(h)
    merged = ovd.merge_topk_host(lists, n_shards, nq)
    # planted neighbours: nearest row and distance as planted (random 256-bit rows are ~128 apart)
    assert np.array_equal(merged["row"][planted, 0], planted_rows)
    assert np.array_equal(merged["dist"][planted, 0], flips.astype(np.uint32))
    assert merged["dist"][:, 1].min() > 60 and merged["dist"][np.setdiff1d(np.arange(nq), planted), 0].min() > 60
    # shard order does not matter; device merge equals host merge
    perm = rng.permutation(n_shards)
    assert ovd.merge_topk_host(np.ascontiguousarray(lists[perm]), n_shards, nq).tobytes() == merged.tobytes()
    d_lists = torch.from_numpy(lists.view(np.uint8).reshape(-1)).cuda()
    d_out = torch.zeros(nq * 24, dtype=torch.uint8, device="cuda")
    _lib.check(L.ovs_knn2_merge_dev(d_lists.data_ptr(), n_shards, nq, d_out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert d_out.cpu().numpy().tobytes() == merged.tobytes()
    # oracle on the sample: per-shard results and the merge of the oracle's own lists
    want = ovd.merge_topk_host(want_sample, n_shards, len(sample))
    assert merged[sample].tobytes() == want.tobytes()


def test_sharded_search_device_path_over_rccl_world_of_one():
    """openvslam_amd.dist.sharded_search with a device: per-shard search on the device, the RCCL
    all-gather (backend nccl, a world of one rank here) and ovs_knn2_merge_dev, against the oracle's
    scan.  Few distinct byte values make equal distances common."""
    import os
    import socket
    import torch
    import torch.distributed as dist
    from openvslam_amd import dist as ovd
    rng = np.random.default_rng(21)
    db = rng.integers(0, 4, (60001, 32), dtype=np.uint8)
    q = rng.integers(0, 4, (700, 32), dtype=np.uint8)
    db[59999] = db[3]
    q[0] = db[3]
    want = oracle_search(db, 777, q)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    L = _lib.lib()
    h = C.c_void_p()
    try:
        d_db = torch.from_numpy(db).to(dev)
        d_q = torch.from_numpy(q).to(dev)
        _lib.check(L.ovs_desc_db_create(0, C.byref(h)))
        _lib.check(L.ovs_desc_db_attach_dev(h, d_db.data_ptr(), len(db), 777))

        def local_search(_q):
            out = torch.zeros(len(q) * 24, dtype=torch.uint8, device=dev)
            _lib.check(L.ovs_desc_db_search_dev(h, d_q.data_ptr(), len(q), out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
            return out
        got = ovd.sharded_search(local_search, q, dist=dist, device=dev)
        assert got.tobytes() == want.tobytes()
        # host lists into the device path
        got2 = ovd.sharded_search(lambda _q: db_search(db, 777, q), q, dist=dist, device=dev)
        assert got2.tobytes() == want.tobytes()
        # on a stream of the caller's that is not the current one: the local search is called with that
        # stream current, so search, all-gather and merge are one ordered chain
        side = torch.cuda.Stream(dev)
        for _ in range(3):
            got3 = ovd.sharded_search(local_search, q, dist=dist, device=dev, stream=side)
            assert got3.tobytes() == want.tobytes()
    finally:
        if h:
            L.ovs_desc_db_destroy(h)
        dist.destroy_process_group()

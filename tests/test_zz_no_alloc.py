"""The database search entry point that allocates nothing (ovs_desc_db_search_noalloc_dev) against
ovs_desc_db_search_dev: same results bit for bit, on two calls in a row, the second one smaller (the
built-in scratch array keeps the first call's contents), a direct comparison with the oracle's scan, the
property the entry point exists for (the handle holds no device memory after it, the device's free memory
does not shrink across it), and a refusal beyond its capacity."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
from openvslam_amd import _abi as A
from openvslam_amd import _lib

pytestmark = pytest.mark.gpu


def _search(fn, h, d_q, nq, dev):
    out = torch.full((nq * 24,), 0xAB, dtype=torch.uint8, device=dev)
    _lib.check(fn(h, d_q.data_ptr(), nq, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
    torch.cuda.synchronize(dev)
    return out.cpu().numpy().view(A.KNN2_DTYPE)


def test_noalloc_search_equals_the_allocating_search():
    dev = torch.device("cuda", 0)
    L = _lib.lib()
    rng = np.random.default_rng(77)
    # few distinct byte values: many equal distances, so the lowest-row rule decides
    db_big = torch.from_numpy(rng.integers(0, 4, (300000, 32), dtype=np.uint8)).to(dev)
    q_big = torch.from_numpy(rng.integers(0, 4, (2000, 32), dtype=np.uint8)).to(dev)
    db_small = torch.from_numpy(rng.integers(0, 4, (5000, 32), dtype=np.uint8)).to(dev)
    q_small = torch.from_numpy(rng.integers(0, 4, (40, 32), dtype=np.uint8)).to(dev)
    h = C.c_void_p()
    _lib.check(L.ovs_desc_db_create(0, C.byref(h)))
    held = C.c_int64(-1)
    try:
        # the entry point first, on a handle that has never searched: it must not grow the handle's
        # partial-result buffer (or anything else) -- the handle holds no device memory afterwards, and the
        # device has as much free memory as before the calls
        _lib.check(L.ovs_desc_db_attach_dev(h, db_big.data_ptr(), db_big.shape[0], 0))
        warm = _search(L.ovs_desc_db_search_noalloc_dev, h, q_big, q_big.shape[0], dev)  # code object, streams: set up
        free_before = torch.cuda.mem_get_info(dev)[0]
        for q in (q_big, q_small, q_big[:129]):
            again = _search(L.ovs_desc_db_search_noalloc_dev, h, q, q.shape[0], dev)
        assert torch.cuda.mem_get_info(dev)[0] >= free_before
        _lib.check(L.ovs_desc_db_device_bytes(h, C.byref(held)))
        assert held.value == 0
        assert again.tobytes() == warm[:129].tobytes()
        # first call: matrix-core kernel, 2000 queries; second call: popcount kernel, 40 queries on a
        # smaller database, reading a prefix of the scratch memory the first call filled
        for db, q, offset in ((db_big, q_big, 1000), (db_small, q_small, 0), (db_big, q_big[:129], 5)):
            _lib.check(L.ovs_desc_db_attach_dev(h, db.data_ptr(), db.shape[0], offset))
            got = _search(L.ovs_desc_db_search_noalloc_dev, h, q, q.shape[0], dev)
            want = _search(L.ovs_desc_db_search_dev, h, q, q.shape[0], dev)
            assert got.tobytes() == want.tobytes()
            assert (got["row"][:, 0] >= offset).all()
            if db is db_small:
                # and directly against the oracle's strict-< scan (robust.cc:260-267)
                h_db, h_q = db.cpu().numpy(), q.cpu().numpy()
                ref = np.zeros(len(h_q), dtype=A.KNN2_DTYPE)
                oracle.check(oracle.lib().ovo_knn2_search(A.ptr(h_db, A.u8p), len(h_db), offset, A.ptr(h_q, A.u8p), len(h_q), A.ptr(ref, A.knn2p)))
                assert got.tobytes() == ref.tobytes()
        # the allocating entry point did grow the handle
        _lib.check(L.ovs_desc_db_device_bytes(h, C.byref(held)))
        assert held.value > 0
        # beyond the capacity: 100 queries x 3000 slices of 4096 rows need 4.8 MB of partial results
        n_rows = 3000 * 4096
        huge = torch.zeros((n_rows, 32), dtype=torch.uint8, device=dev)
        _lib.check(L.ovs_desc_db_attach_dev(h, huge.data_ptr(), n_rows, 0))
        out = torch.zeros(100 * 24, dtype=torch.uint8, device=dev)
        rc = L.ovs_desc_db_search_noalloc_dev(h, q_big.data_ptr(), 100, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        assert rc == _lib.OVS_ERR_CAPACITY if hasattr(_lib, "OVS_ERR_CAPACITY") else rc == -2
        assert b"scratch" in L.ovs_last_error()
        # the allocating entry point takes the same search
        _lib.check(L.ovs_desc_db_search_dev(h, q_big.data_ptr(), 100, out.data_ptr(), torch.cuda.current_stream(dev).cuda_stream))
        torch.cuda.synchronize(dev)
    finally:
        L.ovs_desc_db_destroy(h)

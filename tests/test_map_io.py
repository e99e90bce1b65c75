"""Keyframe descriptors out of a saved map (io::map_database_io::save_message_pack,
src/openvslam/io/map_database_io.cc:19-46).  The tree holds no saved map, so the files are built here
with the schema of the reference's to_json functions (data/keyframe.cc:87-136, data/common.cc:28-82,
data/map_database.cc:329-362) and packed by the msgpack module; a file written by the reference
itself was not available: parity unpinned."""
import ctypes as C

import msgpack
import numpy as np
import pytest

from openvslam_amd import _abi as A
from openvslam_amd import io as ovs_io
from openvslam_amd._lib import OvsError


def _keyframe(rng, n, kf_id):
    descs = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    # a few rows whose words need every MessagePack integer width (fixint .. uint32)
    if n >= 2:
        descs[0] = 0
        descs[1, :8] = [0x7F, 0, 0, 0, 0xFF, 0, 0, 0]
        descs[1, 8:16] = [0xFF, 0xFF, 0, 0, 0xFF, 0xFF, 0xFF, 0xFF]
    words = descs.view("<u4").reshape(n, 8)
    return descs, {
        "src_frm_id": kf_id * 3, "ts": 0.1 * kf_id, "cam": "test camera", "depth_thr": 40.0,
        "rot_cw": [1.0, 0.0, 0.0, 0.0], "trans_cw": [0.5, -0.25, 3.0],
        "n_keypts": n,
        "keypts": [{"pt": [float(i), float(i) * 0.5], "ang": 12.5, "oct": i % 8} for i in range(n)],
        "undists": [[float(i), float(i) * 0.5] for i in range(n)],
        "x_rights": [-1.0] * n, "depths": [-1.0] * n,
        "descs": [[int(w) for w in row] for row in words],
        "lm_ids": [(-1 if i % 3 else i) for i in range(n)],
        "n_scale_levels": 8, "scale_factor": 1.2,
        "span_parent": -1, "span_children": [kf_id + 1], "loop_edges": [],
    }


def _write_map(path, rng, sizes):
    expected = {}
    keyframes = {}
    for kf_id, n in sizes.items():
        d, js = _keyframe(rng, n, kf_id)
        expected[kf_id] = d
        keyframes[str(kf_id)] = js
    root = {"cameras": {"test camera": {"model_type": "perspective", "cols": 640, "rows": 480, "fx": 500.0}},
            "keyframes": keyframes,
            "landmarks": {"7": {"1st_keyfrm": 0, "pos_w": [1.0, 2.0, 3.0], "ref_keyfrm": 0, "n_vis": 5, "n_fnd": 4}},
            "frame_next_id": 1000, "keyframe_next_id": 200, "landmark_next_id": 12345}
    path.write_bytes(msgpack.packb(root, use_single_float=False))
    return expected


def test_load_keyframe_descriptors(tmp_path):
    rng = np.random.default_rng(1)
    # ids as the reference writes them: strings, in the lexicographic order of std::map<std::string, ...>
    sizes = {0: 5, 1: 300, 10: 17, 100: 1, 2: 0, 33: 70000}
    expected = _write_map(tmp_path / "map.msg", rng, sizes)
    descs, ids, offsets = ovs_io.load_keyframe_descriptors(tmp_path / "map.msg")
    assert ids.tolist() == sorted(sizes)
    assert offsets[0] == 0 and offsets[-1] == sum(sizes.values()) == len(descs)
    for i, kf_id in enumerate(ids.tolist()):
        assert offsets[i + 1] - offsets[i] == sizes[kf_id]
        assert np.array_equal(descs[offsets[i]:offsets[i + 1]], expected[kf_id])


def test_map_errors(tmp_path):
    with pytest.raises(OvsError) as e:
        ovs_io.load_keyframe_descriptors(tmp_path / "does_not_exist.msg")
    assert e.value.code == A.OVS_ERR_INVALID_ARGUMENT and "cannot load the file" in str(e.value)
    rng = np.random.default_rng(2)
    _write_map(tmp_path / "map.msg", rng, {3: 40, 4: 50})
    whole = (tmp_path / "map.msg").read_bytes()
    # truncated anywhere inside the keyframes: an error, never a crash
    for cut in (1, 10, len(whole) // 3, len(whole) // 2, len(whole) - 200):
        (tmp_path / "cut.msg").write_bytes(whole[:cut])
        with pytest.raises(OvsError):
            ovs_io.load_keyframe_descriptors(tmp_path / "cut.msg")
    # other structures
    for bad in ([1, 2, 3], {"cameras": {}}, {"keyframes": {"x1": {"descs": []}}}, {"keyframes": {"1": {"n_keypts": 0}}},
                {"keyframes": {"1": {"descs": [[1, 2, 3, 4, 5, 6, 7]]}}}, {"keyframes": {"1": {"descs": [[1, 2, 3, 4, 5, 6, 7, -8]]}}},
                {"keyframes": {"1": {"descs": [[1, 2, 3, 4, 5, 6, 7, 2 ** 32]]}}}):
        (tmp_path / "bad.msg").write_bytes(msgpack.packb(bad))
        with pytest.raises(OvsError):
            ovs_io.load_keyframe_descriptors(tmp_path / "bad.msg")
    # an empty keyframes object is a valid, empty database
    (tmp_path / "empty.msg").write_bytes(msgpack.packb({"keyframes": {}}))
    descs, ids, offsets = ovs_io.load_keyframe_descriptors(tmp_path / "empty.msg")
    assert len(descs) == 0 and len(ids) == 0 and offsets.tolist() == [0]


@pytest.mark.gpu
def test_gpu_search_in_a_loaded_map(tmp_path):
    """A saved map becomes the descriptor database: rows taken from it are found again at distance 0
    and point back to (keyframe id, keypoint index)."""
    from openvslam_amd import _lib
    rng = np.random.default_rng(3)
    sizes = {5: 900, 6: 1100, 9: 1000, 12: 1200}
    expected = _write_map(tmp_path / "map.msg", rng, sizes)
    descs, ids, offsets = ovs_io.load_keyframe_descriptors(tmp_path / "map.msg")
    L = _lib.lib()
    h = C.c_void_p()
    _lib.check(L.ovs_desc_db_create(0, C.byref(h)))
    _lib.check(L.ovs_desc_db_upload(h, A.ptr(descs, A.u8p), len(descs), 0))
    picks = [(5, 2), (6, 1099), (9, 2), (12, 600)] + [(6, int(i)) for i in rng.integers(2, 1100, 200)]
    queries = np.ascontiguousarray(np.stack([expected[k][i] for k, i in picks]))
    out = np.zeros(len(picks), dtype=A.KNN2_DTYPE)
    _lib.check(L.ovs_desc_db_search(h, A.ptr(queries, A.u8p), len(picks), A.ptr(out, A.knn2p)))
    L.ovs_desc_db_destroy(h)
    for (kf_id, kp), res in zip(picks, out):
        assert res["dist"][0] == 0
        row = int(res["row"][0])
        k = int(np.searchsorted(offsets, row, side="right") - 1)
        assert (int(ids[k]), row - int(offsets[k])) == (kf_id, kp)

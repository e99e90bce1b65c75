"""Host-side mirror of the part of io::map_database_io (src/openvslam/io/map_database_io.cc:48-97)
that the descriptor database needs: the keyframe descriptors of a saved MessagePack map."""
import ctypes as C
import os

import numpy as np

from . import _abi as A
from ._lib import check, lib


def load_keyframe_descriptors(path):
    """Returns (descs uint8 [rows, 32], keyframe_ids int32 [k], row_offsets int64 [k + 1]); the rows of
    keyframe ids[i] are descs[row_offsets[i]:row_offsets[i + 1]], keyframes in ascending id order."""
    L = lib()
    h = C.c_void_p()
    check(L.ovs_map_open(os.fsencode(path), C.byref(h)))
    try:
        k = L.ovs_map_num_keyframes(h)
        rows = L.ovs_map_num_descriptors(h)
        ids = np.zeros(k, dtype=np.int32)
        offsets = np.zeros(k + 1, dtype=np.int64)
        descs = np.zeros((rows, 32), dtype=np.uint8)
        check(L.ovs_map_keyframe_rows(h, A.ptr(ids, A.i32p), offsets.ctypes.data_as(C.POINTER(C.c_int64))))
        check(L.ovs_map_copy_descriptors(h, A.ptr(descs, A.u8p)))
    finally:
        L.ovs_map_close(h)
    return descs, ids, offsets

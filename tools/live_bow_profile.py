#!/usr/bin/env python3
"""Where the time of one frame::compute_bow call goes at B = 1 (host buffers in and out): wall time per call and
per-kernel device time.  Run on the GPU box."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from openvslam_amd import _lib, bow  # noqa: E402
from test_oracle_bow import noisy_leaf_descriptors  # noqa: E402


def main():
    voc = bow.synthetic_vocabulary(10, 6, seed=16)
    v = bow.bow_vocabulary(voc["k"], voc["L"], voc["node_descs"], voc["child_offsets"], voc["children"], voc["word_ids"], voc["weights"])
    out = {}
    for n in (1000, 2000, 4000):
        d = noisy_leaf_descriptors(voc, n, seed=n)
        for _ in range(5):
            v.transform(d, 4)
        reps = 100
        t0 = time.perf_counter()
        for _ in range(reps):
            v.transform(d, 4)
        wall = (time.perf_counter() - t0) / reps * 1e3
        L = _lib.lib()
        L.ovs_timing_reset()
        L.ovs_timing_enable(1)
        for _ in range(10):
            v.transform(d, 4)
        L.ovs_timing_enable(0)
        out[str(n)] = {"wall_ms": round(wall, 4), "kernel_ms": {k: round(ms / cnt, 4) for k, (ms, cnt) in _lib.timing_report().items()}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

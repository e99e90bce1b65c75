#!/usr/bin/env python3
"""Where the time of one projection::match_current_and_last_frames call goes at B = 1 (host buffers in and
out): wall time per call, per-kernel device time (event pairs around every launch, which by themselves
lengthen the call), and the same for match_frame_and_landmarks.  Run on the GPU box."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openvslam_amd import _abi as A  # noqa: E402
from openvslam_amd import _lib, match, synth  # noqa: E402
from openvslam_amd.feature import orb_extractor, orb_params  # noqa: E402


def main():
    w, h = 1241, 376
    frames = synth.make_sequence(w, h, 2, seed=5, dx=3, dy=1)
    ext = orb_extractor(orb_params(max_num_keypts=2000))
    k0, d0 = ext.extract(frames[0])
    k1, d1 = ext.extract(frames[1])
    view = A.make_frame_view(k1, d1, ext.get_scale_factors(), A.make_grid(0.0, 0.0, float(w), float(h)))
    xy = np.stack([k0["x"] - 3.0, k0["y"] - 1.0], axis=1).astype(np.float32)
    q = A.make_proj_queries(xy, k0["octave"], d0, angle=k0["angle"])
    L = _lib.lib()
    out = {}
    for name, call in (("current_and_last", lambda: match.projection(0.9, True).match_current_and_last_frames(view, q, 20.0)),
                       ("frame_and_landmarks", lambda: match.projection(0.8, True).match_frame_and_landmarks(view, q, 5.0))):
        for _ in range(5):
            call()
        reps = 200
        t0 = time.perf_counter()
        for _ in range(reps):
            res = call()
        wall = (time.perf_counter() - t0) / reps * 1e3
        L.ovs_timing_reset()
        L.ovs_timing_enable(1)
        for _ in range(20):
            call()
        L.ovs_timing_enable(0)
        rep = {k: round(v[0] / 20, 4) for k, v in _lib.timing_report().items()}
        out[name] = {"wall_ms": round(wall, 4), "kernel_ms": rep, "matches": int(res[0])}
    print(json.dumps(out))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generates tests/golden/*.npz: outputs of the CPU oracle on small seeded inputs.

The reference cannot be built in this environment (it needs OpenCV, Eigen, yaml-cpp, g2o, DBoW2)
and its own tests hold no golden keypoints or descriptors, so these fixtures are NOT reference
outputs: they pin the oracle (and through it the device code) against regressions.  The
known-answer vectors the reference's tests do hold are restated in
tests/test_oracle_known_answers.py.

    python tools/gen_golden.py
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import oracle  # noqa: E402
from openvslam_amd import synth  # noqa: E402


def main():
    out = ROOT / "tests" / "golden"
    frames = synth.make_sequence(320, 240, 2, seed=20240607, dx=3, dy=1)
    ext = oracle.OrbExtractor(max_num_keypts=400)
    k0, d0 = ext.extract(frames[0])
    pyr3 = ext.pyramid_level(3)
    blur2 = ext.blurred_level(2)
    k1, d1 = ext.extract(frames[1])
    np.savez_compressed(out / "extract_320x240.npz", frame0=frames[0], frame1=frames[1], kps0=k0, desc0=d0, kps1=k1, desc1=d1,
                        pyramid_level3=pyr3, blurred_level2=blur2)
    print("extract fixture:", len(k0), len(k1), "keypoints")


if __name__ == "__main__":
    main()

"""Randomised parity run of the extractor against the CPU oracle (GPU box): image sizes, level counts,
scale factors, quotas, thresholds, image masks and rectangle masks drawn from a seeded generator.
usage: PYTHONPATH=. python3 tools/fuzz_extract.py [cases] [seed]"""
import sys

import numpy as np

sys.path.insert(0, "tests")
import oracle  # noqa: E402
from openvslam_amd import synth  # noqa: E402
from openvslam_amd.feature import orb_extractor, orb_params  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for k in range(cases):
        levels = int(rng.integers(1, 9))
        scale = float(rng.choice([1.1, 1.2, 1.2, 1.25, 1.5, 1.9, 2.0]))
        smallest = 60.0 * scale ** (levels - 1)
        if smallest > 700:
            levels = max(1, levels - 4)
            smallest = 60.0 * scale ** (levels - 1)
        w = int(rng.integers(int(smallest) + 8, max(int(smallest) + 9, 1400)))
        h = int(rng.integers(int(smallest) + 8, max(int(smallest) + 9, 800)))
        n = int(rng.choice([10, 100, 500, 1000, 2000, 3000]))
        ini = int(rng.choice([20, 20, 40, 7, 5, 100]))
        mn = int(rng.choice([7, 7, 3, 1, 20]))
        mn = min(mn, ini)
        rects = []
        if rng.random() < 0.3:
            a, b = sorted(rng.uniform(0, 1, 2))
            c, d = sorted(rng.uniform(0, 1, 2))
            if b - a > 0.02 and d - c > 0.02:
                rects = [[float(a), float(b), float(c), float(d)]]
        kind = rng.integers(0, 3)
        if kind == 0:
            img = synth.make_frame(w, h, seed=int(rng.integers(1 << 30)))
        elif kind == 1:
            img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        else:
            img = np.clip(128 + rng.normal(0, 6, (h, w)), 0, 255).astype(np.uint8)
            img[h // 3:h // 2, w // 4:w // 2] += 40
        mask = None
        if rng.random() < 0.3:
            mask = np.full((h, w), 255, np.uint8)
            mask[:, : w // 3] = 0 if rng.random() < 0.5 else 255
            mask[h // 2:, w // 2:] = 0
        params = dict(max_num_keypts=n, scale_factor=scale, num_levels=levels, ini_fast_thr=ini, min_fast_thr=mn, mask_rects=rects)
        tag = "case %d: %dx%d levels %d scale %.2f kp %d thr %d/%d rects %d mask %s kind %d" % (k, w, h, levels, scale, n, ini, mn, len(rects), mask is not None, kind)
        try:
            ref = oracle.OrbExtractor(**params)
            rk, rd = ref.extract(img, mask)
        except Exception as e:  # parameters the oracle refuses must be refused by the device path too
            try:
                orb_extractor(orb_params(**params)).extract(img, mask)
                print("MISMATCH (oracle refused, device accepted)", tag, e)
                bad += 1
            except Exception:
                print("both refuse", tag)
            continue
        try:
            ext = orb_extractor(orb_params(**params))
            gk, gd = ext.extract(img, mask)
        except Exception as e:
            if "quota per level too large" in str(e):
                print("limit (per-level quota beyond the quadtree's LDS node arrays)", tag)
                continue
            print("MISMATCH (device refused)", tag, e)
            bad += 1
            continue
        ok = len(gk) == len(rk) and all(np.array_equal(gk[f], rk[f]) for f in ("octave", "x", "y", "response", "size", "class_id")) \
            and np.array_equal(gk["angle"].view(np.uint32), rk["angle"].view(np.uint32)) and np.array_equal(gd, rd)
        print("ok " if ok else "MISMATCH", tag, "->", len(gk), "keypoints")
        bad += 0 if ok else 1
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

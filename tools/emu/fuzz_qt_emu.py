#!/usr/bin/env python3
"""Randomised run of the emulated quadtree kernel against the oracle (CPU only).

    fuzz_qt_emu.py [--cases 100] [--seed 0] [--hip FILE --patched]

Draws image sizes (landscape, portrait, near-square, barely larger than the border), level counts, scale
factors, quotas from a handful to thousands, FAST thresholds and four kinds of image content, and runs
tools/emu/run_qt_emu.py for each; every third case with the threads of the workgroup in reverse or
shuffled order.  Prints the failing command lines.
"""
import argparse
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--hip", default=None)
    ap.add_argument("--patched", action="store_true")
    ap.add_argument("--work", default="/tmp/qt_emu_fuzz")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    failed = []
    for case in range(a.cases):
        kind = ["synth", "noise", "faint", "sparse"][int(rng.integers(0, 4))]
        shape = int(rng.integers(0, 4))
        if shape == 0:
            w, h = int(rng.integers(200, 1400)), int(rng.integers(120, 500))
        elif shape == 1:
            w, h = int(rng.integers(120, 500)), int(rng.integers(200, 1400))
        elif shape == 2:
            s = int(rng.integers(100, 900))
            w, h = s, s + int(rng.integers(-3, 4))
        else:
            w, h = int(rng.integers(60, 200)), int(rng.integers(60, 200))
        scale = float(np.round(rng.uniform(1.05, 2.0), 2))
        # every level must stay larger than the 2 x 19 px border
        max_levels = 1
        while max_levels < 12 and min(w, h) / scale ** max_levels > 45:
            max_levels += 1
        levels = int(rng.integers(1, max_levels + 1))
        n = int([rng.integers(1, 20), rng.integers(20, 500), rng.integers(500, 3000), rng.integers(3000, 7000)][int(rng.integers(0, 4))])
        ini = int(rng.integers(5, 60))
        mn = int(rng.integers(1, ini + 1))
        cmd = [sys.executable, os.path.join(HERE, "run_qt_emu.py"), "--work", a.work, "--size", "%dx%d" % (w, h),
               "--levels", str(levels), "--scale", str(scale), "--n", str(n), "--image", kind, "--seed", str(case),
               "--ini-thr", str(ini), "--min-thr", str(mn)]
        if case > 0:
            cmd.append("--no-build")
        if a.hip:
            cmd += ["--hip", a.hip]
        if a.patched:
            cmd += ["--patched", "--strict-lds"]
        env = dict(os.environ)
        if case % 3 == 1:
            env["EMU_REVERSE"] = "1"
        elif case % 3 == 2:
            env["EMU_SHUFFLE"] = str(case)
        r = subprocess.run(cmd, capture_output=True, text=True, env=env)
        ok = r.returncode == 0 and r.stdout.strip().endswith("OK")
        print("%3d %s %s %dx%d levels %d scale %.2f n %d thr %d/%d" % (case, "ok  " if ok else "FAIL", kind, w, h, levels, scale, n, ini, mn), flush=True)
        if not ok:
            failed.append(" ".join(cmd))
            print(r.stdout[-1500:] + r.stderr[-1500:])
    print("%d of %d cases failed" % (len(failed), a.cases))
    for f in failed:
        print(f)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())

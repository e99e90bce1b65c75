#!/usr/bin/env python3
"""Runs the quadtree kernel of ovs_orb.hip on the host-side SIMT emulator and checks its output against
the CPU oracle: candidates per level come from the oracle's FAST stage, the expected keypoints from
its full extraction.  No GPU is involved: this checks the kernel's index arithmetic and data flow (every
buffer sits in guarded memory and starts filled with a chosen byte), not its timing or memory-model
behaviour.

    run_qt_emu.py [--hip FILE] [--patched] [--fill 0xFF] [--levels 8] [--n 2000] [--size 1241x376]
                  [--scale 1.2] [--image noise|synth|faint|sparse] [--seed 1] [--strict-lds] [--no-build]

--hip      kernel source to cut the quadtree from (default: openvslam_amd/csrc/ovs_orb.hip)
--patched  the source has the NODES_LDS template parameter: levels whose node arrays do not fit in LDS run
           the instance that keeps them in a global workspace (detected from the source when not given)
"""
import argparse
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

import oracle  # noqa: E402
from openvslam_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hip", default=os.path.join(ROOT, "openvslam_amd", "csrc", "ovs_orb.hip"))
    ap.add_argument("--patched", action="store_true")
    ap.add_argument("--fill", default="0xFF")
    ap.add_argument("--levels", type=int, default=8)
    ap.add_argument("--n", type=int, default=2000)
    ap.add_argument("--size", default="1241x376")
    ap.add_argument("--scale", type=float, default=1.2)
    ap.add_argument("--image", default="noise")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--work", default="/tmp/qt_emu")
    ap.add_argument("--no-build", action="store_true", help="reuse the emulator binary in the work directory")
    ap.add_argument("--ini-thr", type=int, default=20)
    ap.add_argument("--min-thr", type=int, default=7)
    ap.add_argument("--strict-lds", action="store_true", help="a read past the LDS allocation is an error (the hardware returns 0)")
    a = ap.parse_args()
    w, h = (int(v) for v in a.size.split("x"))
    with open(a.hip) as f:
        a.patched = a.patched or "template <bool NODES_LDS>" in f.read()
    os.makedirs(a.work, exist_ok=True)
    here = os.path.dirname(os.path.abspath(__file__))

    inc = os.path.join(a.work, "qt_kernel.inc")
    exe = os.path.join(a.work, "qt_emu_patched" if a.patched else "qt_emu")
    if not (a.no_build and os.path.exists(exe)):
        subprocess.run([sys.executable, os.path.join(here, "gen_quadtree_emu.py"), a.hip, inc], check=True)
        cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-rdynamic", "-Wno-unknown-pragmas",
               "-I", a.work, "-I", here, os.path.join(here, "qt_emu_main.cc"), "-o", exe]
        if a.patched:
            cmd.insert(1, "-DQT_PATCHED")
        subprocess.run(cmd, check=True)

    if a.image == "noise":
        img = np.random.default_rng(a.seed).integers(0, 256, (h, w), dtype=np.uint8)
    elif a.image == "faint":
        # low contrast: most cells only answer at the minimum threshold, many stay empty
        rng = np.random.default_rng(a.seed)
        img = np.clip(128 + rng.normal(0, 3.5, (h, w)), 0, 255).astype(np.uint8)
    elif a.image == "sparse":
        # a handful of isolated corners: levels with fewer candidates than initial nodes
        rng = np.random.default_rng(a.seed)
        img = np.full((h, w), 100, np.uint8)
        for _ in range(int(rng.integers(1, 12))):
            y, x = int(rng.integers(0, h - 8)), int(rng.integers(0, w - 8))
            img[y:y + 8, x:x + 8] = 220
    else:
        img = synth.make_frame(w, h, seed=a.seed)
    ref = oracle.OrbExtractor(max_num_keypts=a.n, num_levels=a.levels, scale_factor=a.scale,
                              ini_fast_thr=a.ini_thr, min_fast_thr=a.min_thr)
    rk, _ = ref.extract(img)
    quotas = ref.num_keypts_per_level()
    words = [a.levels]
    for lvl in range(a.levels):
        lw, lh = (w, h) if lvl == 0 else ref.pyramid_level(lvl).shape[::-1]
        c = ref.candidates(lvl)
        words += [lw, lh, int(quotas[lvl]), len(c)]
        rec = np.stack([c["x"].astype(np.int32), c["y"].astype(np.int32), c["response"].astype(np.int32)], axis=1)
        words += rec.reshape(-1).tolist()
        print("level %d: %d x %d, quota %d, %d candidates" % (lvl, lw, lh, quotas[lvl], len(c)))
    fin, fout = os.path.join(a.work, "in.bin"), os.path.join(a.work, "out.bin")
    np.asarray(words, dtype=np.int32).tofile(fin)
    r = subprocess.run([exe, fin, fout, a.fill] + (["strict"] if a.strict_lds else []))
    if r.returncode != 0:
        print("emulator exit code", r.returncode)
        return 2 if r.returncode == 2 else 1  # 2: the guarded address space could not be reserved

    out = np.fromfile(fout, dtype=np.int32)
    pos, bad = 0, 0
    for lvl in range(a.levels):
        n = int(out[pos])
        pos += 1
        xy, resp = out[pos:pos + 2 * n:2], out[pos + 1:pos + 2 * n:2]
        pos += 2 * n
        exp = rk[rk["octave"] == lvl]
        if n != len(exp):
            print("level %d: %d keypoints, oracle %d" % (lvl, n, len(exp)))
            bad += 1
            continue
        # orb_extractor.cc:380-389 and :598-606: pt += 19, then pt *= scale factor of the level (fp32)
        sf = np.float32(1.0)
        for _ in range(lvl):
            sf = np.float32(sf * np.float32(a.scale))
        x = (xy & 0xFFFF).astype(np.float32) + np.float32(19)
        y = ((xy >> 16) & 0xFFFF).astype(np.float32) + np.float32(19)
        if lvl > 0:
            x, y = x * sf, y * sf
        same = np.array_equal(x, exp["x"]) and np.array_equal(y, exp["y"]) and np.array_equal(resp.astype(np.float32), exp["response"])
        print("level %d: %d keypoints, %s" % (lvl, n, "identical to the oracle's" if same else "DIFFERENT from the oracle's"))
        bad += 0 if same else 1
    print("OK" if bad == 0 else "%d levels differ" % bad)
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())

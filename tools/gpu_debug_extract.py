"""Stage-by-stage comparison of the device extractor with the oracle (diagnostics)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import numpy as np
import oracle
from openvslam_amd import synth
from openvslam_amd.feature import orb_extractor, orb_params

w, h, n = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (1241, 376, 2000)
img = synth.make_frame(w, h, seed=1234 + w)
ref = oracle.OrbExtractor(max_num_keypts=n)
rk, rd = ref.extract(img)
ext = orb_extractor(orb_params(max_num_keypts=n))
gk, gd = ext.extract(img)
print("counts", len(gk), len(rk))
for lvl in range(1, 8):
    a, b = ext.image_pyramid(lvl), ref.pyramid_level(lvl)
    print("pyr", lvl, a.shape, b.shape, "diff px", int((a != b).sum()) if a.shape == b.shape else "shape")
for lvl in range(8):
    a, b = ext.blurred_level(lvl), ref.blurred_level(lvl)
    if b is not None:
        print("blur", lvl, "diff px", int((a != b).sum()))
for lvl in range(8):
    g = gk[gk["octave"] == lvl]; r = rk[rk["octave"] == lvl]
    same = len(g) == len(r) and np.array_equal(g["x"], r["x"]) and np.array_equal(g["y"], r["y"])
    print("level", lvl, len(g), len(r), "same order" if same else "DIFF",
          "set equal" if set(zip(g["x"].tolist(), g["y"].tolist())) == set(zip(r["x"].tolist(), r["y"].tolist())) else "set differs")
if len(gk) == len(rk):
    print("angle diff", int((gk["angle"].view(np.uint32) != rk["angle"].view(np.uint32)).sum()),
          "resp diff", int((gk["response"] != rk["response"]).sum()),
          "desc rows diff", int((gd != rd).any(axis=1).sum()))
a, b = ext.image_pyramid(1), ref.pyramid_level(1)
ys, xs = np.nonzero(a != b)
print("diff rows", np.unique(ys)[:20], "n rows", len(np.unique(ys)))
print("diff cols", np.unique(xs)[:20], "n cols", len(np.unique(xs)))
print("sample", [(int(y), int(x), int(a[y, x]), int(b[y, x])) for y, x in list(zip(ys, xs))[:10]])

"""The 256 x 4 BRIEF sampling offsets (src/openvslam/feature/orb_point_pairs.h) are data the
device code and the oracle both carry as generated headers; tools/gen_orb_pattern.py records the
SHA-256 of the int8 table.  Both headers must hold that table."""
import hashlib
import re
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def table_of(path):
    text = path.read_text()
    body = text[text.index("] = {") + 5:text.rindex("}")]
    vals = [int(v) for v in re.findall(r"-?\d+", body)]
    return np.array(vals, dtype=np.int8)


def test_pattern_tables_match_the_recorded_hash():
    want = (ROOT / "tests" / "golden" / "orb_pattern_sha256.txt").read_text().split()[0]
    for p in (ROOT / "openvslam_amd" / "csrc" / "orb_pattern.h", ROOT / "oracle" / "ovo_orb_pattern.h"):
        t = table_of(p)
        assert t.size == 256 * 4, (p, t.size)
        assert np.abs(t.astype(int)).max() <= 13          # the 31x31 patch of the descriptor
        assert hashlib.sha256(t.tobytes()).hexdigest() == want, p

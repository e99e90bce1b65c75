"""The device code sorts the 30 rotation-histogram bins with a port of libstdc++'s std::sort
(introsort + final insertion sort), because angle_checker::get_valid_matches
(src/openvslam/match/angle_checker.h:86-114) sorts with a comparator that has ties and the
outcome on ties is implementation-defined.  tests/cpp/sort_port_check.cc compares the port,
compiled for the host, against the std::sort of this toolchain."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_sort_port_equals_libstdcxx_sort(tmp_path):
    exe = tmp_path / "sort_port_check"
    subprocess.run(["g++", "-O2", "-std=c++17", str(ROOT / "tests" / "cpp" / "sort_port_check.cc"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=120)
    assert r.stdout.startswith("OK"), r.stdout

"""The header-only C++ layer (openvslam_amd/cpp/ovs_shim.hpp) compiles as C++14 against stand-ins
for the caller's frame / landmark / matrix types (tests/cpp/shim_check.cc), links with the C-ABI
library, fails loudly without a GPU and, with one, runs extract + projection + brute-force matching
through the reference's class names."""
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def build(tmp_path):
    exe = tmp_path / "shim_check"
    lib = ROOT / "openvslam_amd" / "lib"
    subprocess.run(["g++", "-std=c++14", "-pthread", "-Wall", "-Wextra", "-Werror", "-I", str(ROOT / "include"), "-I", str(ROOT / "openvslam_amd" / "cpp"),
                    str(ROOT / "tests" / "cpp" / "shim_check.cc"), "-o", str(exe), "-L", str(lib), "-lovs_hip", "-Wl,-rpath," + str(lib)],
                   check=True)
    return exe


def test_shim_compiles_links_and_fails_loudly_without_gpu(tmp_path):
    import torch
    exe = build(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    if not torch.cuda.is_available():
        assert "OK no-gpu" in r.stdout and "no HIP device" in r.stdout


@pytest.mark.gpu
def test_shim_extract_and_match_on_gpu(tmp_path):
    exe = build(tmp_path)
    r = subprocess.run([str(exe), "run"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("OK keypts")


@pytest.mark.gpu
def test_shim_from_four_threads_on_gpu(tmp_path):
    # system.cc:126-127, data/frame.cc:64-67: extractors and matchers from several std::threads at once
    exe = build(tmp_path)
    r = subprocess.run([str(exe), "threads"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.startswith("OK threads")

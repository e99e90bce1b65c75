"""The quadtree kernel's own source on a host-side SIMT emulator, against the oracle (no GPU).

tools/emu cuts k_quadtree out of openvslam_amd/csrc/ovs_orb.hip, compiles it with g++ against an
emulator of one workgroup (a fiber per thread, barriers, ballot / readlane / wave scan) and runs it on the
candidates the oracle's FAST stage finds; the keypoints it keeps must be the oracle's, level by level, in
the same order.  Every buffer the kernel indexes sits in guarded memory that starts filled with 0xFF, so an
access through a stale or wild index ends the run, and so does a read past the LDS allocation (the
hardware would return 0; --strict-lds).  Levels whose quota needs more node records than 160 KB of LDS hold
run the kernel instance that keeps them in a global workspace, as the host code splits the launch.  This covers the kernel's index arithmetic and data flow (orb_extractor.cc:398-590 as the kernel
restates it), not its timing: the GPU parity tests stay the gate for the product.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = os.path.join(ROOT, "tools", "emu", "run_qt_emu.py")


def _run(tmp_path, *args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, RUN, "--work", str(tmp_path), "--strict-lds"] + list(args), capture_output=True, text=True, env=e, timeout=600)
    if r.returncode == 2:
        pytest.skip("the emulator could not reserve its guarded address space here: " + r.stderr[-200:])
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.strip().endswith("OK")


@pytest.mark.parametrize("args", [
    ("--size", "640x480", "--n", "1000", "--image", "synth", "--seed", "3"),       # sweeps + fill-up phase
    ("--size", "1241x376", "--n", "2000", "--image", "synth", "--seed", "1"),      # three initial nodes
    ("--size", "376x1241", "--n", "1500", "--image", "synth", "--seed", "2"),      # portrait: initial nodes stacked
    ("--size", "640x480", "--n", "4000", "--levels", "4", "--image", "noise", "--seed", "4"),  # ~19 k candidates, large quota
    ("--size", "320x240", "--n", "3000", "--levels", "3", "--image", "synth", "--seed", "5"),  # fewer candidates than the quota
    # example/tum_rgbd/TUM_RGBD_rgbd_4.yaml:33: four levels in the workspace instance, four in LDS
    ("--size", "640x480", "--n", "10000", "--image", "noise", "--seed", "7"),
    ("--size", "1241x376", "--n", "3000", "--levels", "1", "--image", "noise", "--seed", "8"),  # one level, workspace only
])
def test_quadtree_kernel_source_matches_oracle(tmp_path, args):
    _run(tmp_path, *args)


@pytest.mark.parametrize("env", [{"EMU_REVERSE": "1"}, {"EMU_SHUFFLE": "7"}])
def test_quadtree_result_does_not_depend_on_thread_order(tmp_path, env):
    # threads of the workgroup run in reverse / in a new random order between synchronisation points
    _run(tmp_path, "--size", "752x480", "--n", "1000", "--image", "synth", "--seed", "6", env=env)

#!/usr/bin/env python3
"""Builds translation units of the HIP library for the host, against tools/emu/hip_host (a stand-in for the
HIP runtime that runs kernels as fibers in guarded memory): the library's own host code and kernels, no GPU.

    build_host_lib.py [--out FILE.so] [--work DIR] [unit.hip ...]      (default units: the matcher entry points)

The only changes to the sources are mechanical: `extern __shared__ T name[];` (dynamic LDS has no host
spelling) becomes `T* name = emu::dyn_lds<T>();` on the same line, and the register class of the empty
`asm volatile("" : "+v"(x))` statements (optimisation barriers on a vector register) becomes "+r".  The compiler is
the clang++ of the ROCm installation in host mode (vector extension types), g++ if that is missing (enough for the
units without them).  The result is a test tool: nothing in
the package loads it.
"""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "openvslam_amd", "csrc")
HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_UNITS = ["ovs_common.hip", "ovs_projection.hip", "ovs_bow.hip", "ovs_pack.hip", "ovs_stereo.hip", "ovs_camera.hip", "ovs_pre.hip",
                 "ovs_mapio.hip", "ovs_match.hip", "ovs_orb.hip"]
DYN_LDS = re.compile(r"extern\s+__shared__\s+(\w+)\s+(\w+)\[\];")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def newest(paths):
    return max(os.path.getmtime(p) for p in paths)


def build(out, work, units, force=False, opt="-O1"):
    os.makedirs(work, exist_ok=True)
    srcs = [os.path.join(CSRC, u) for u in units]
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + \
        [os.path.join(HERE, "hip_host", "hip", "hip_runtime.h"), os.path.join(ROOT, "include", "ovs.h"), os.path.abspath(__file__)]
    if not force and os.path.exists(out) and os.path.getmtime(out) >= newest(deps):
        return out
    objs = []
    for u, src in zip(units, srcs):
        text = open(src).read()
        text, n = DYN_LDS.subn(lambda m: "%s* %s = emu::dyn_lds<%s>();" % (m.group(1), m.group(2), m.group(1)), text)
        text = text.replace('"+v"(', '"+r"(')
        cc = os.path.join(work, u + ".cc")
        with open(cc, "w") as f:
            f.write('#line 1 "%s"\n' % u)
            f.write(text)
        obj = os.path.join(work, u + ".o")
        cxx = [CLANG, "-x", "c++", "-Wno-unknown-attributes", "-Wno-unused-value"] if os.path.exists(CLANG) else ["g++"]
        cmd = cxx + ["-std=c++17", opt, "-g", "-fPIC", "-ffp-contract=off", "-fno-omit-frame-pointer", "-fno-strict-aliasing",
               "-Wno-unknown-pragmas", "-Wno-attributes", "-DOVS_HOST_EMULATION=1",
               "-I", os.path.join(HERE, "hip_host"), "-I", CSRC, "-c", cc, "-o", obj]
        subprocess.run(cmd, check=True)
        objs.append(obj)
    subprocess.run(["g++", "-shared", "-rdynamic", "-o", out] + objs + ["-lpthread"], check=True)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="/tmp/ovs_emu/libovs_host_emu.so")
    ap.add_argument("--work", default="/tmp/ovs_emu")
    ap.add_argument("--force", action="store_true")
    ap.add_argument("units", nargs="*", default=DEFAULT_UNITS)
    a = ap.parse_args()
    print(build(a.out, a.work, a.units, a.force))


if __name__ == "__main__":
    sys.exit(main())

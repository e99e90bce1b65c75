#!/usr/bin/env python3
"""Cuts the quadtree kernel (and the plan structures it reads) out of a copy of ovs_orb.hip into an
include file for the host-side SIMT emulator (simt_emu.h): no line of the kernel is restated by hand,
so what the emulator runs is the kernel's own source.

    gen_quadtree_emu.py <ovs_orb.hip> <out.inc>
"""
import sys


def cut(s, start, end):
    a = s.index(start)
    b = s.index(end, a)
    return s[a:b]


def main():
    src = open(sys.argv[1]).read()
    out = []
    out.append(cut(src, "constexpr int kMaxLevels = 16;", "typedef unsigned int u32;"))
    out.append(cut(src, "struct level_geom {", "struct image_src {"))
    body = cut(src, "constexpr u32 kLeaf = 0x80000000u;",
               "// -------------------------------------------------------------------------------------------------\n"
               "// Orientation + descriptor, one wavefront per keypoint.")
    # the dynamic LDS block is a guarded host allocation
    assert "    extern __shared__ u32 smem[];\n" in body
    body = body.replace("    extern __shared__ u32 smem[];\n", "    u32* smem = g_smem;\n")
    out.append(body)
    open(sys.argv[2], "w").write("\n".join(out))


if __name__ == "__main__":
    main()

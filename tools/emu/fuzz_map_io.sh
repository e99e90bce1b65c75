#!/bin/bash
# Builds and runs tools/emu/fuzz_map_io.cc (see there).  usage: fuzz_map_io.sh [iterations]
set -e
here="$(cd "$(dirname "$0")" && pwd)"
root="$(cd "$here/../.." && pwd)"
work=/tmp/ovs_emu/mapfuzz
mkdir -p "$work"
cxx=/opt/rocm/lib/llvm/bin/clang++
[ -x "$cxx" ] || cxx=g++
(cd "$root" && python3 -c "
import sys
from pathlib import Path
sys.path[:0] = ['tests', '.']
import numpy as np
import test_map_io as T
T._write_map(Path('$work/seed.msgpack'), np.random.default_rng(5), {3: 40, 0: 7, 12: 0, 9: 120})
")
for u in ovs_mapio ovs_common; do
  sed 's/"+v"(/"+r"(/' "$root/openvslam_amd/csrc/$u.hip" > "$work/$u.cc"
done
"$cxx" -x c++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -fno-omit-frame-pointer -Wno-unknown-attributes \
  -I "$here/hip_host" -I "$root/openvslam_amd/csrc" "$work/ovs_mapio.cc" "$work/ovs_common.cc" "$here/fuzz_map_io.cc" -o "$work/fuzz_map_io" -lpthread
"$work/fuzz_map_io" "$work/seed.msgpack" "$work/mutated.msgpack" "${1:-20000}"

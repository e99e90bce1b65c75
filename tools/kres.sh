#!/bin/bash
# Per-kernel register / scratch / LDS usage of one HIP source (compiler remarks), one line per kernel.
# usage: tools/kres.sh openvslam_amd/csrc/ovs_projection.hip
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -c \
  -Rpass-analysis=kernel-resource-usage "$1" -o /dev/null 2>&1 | \
python3 -c '
import re,sys,subprocess
cur=None
for line in sys.stdin:
    m=re.search(r"Function Name: (\S+)",line)
    if m:
        cur=subprocess.run(["c++filt",m.group(1)],capture_output=True,text=True).stdout.strip()
        cur=re.sub(r"\(anonymous namespace\)::","",cur).split("(")[0][:60]; vals={}
        continue
    m=re.search(r"remark:\s+([^:]*): (\S+) \[-R",line)
    if m and cur:
        vals[m.group(1)]=m.group(2)
        if m.group(1).startswith("LDS Size"):
            print("%-62s sgpr %-4s vgpr %-4s agpr %-3s scratch %-6s occ %-2s lds %s"%(cur,vals.get("TotalSGPRs"),vals.get("VGPRs"),vals.get("AGPRs"),vals.get("ScratchSize [bytes/lane]"),vals.get("Occupancy [waves/SIMD]"),vals.get("LDS Size [bytes/block]")))
'

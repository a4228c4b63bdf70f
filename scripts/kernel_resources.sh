#!/bin/bash
# register / spill / LDS table of every kernel in one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), no GPU needed.
# usage: scripts/kernel_resources.sh gan-class-transfer2_amd/csrc/tapgemm_mfma.hip
f=$1
cd "$(dirname "$f")"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Wno-inline-asm --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-result -I../../include \
  -c "$(basename "$f")" -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys, re
cur = None
rows = []
for line in sys.stdin:
    m = re.search(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: +(\S+)", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    elif cur is not None:
        cur[k] = v
for r in rows:
    print("%-100s vgpr %4s agpr %4s spill %4s scratch %5s lds %7s occ %s" % (r["name"][:100], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize"), r.get("LDS Size"), r.get("Occupancy")))
'

#!/bin/bash
# register / spill table of one HIP source:  bash tools/kres.sh kernel_sweep_wide.hip [extra flags]
cd /root/repo/bnmtf_amd/csrc
src=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -Rpass-analysis=kernel-resource-usage "$@" -c $src -o /tmp/kres.o 2>&1 | python3 -c '
import sys, re
cur = {}
for line in sys.stdin:
    m = re.search(r"remark: [^ ]+ +(Function Name|VGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\S+)", line)
    if not m: continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        if cur: print(cur)
        cur = {"fn": v[-60:]}
    else: cur[k.split()[0] + ("Spill" if "Spill" in k else "")] = v
if cur: print(cur)
'

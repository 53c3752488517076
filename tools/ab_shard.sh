#!/bin/bash
# A/B of library builds on the shard shapes (one GPU box): tools/ab_shard.sh rows libA.so libB.so ...
rows=$1; shift
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
for l in "$@" "$@"; do
  export BNMTF_LIB=$repo/$l
  rm -rf /tmp/abs; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abs -o s -- python3 $repo/tools/shard_shape_times.py $rows > /dev/null 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/abs/**/s_kernel_stats.csv",recursive=True)[0]
print("$l", "$rows", " ".join(f'{r["Name"].split("(")[0][-32:]}={float(r["AverageNs"])/1000:.1f}' for r in list(csv.DictReader(open(f)))[:3]))
PY
done

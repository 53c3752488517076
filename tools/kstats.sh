#!/bin/bash
# per-kernel summary of one bench.py run on the GPU box: tools/kstats.sh <tag> [bench args]   (writes gpurun_out/<tag>_kernel_stats.csv)
tag=$1; shift
repo=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o s -- python3 $repo/bench.py --no-cpu-baseline "$@" > $repo/gpurun_out/ks_$tag.log 2>&1
find /tmp/ks_$tag -name "*kernel_stats.csv" -exec cp {} $repo/gpurun_out/${tag}_kernel_stats.csv \;
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$repo/gpurun_out/${tag}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    n=r["Name"]; n=n[:n.index("(")] if "(" in n else n
    print(f'{n[-50:]:50s} {r["Calls"]:>6s} {float(r["AverageNs"])/1000:8.1f} us  {100*float(r["TotalDurationNs"])/tot:5.1f}%')
PY

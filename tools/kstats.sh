#!/bin/bash
# Per-kernel average durations of one workload:  bash tools/kstats.sh WORKLOAD [TAG]   -> gpurun_out/kstats_TAG.csv (+ top lines on stdout)
w=${1:-bnmtf_4096_k32}; tag=${2:-$w}
repo=$(pwd); mkdir -p $repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -o s -- python3 $repo/bench.py --workload $w --steps 40 --warmup 5 --repeats 1 --no-cpu-baseline --no-samples > /dev/null 2> /tmp/ks_$tag.err
f=$(find /tmp/ks_$tag -name "s_kernel_stats.csv" | head -1)
cp $f $repo/gpurun_out/kstats_$tag.csv
python3 - $f <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print("%-70s calls %6s avg %9.1f us  %5s%%" % (r["Name"].split("(")[0][-70:], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY

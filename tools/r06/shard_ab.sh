#!/bin/bash
# Round 6: per-rank kernel times at the shard shapes of cfg3 (rows x 8192, K = 64; tools/shard_shape_times.py) with the
# unit-per-wave sweep (default) and with the pair-layout shapes it replaces (BNMTF_UNIT=0), same box, under rocprofv3.
mkdir -p gpurun_out/r06; repo=$PWD
cd /tmp && export TMPDIR=/tmp
for rows in "$@"; do
  for v in unit pairs; do
    if [ $v = pairs ]; then export BNMTF_UNIT=0; else unset BNMTF_UNIT; fi
    rm -rf /tmp/sh_${v}_$rows
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sh_${v}_$rows -o s -- python3 $repo/tools/shard_shape_times.py $rows > /dev/null 2>&1
    f=$(find /tmp/sh_${v}_$rows -name "s_kernel_stats.csv" | head -1)
    cp $f $repo/gpurun_out/r06/shard_${rows}x8192_${v}_kernel_stats.csv
    python3 -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'sweep' in r['Name'] or 'gemm' in r['Name']: print('$v rows=$rows', r['Name'][:70], 'calls', r['Calls'], 'avg %.1f us' % (float(r['AverageNs'])/1e3))
"
  done
done 2>&1 | tee -a $repo/gpurun_out/r06/shard_ab.txt
unset BNMTF_UNIT

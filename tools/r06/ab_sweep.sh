#!/bin/bash
# Round 6: same-box A/B of sweep-kernel variants (tools/lib_w_*.so, built by tools/variant.sh) at the headline size.
# usage: bash tools/r06/ab_sweep.sh REPS name1 name2 ...   (names of tools/lib_w_<name>.so; "ship" = the shipped library)
reps=$1; shift
mkdir -p gpurun_out/r06
for r in $(seq 1 $reps); do
  for n in "$@"; do
    if [ "$n" = ship ]; then unset BNMTF_LIB; else export BNMTF_LIB=$(realpath tools/lib_w_$n.so); fi
    python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-clock --repeats 3 --min-timed-s 0.3 $AB_ARGS 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$n', 'it/s', round(d['value'],1), 'resident', round(d['device_resident']['value'],1) if d.get('device_resident') else None, {k:round(v['avg_us'],1) for k,v in d.get('kernels',{}).items()}, 'mse', [round(x,4) for x in d['mse_first_last']])"
  done
done 2>&1 | tee -a gpurun_out/r06/ab_sweep.txt

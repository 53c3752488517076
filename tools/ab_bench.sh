#!/bin/bash
# A/B of library builds on ONE GPU box (rates differ by a few per cent between boxes, so variants are only comparable
# inside one gpurun call): tools/ab_bench.sh reps libA.so libB.so ...   -- runs bench.py alternately with each library
# (selected through BNMTF_LIB; the shipped library is not touched).  Extra bench arguments: AB_ARGS="--workload ...".
reps=$1; shift
for r in $(seq 1 $reps); do
  for l in "$@"; do
    BNMTF_LIB=$(realpath "$l") python bench.py --steps 100 --warmup 10 --no-cpu-baseline --repeats 1 $AB_ARGS 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$l', round(d['value'],1), {k:round(v['avg_us'],1) for k,v in d.get('kernels',{}).items()})"
  done
done

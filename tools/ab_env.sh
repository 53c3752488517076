#!/bin/bash
# A/B of an environment switch on ONE GPU box: tools/ab_env.sh reps VAR=a VAR=b ... [-- bench args]
reps=$1; shift
vars=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vars+=("$1"); shift; done; [ "$1" == "--" ] && shift
for r in $(seq 1 $reps); do
  for v in "${vars[@]}"; do
    env $v python bench.py --steps 100 --warmup 10 --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), {k:round(x['avg_us'],1) for k,x in d['kernels'].items()})"
  done
done

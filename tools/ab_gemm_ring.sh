#!/bin/bash
# same-box A/B of the contraction's register ring (BNMTF_GEMM_RING=old: loads behind conditions; needs a library built with
# `make EXPERIMENTS=1`, the shipped one ignores the switch) -- bench lines, no profiler
for rep in 1 2; do
for r in old new; do
  for w in bnmf_8192_k64 bnmf_4096_k32 vb_8192_k64; do
    if [ $r = old ]; then export BNMTF_GEMM_RING=old; else unset BNMTF_GEMM_RING; fi
    python bench.py --workload $w --no-cpu-baseline --repeats 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$r $w', round(d['value']), 'it/s', d['roofline']['avg_launch_us'], 'us gemm')"
  done
done
done

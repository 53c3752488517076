for d in 0 1 2 3 4 6 7; do
  BNMTF_SWEEP_DBG=$d python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 > /tmp/o.json
  python -c "
import json; d=json.load(open('/tmp/o.json')); print('dbg', $d, round(d['kernels']['sweep_rows']['avg_us'],1), round(d['kernels']['sweep_cols']['avg_us'],1))"
done

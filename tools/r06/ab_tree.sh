# same-box A/B of two whole trees (library + Python + bench.py): the current one and a copy of an older commit under tools/old_r06a
for i in 1 2; do
  for v in cur old; do
    if [ $v = cur ]; then d=.; else d=tools/old_r06a; fi
    (cd $d && timeout 300 python bench.py --no-cpu-baseline 2>/dev/null) | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', round(d['value'],1), round(d['device_resident']['value'],1), r['sclk_mhz'], round(r['cycles_per_iteration']), r['kernels_us'])"
  done
done

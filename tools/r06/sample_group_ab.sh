for g in default 1 2 8 16; do
  if [ $g = default ]; then unset BNMTF_LIB; else export BNMTF_LIB=tools/lib_grp$g.so; fi
  for w in bnmf_4096_k32 bnmf_8192_k64; do
    timeout 300 python bench.py --workload $w --no-cpu-baseline --no-clock 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$g', '$w', round(d['value'],1), round(d['device_resident']['value'],1) if d.get('device_resident') else None)"
  done
done

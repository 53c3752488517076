mkdir -p gpurun_out/ab
for rep in 1 2; do
for v in new prev; do
  if [ $v = prev ]; then export BNMTF_LIB=$PWD/tools/lib_prev.so; else unset BNMTF_LIB; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/ab/head_${v}_$rep.json 2>gpurun_out/ab/err.txt
  python bench.py --workload bnmf_4096_k32 --steps 50 --warmup 10 --no-cpu-baseline > gpurun_out/ab/c2_${v}_$rep.json 2>>gpurun_out/ab/err.txt
done; done
unset BNMTF_LIB
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/ab/*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
        print(f.split('/')[-1], d['value'], r.get('sclk_mhz'), r.get('cycles_per_iteration'), r.get('cycles_per_iteration_device_resident'), r.get('kernels_us'))
    except Exception as e: print(f, 'ERR', e)
PY
timeout 900 python -m pytest tests/test_wide_sweep_gpu.py tests/test_sweep_gpu.py -m gpu -x -q 2>&1 | tail -5

# same-box A/B of two library builds (the shipped one against tools/lib_prev.so) at the shard shapes (rows x 8192, under rocprofv3) and for lone GDSC-size models on the multi-launch path
mkdir -p gpurun_out/abs; repo=$PWD
cd /tmp && export TMPDIR=/tmp
for v in new prev new prev; do
  if [ $v = prev ]; then export BNMTF_LIB=$repo/tools/lib_prev.so; else unset BNMTF_LIB; fi
  for rows in 1024 2048; do
    rm -rf /tmp/abs_${v}_$rows
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abs_${v}_$rows -o s -- python3 $repo/tools/shard_shape_times.py $rows > /dev/null 2>&1
    f=$(find /tmp/abs_${v}_$rows -name "s_kernel_stats.csv" | head -1)
    python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'sweep_chip_kernel' in r['Name'] or 'gemm' in r['Name']: print('$v rows=$rows', r['Name'][:48], 'calls', r['Calls'], 'avg %.1f us' % (float(r['AverageNs'])/1e3))
"
  done
done
unset BNMTF_LIB
cd $repo
for v in new prev; do
  if [ $v = prev ]; then export BNMTF_LIB=$repo/tools/lib_prev.so; else unset BNMTF_LIB; fi
  python3 - <<'PY'
import time, numpy as np, bnmtf_amd, os
from bnmtf_amd.synthetic import generate_bnmf, generate_bnmtf
R, M, _, _ = generate_bnmf(622, 138, 25, 0.19, seed_data=1, seed_mask=2)
np.random.seed(1)
b = bnmtf_amd.bnmf_gibbs_optimised(R, M, 25, dict(alpha=1., beta=1., lambdaU=0.1, lambdaV=0.1), seed=3, verbose=False); b.initialise('random'); b.set_small_path(False)
b.run(50, store_samples=False); t0 = time.perf_counter(); b.run(2000, store_samples=False); t1 = time.perf_counter() - t0
R, M, _, _, _ = generate_bnmtf(622, 138, 10, 10, 0.19, seed_data=1, seed_mask=2)
c = bnmtf_amd.bnmtf_gibbs_optimised(R, M, 10, 10, dict(alpha=1., beta=1., lambdaF=0.1, lambdaS=0.1, lambdaG=0.1), seed=3, verbose=False); c.initialise('random', 'random'); c.set_small_path(False)
c.run(50, store_samples=False); t0 = time.perf_counter(); c.run(2000, store_samples=False); t2 = time.perf_counter() - t0
print(os.environ.get("BNMTF_LIB", "new")[-12:], "lone 622x138 multi-launch: bnmf K=25 %.0f it/s, bnmtf K=L=10 %.0f it/s" % (2000 / t1, 2000 / t2))
PY
done
unset BNMTF_LIB
timeout 900 python -m pytest tests/test_sharded_gpu.py tests/test_bnmf_gibbs_gpu.py tests/test_icm_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3

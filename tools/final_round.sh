#!/bin/bash
# The round's closing run on the GPU box: the gpu suite, the profiles of the four large workloads, the bench lines kept under profiles/bench/.
tag=$1
mkdir -p gpurun_out/bench_$tag
timeout 1500 python -X faulthandler -m pytest tests -m gpu -q > gpurun_out/bench_$tag/gpu_suite_full.txt 2>&1      # (whole output kept: an abort's message and stacks are in it)
grep -E "passed|failed|error|Abort|Fatal|fault" gpurun_out/bench_$tag/gpu_suite_full.txt | tail -5 > gpurun_out/bench_$tag/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/bench_$tag/gpu_suite.txt 2>&1
bash tools/profile_round.sh $tag bnmf_8192_k64 bnmf_4096_k32 bnmtf_4096_k32 vb_8192_k64 bnmtf_vb_4096_k32 > gpurun_out/bench_$tag/profile.log 2>&1
python bench.py > gpurun_out/bench_$tag/${tag}_bnmf_8192_k64.json 2> gpurun_out/bench_$tag/err.txt
python bench.py --workload bnmf_4096_k32 --steps 50 > gpurun_out/bench_$tag/${tag}_bnmf_4096_k32.json 2>> gpurun_out/bench_$tag/err.txt
python bench.py --workload bnmtf_4096_k32 --steps 50 > gpurun_out/bench_$tag/${tag}_bnmtf_4096_k32.json 2>> gpurun_out/bench_$tag/err.txt
python bench.py --workload vb_8192_k64 > gpurun_out/bench_$tag/${tag}_vb_8192_k64.json 2>> gpurun_out/bench_$tag/err.txt
python bench.py --workload bnmtf_vb_4096_k32 --steps 20 --warmup 3 > gpurun_out/bench_$tag/${tag}_bnmtf_vb_4096_k32.json 2>> gpurun_out/bench_$tag/err.txt
python bench.py --workload bnmf_1024_k16 > gpurun_out/bench_$tag/${tag}_bnmf_1024_k16.json 2>> gpurun_out/bench_$tag/err.txt
bash tools/r06/shard_ab.sh 1024 2048 > gpurun_out/bench_$tag/shard_shapes.txt 2>&1
for w in bnmtf_toy_100x80_k5 bnmtf_gdsc_622x138_k5 bnmf_toy_100x80_k10 bnmf_gdsc_622x138_k25; do
  python bench.py --workload $w > gpurun_out/bench_$tag/${tag}_$w.json 2>> gpurun_out/bench_$tag/err.txt
done
python bench.py --workload cv_gdsc_bnmtf --slots 1 4 8 > gpurun_out/bench_$tag/${tag}_cv_gdsc_bnmtf.json 2>> gpurun_out/bench_$tag/err.txt
python bench.py --workload cv_gdsc > gpurun_out/bench_$tag/${tag}_cv_gdsc_slots.json 2>> gpurun_out/bench_$tag/err.txt
python bench.py --workload cv_gdsc --cv-batched --slots 1 > gpurun_out/bench_$tag/${tag}_cv_gdsc_batched.json 2>> gpurun_out/bench_$tag/err.txt
python bench.py --workload cv_gdsc_vb --slots 1 4 8 > gpurun_out/bench_$tag/${tag}_cv_gdsc_vb.json 2>> gpurun_out/bench_$tag/err.txt
python bench.py --workload cv_gdsc_vb --cv-batched --slots 1 2 4 > gpurun_out/bench_$tag/${tag}_cv_gdsc_vb_batched.json 2>> gpurun_out/bench_$tag/err.txt
rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | head -4 > gpurun_out/bench_$tag/smi.txt
cat gpurun_out/bench_$tag/gpu_suite.txt
for f in gpurun_out/bench_$tag/*.json; do python - "$f" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(sys.argv[1].split("/")[-1], round(d["value"], 1), d["unit"], r.get("sclk_mhz"), r.get("cycles_per_iteration"), r.get("frac"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done

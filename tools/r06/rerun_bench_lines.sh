tag=${1:-r06c}
mkdir -p gpurun_out/bench_${tag}2
python bench.py > gpurun_out/bench_${tag}2/${tag}_bnmf_8192_k64.json 2>/dev/null
python bench.py --workload bnmf_4096_k32 --steps 50 > gpurun_out/bench_${tag}2/${tag}_bnmf_4096_k32.json 2>/dev/null
python bench.py --workload bnmtf_4096_k32 --steps 50 > gpurun_out/bench_${tag}2/${tag}_bnmtf_4096_k32.json 2>/dev/null
python bench.py --workload vb_8192_k64 > gpurun_out/bench_${tag}2/${tag}_vb_8192_k64.json 2>/dev/null
python bench.py --workload bnmtf_vb_4096_k32 --steps 20 --warmup 3 > gpurun_out/bench_${tag}2/${tag}_bnmtf_vb_4096_k32.json 2>/dev/null
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_${tag}2/${tag}_steps20.json 2>/dev/null
for f in gpurun_out/bench_${tag}2/*.json; do python - "$f" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d.get("roofline") or {}
print(sys.argv[1].split("/")[-1], round(d["value"], 1), r.get("sclk_mhz"), r.get("cycles_per_iteration") and round(r["cycles_per_iteration"]), r.get("frac") and round(r["frac"], 3), r.get("traffic"), (d.get("cpu_baseline") or {}).get("value"))
PY
done

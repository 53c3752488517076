# L2 (TCC) hit / miss / fabric-read request counts per launch of the sweep and contraction kernels (GPU box): bash tools/l2_counters.sh
cd /tmp && export TMPDIR=/tmp
for w in bnmf_8192_k64 vb_8192_k64; do
rm -rf /tmp/l2_$w
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --output-format csv -d /tmp/l2_$w -o s -- python3 /root/repo/bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("/tmp/l2_$w/**/s_counter_collection.csv",recursive=True)
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    n=r["Kernel_Name"].split("(")[0][-40:]
    acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n,d in acc.items():
    if "sweep" in n or "gemm" in n:
        print("$w", n, {k: "%.3g"%(sum(v)/len(v)) for k,v in d.items()})
PY
done

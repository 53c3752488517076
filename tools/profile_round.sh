#!/bin/bash
# Profiles of one build on the GPU box:  bash tools/profile_round.sh TAG [workload ...]
#   * rocprofv3 --kernel-trace --stats of `bench.py --workload W` (per-kernel average durations)
#   * separate --pmc passes (never combined with the trace domains gpurun refuses): FETCH_SIZE, WRITE_SIZE (HBM traffic,
#     gfx950 correction 2 x FETCH + WRITE, MI355X_MICROARCH.md "HBM"), and two SQ passes (vector / matrix / LDS busy).
# Summaries land in gpurun_out/prof_TAG/ ; tools/summarise_profiles.py turns them into profiles/TAG_* and traffic.json.
tag=$1; shift
wls=${@:-bnmf_8192_k64}
repo=$(pwd)
out=$repo/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
md5sum $repo/bnmtf_amd/lib/libbnmtf_hip.so | cut -d' ' -f1 > $out/lib_md5.txt
for w in $wls; do
  args="$repo/bench.py --workload $w --steps 40 --warmup 5 --repeats 1 --no-cpu-baseline --no-samples --no-clock"
  rm -rf /tmp/pr_$w
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr_$w/stats -o s -- python3 $args > $out/${w}_bench_under_profiler.json 2> $out/${w}_stats.err
  cp $(find /tmp/pr_$w/stats -name "s_kernel_stats.csv" | head -1) $out/${w}_kernel_stats.csv
  for pass in "FETCH_SIZE" "WRITE_SIZE" \
              "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
              "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAVES" \
              "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
    name=$(echo $pass | cut -d' ' -f1)
    rocprofv3 --kernel-trace --pmc $pass --output-format csv -d /tmp/pr_$w/$name -o s -- python3 $repo/bench.py --workload $w --steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-samples --no-clock > /dev/null 2> $out/${w}_pmc_$name.err
    f=$(find /tmp/pr_$w/$name -name "s_counter_collection.csv" | head -1)
    python3 - "$f" "$out/${w}_pmc_$name.csv" <<'PY'
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    key = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
    acc.setdefault(key, []).append(float(r["Counter_Value"]))
with open(sys.argv[2], "w") as o:
    o.write("kernel,counter,launches,avg_per_launch\n")
    for (k, c), v in acc.items():
        o.write('"%s",%s,%d,%.6g\n' % (k, c, len(v), sum(v) / len(v)))
PY
  done
done
ls -la $out

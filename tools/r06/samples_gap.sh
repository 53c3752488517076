# kernel time per iteration with and without the sample hand-off (bnmf_4096_k32): which kernels lengthen, or do gaps appear?
cd /tmp; export TMPDIR=/tmp
for m in samples nosamples; do
  extra=""; [ $m = nosamples ] && extra="--no-samples"
  rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/gap_$m -o p -- python3 $GRAFT_REPO_ROOT/bench.py --workload bnmf_4096_k32 --no-cpu-baseline --no-clock --repeats 3 $extra > $GRAFT_REPO_ROOT/gpurun_out/gap_$m.json 2>/dev/null
done

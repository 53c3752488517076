# where the masked column Grams' time goes: variants without the products / without the packed stores (wrong results: timing only)
cd /tmp; export TMPDIR=/tmp
for v in base scol_nomfma scol_noepi scol_neither; do
  if [ $v = base ]; then unset BNMTF_LIB; else export BNMTF_LIB=$GRAFT_REPO_ROOT/tools/lib_$v.so; fi
  rm -rf /tmp/sc_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sc_$v -o p -- python3 $GRAFT_REPO_ROOT/bench.py --workload bnmtf_4096_k32 --no-cpu-baseline --no-clock --repeats 2 --no-samples > /dev/null 2>&1
  python3 -c "
import csv
for r in csv.DictReader(open('/tmp/sc_$v/p_kernel_stats.csv')):
    if 'scol_gram' in r['Name']: print('$v', round(float(r['AverageNs'])/1e3,1), 'us')
"
done

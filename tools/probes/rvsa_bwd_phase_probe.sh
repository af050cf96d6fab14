cd /tmp; export TMPDIR=/tmp
for n in 1 2 3 4 full; do
  if [ $n = full ]; then L=""; else L="MTP_HIP_LIB=$GRAFT_REPO_ROOT/tools/_abl/libmtp_hip_rvsa_stop$n.so"; fi
  env $L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st$n -o t -- python $GRAFT_REPO_ROOT/tools/bench_ops.py attn > /tmp/st$n.log 2>&1
  f=$(find /tmp/st$n -name t_kernel_stats.csv | head -1)
  python -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'rvsa_bwd5' in r['Name']: print('stop after phase $n: rvsa_bwd5 %.1f us' % (float(r['AverageNs'])/1e3))
"
done

# timing probe (round 6): rvsa_bwd5 with the bilinear gather of K_sel / V_sel replaced by a read of saved rows (tools/_abl/libmtp_hip_kselprobe.so, -DRVSA_KSEL_PROBE:
# wrong values, right traffic) against the shipped kernel, kernel times from rocprofv3, interleaved
cd /tmp; export TMPDIR=/tmp
for r in 1 2 3; do
for n in base ksel; do
  if [ $n = base ]; then L=""; else L="MTP_HIP_LIB=$GRAFT_REPO_ROOT/tools/_abl/libmtp_hip_kselprobe.so"; fi
  env $L timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks$n$r -o t -- python $GRAFT_REPO_ROOT/tools/bench_ops.py attn > /tmp/ks$n$r.log 2>&1
  f=$(find /tmp/ks$n$r -name t_kernel_stats.csv | head -1)
  python -c "
import csv
for r in csv.DictReader(open('$f')):
    if 'rvsa_bwd5' in r['Name'] or 'rvsa_fwd4' in r['Name'] or 'rvsa_scatter' in r['Name']: print('$n round $r: %-40s %.1f us' % (r['Name'][:40], float(r['AverageNs'])/1e3))
"
done
done

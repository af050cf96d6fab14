#!/bin/bash
# round 6: per-kernel statistics of the whole batch, of half a batch alone on 128 CUs, and of the two halves side by side on two CU-masked streams
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
R=$PWD; O=$PWD/gpurun_out; mkdir -p $O
i=0
for c in "whole" "half a batch alone on a 128-CU stream" "2 halves, 2 CU-masked streams (interleaved)" "half a batch alone on the whole chip"; do
  i=$((i+1)); n=case$i
  (cd /tmp && timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$n -o p -- python $R/tools/probes/two_half_batches_probe.py 1 --cu-mask "--only=$c" > $O/prof_$n.log 2>&1)
  f=$(find $O/prof_$n -name 'p_kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $O/r06_half_probe_stats_$n.csv
  grep -v "^[WE]2026" $O/prof_$n.log | head -3
  rm -rf $O/prof_$n
done
ls -la $O

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04b; mkdir -p $O
export TMPDIR=/tmp
C2=$((256 + (1 << 22)))
timeout 600 python -m pytest tests/test_hip_ops.py -q -k "p8 or gemm_nt" --timeout 600 2>&1 | tail -8 > $O/pytest_c2.log
cat $O/pytest_c2.log
MTP_AB_ROTATE=8 timeout 420 python tools/ab_gemm.py 3 512 $C2 > $O/ab_gemm_c2.txt 2>&1
cat $O/ab_gemm_c2.txt
timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
MTP_NT_VARIANT=$((1 << 22)) timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err
cut -c1-260 $O/bench_default.json $O/bench_c2.json; tail -3 $O/bench_c2.err

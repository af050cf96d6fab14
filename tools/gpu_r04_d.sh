#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r04d; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_ops.py -q -k "p8 or gemm_nt" --timeout 600 2>&1 | tail -8 | tee $O/pytest.log
MTP_AB_ROTATE=8 timeout 420 python tools/ab_gemm.py 3 $((1 << 23)) 0 2>&1 | grep -v amdgpu.ids | tee $O/ab_gemm_split.txt
MTP_NT_VARIANT=$((1 << 23)) timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_nosplit.json 2> $O/bench.err
timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_split.json 2>> $O/bench.err
MTP_NT_VARIANT=$((1 << 23)) timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_nosplit2.json 2>> $O/bench.err
timeout 300 python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_split2.json 2>> $O/bench.err
cut -c1-200 $O/bench_nosplit.json $O/bench_split.json $O/bench_nosplit2.json $O/bench_split2.json; tail -3 $O/bench.err

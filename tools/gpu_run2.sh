#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_hip_backbone.py -m gpu -q -k "fused_adamw or small_model_forward" --timeout 600 2>&1 | tail -15 > gpurun_out/r3b_pytest.log
timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r3b_bench_fused.json 2> gpurun_out/r3b_bench_fused.err
MTP_FUSED_ADAMW=0 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r3b_bench_unfused.json 2> gpurun_out/r3b_bench_unfused.err
cd /tmp && MTP_NT_STREAMK=1 timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r3b_trace_sk" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-timer > "$GRAFT_REPO_ROOT/gpurun_out/r3b_trace.log" 2>&1
cd "$GRAFT_REPO_ROOT"; tail -5 gpurun_out/r3b_pytest.log; cat gpurun_out/r3b_bench_fused.json gpurun_out/r3b_bench_unfused.json | cut -c1-300

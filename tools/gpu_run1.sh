#!/bin/bash
# round-3 GPU call 1: the whole -m gpu suite (new: stream-K NT GEMM, rs_ag / bf16 comm modes, f13 at 1e-3, InternImage-XL 512^2),
# the NT GEMM A/B incl. the stream-K form, the headline bench with and without it, a per-dispatch kernel trace
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --timeout 600 2>&1 | tail -40 > gpurun_out/r3a_pytest.log
timeout 300 python tools/ab_gemm.py 5 1024 512 $((512+131072)) > gpurun_out/r3a_ab_gemm.txt 2>&1
timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r3a_bench_base.json 2> gpurun_out/r3a_bench_base.err
MTP_NT_STREAMK=1 timeout 300 python bench.py --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/r3a_bench_sk.json 2> gpurun_out/r3a_bench_sk.err
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/r3a_trace" -o t -- python "$GRAFT_REPO_ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-gemm-timer > "$GRAFT_REPO_ROOT/gpurun_out/r3a_trace.log" 2>&1
cd "$GRAFT_REPO_ROOT"; ls -la gpurun_out/r3a_trace* | head; tail -5 gpurun_out/r3a_pytest.log; cat gpurun_out/r3a_ab_gemm.txt | tail -20; cat gpurun_out/r3a_bench_base.json gpurun_out/r3a_bench_sk.json | cut -c1-400

#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_dcnv3.py -m gpu -q --timeout 300 2>&1 | tail -25 > gpurun_out/r3j_pytest.log
timeout 600 python tools/bench_ops.py dcnv3 > gpurun_out/r3j_dcnv3_ops.txt 2>&1
timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r3j_bench_intern.json 2> gpurun_out/r3j_bench_intern.err
MTP_DCNV3_VARIANT=2 timeout 300 python bench.py --model internimage_xl --image-size 512 --batch 8 --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/r3j_bench_intern_scatter.json 2>> gpurun_out/r3j_bench_intern.err
timeout 600 python tools/ab_gemm.py 3 -1 512 > gpurun_out/r3j_ab_gemm_hipblaslt.txt 2>&1
ls /sys/class/drm/ > gpurun_out/r3j_sysfs.txt 2>&1
for d in /sys/class/drm/card*/device; do echo $d; ls $d | head -80; ls $d/hwmon/* 2>/dev/null; cat $d/pp_dpm_sclk 2>/dev/null | head; for f in $d/hwmon/*/power1_average $d/hwmon/*/freq1_input $d/hwmon/*/power1_input; do echo $f; cat $f 2>/dev/null; done; done >> gpurun_out/r3j_sysfs.txt 2>&1
(timeout 20 /opt/rocm/bin/rocm-smi --showclocks --showpower 2>&1 | head -40) >> gpurun_out/r3j_sysfs.txt
cat gpurun_out/r3j_pytest.log; cat gpurun_out/r3j_dcnv3_ops.txt | tail -32; cut -c1-300 gpurun_out/r3j_bench_intern.json gpurun_out/r3j_bench_intern_scatter.json; tail -5 gpurun_out/r3j_bench_intern.err; tail -9 gpurun_out/r3j_ab_gemm_hipblaslt.txt

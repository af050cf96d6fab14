#!/bin/bash
# re-take the source-hash-bound part of the profile set (kernel statistics, both PMC passes, HBM table) and the bench line that quotes it, after the last kernel-source edit
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=$PWD
O=$R/gpurun_out/final_r05b
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_single -o t -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-gemm-timer --no-forward-only --wgrad-side-stream 0 > $O/trace_single.log 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-timer --no-forward-only > $O/pmc_fetch.log 2>&1
timeout -s KILL 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-gemm-timer --no-forward-only > $O/pmc_write.log 2>&1
cd $R
FETCH=$(find $O/pmc_fetch -name "p_counter_collection.csv" | head -1); WRITE=$(find $O/pmc_write -name "p_counter_collection.csv" | head -1)
python tools/pmc_hbm.py $FETCH $WRITE > profiles/r05_pmc_hbm.json
cp $(find $O/trace_single -name "t_kernel_stats.csv" | head -1) profiles/r05_rocprofv3_kernel_stats_single_stream.csv
python tools/hbm_fractions.py profiles/r05_rocprofv3_kernel_stats_single_stream.csv profiles/r05_pmc_hbm.json --json profiles/r05_hbm_fractions.json > profiles/r05_hbm_fractions.txt
cp profiles/r05_pmc_hbm.json profiles/r05_rocprofv3_kernel_stats_single_stream.csv profiles/r05_hbm_fractions.json profiles/r05_hbm_fractions.txt $O/
timeout -s KILL 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench_n1.err
rm -rf $O/trace_single $O/pmc_fetch $O/pmc_write
cut -c1-300 $O/bench_n1.json; python -c "
import json;d=json.load(open('$O/bench_n1.json'));r=d['roofline'];print(r['traffic'], str(r['traffic_source'])[:80]);print(str(r.get('hbm_bound_kernels'))[:400])"

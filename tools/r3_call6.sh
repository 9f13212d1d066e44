#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3f
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_query_program.py tests/test_gpu_milstein_general.py tests/test_gpu_parity.py tests/test_gpu_brownian_stats.py tests/test_gpu_regressions.py tests/test_gpu_adaptive_device.py -q > $OUT/pytest_some.txt 2>&1
tail -8 $OUT/pytest_some.txt
timeout 200 python tools/bench_query.py > $OUT/bench_query.txt 2>&1
cat $OUT/bench_query.txt
timeout 200 python tools/query_regress.py check tests/golden/query_kernel_r1.pt > $OUT/query_regress.txt 2>&1
tail -3 $OUT/query_regress.txt
ls $OUT

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3f
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_query_program.py tests/test_gpu_milstein_general.py tests/test_gpu_graph_auto.py tests/test_gpu_parity.py tests/test_gpu_brownian_stats.py tests/test_gpu_regressions.py tests/test_gpu_adaptive_device.py -q -x > $OUT/pytest_some.txt 2>&1
tail -8 $OUT/pytest_some.txt
timeout 200 python tools/bench_query.py > $OUT/bench_query.txt 2>&1
cat $OUT/bench_query.txt
timeout 200 python tools/query_regress.py check tests/golden/query_kernel_r1.pt > $OUT/query_regress.txt 2>&1
tail -3 $OUT/query_regress.txt
timeout 300 python tools/host_overhead_train.py > $OUT/host_overhead_train.txt 2>&1
grep "fwd+bwd" $OUT/host_overhead_train.txt
timeout 300 python bench.py --workload c3_milstein_general_gradfree_b16384_d32_m16 --steps 3 --warmup 2 --no-cpu-baseline > $OUT/bench_gf.json 2>$OUT/bench_gf.err
python -c "
import json; d=json.load(open('$OUT/bench_gf.json')); print('gf general milstein ms/solve', d['ms_per_step'], d['roofline']['launch_us'])"
ls $OUT

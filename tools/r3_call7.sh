#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3g
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -8 $OUT/pytest_gpu.txt
timeout 600 bash tools/profile_traffic.sh r3g > $OUT/traffic_summary.txt 2>&1
cp gpurun_out/traffic_r3g/traffic.json $OUT/traffic.json 2>/dev/null
cp gpurun_out/traffic_r3g/traffic.json profiles/traffic_latest.json 2>/dev/null
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc=$?"; tail -3 $OUT/bench_default.err
timeout 300 python tools/host_overhead_train.py > $OUT/host_overhead_train.txt 2>&1
grep "fwd+bwd" $OUT/host_overhead_train.txt
ls -la $OUT

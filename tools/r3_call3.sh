#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3c
mkdir -p $OUT
cd $R
timeout 300 python tools/probe_failed_capture.py > $OUT/probe_failed_capture.txt 2>&1
cat $OUT/probe_failed_capture.txt
timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -12 $OUT/pytest_gpu.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc=$?"; tail -3 $OUT/bench_default.err
timeout 300 python tools/host_overhead_train.py > $OUT/host_overhead_train.txt 2>&1
ls -la $OUT

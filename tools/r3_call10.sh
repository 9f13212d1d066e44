#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3k
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
tail -6 $OUT/pytest_gpu.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc=$?"; tail -2 $OUT/bench_default.err | cut -c1-200
timeout 300 python tools/host_overhead_train.py > $OUT/host_overhead_train.txt 2>&1
grep -v "amdgpu.ids\|Warning\|run_backward" $OUT/host_overhead_train.txt | cut -c1-220
find gpurun_out -size +4M -delete
ls -la $OUT

#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3d
mkdir -p $OUT
cd $R
timeout 300 python tools/probe_torch_profiler.py > $OUT/probe_torch_profiler.txt 2>&1
tail -40 $OUT/probe_torch_profiler.txt
timeout 600 python -m pytest tests/test_gpu_milstein_general.py tests/test_gpu_sharding.py tests/test_gpu_graph_auto.py -q > $OUT/pytest_some.txt 2>&1
tail -5 $OUT/pytest_some.txt
timeout 1200 bash tools/profile_traffic.sh r3d > $OUT/traffic_summary.txt 2>&1
cp gpurun_out/traffic_r3d/traffic.json $OUT/traffic.json 2>/dev/null
tail -60 $OUT/traffic_summary.txt
ls -la $OUT

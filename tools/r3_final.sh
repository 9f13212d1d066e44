#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r3r}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_some.txt 2>&1
tail -4 $OUT/pytest_some.txt
timeout 900 bash tools/profile_traffic.sh $TAG > $OUT/traffic_summary.txt 2>&1
cp gpurun_out/traffic_$TAG/traffic.json $OUT/traffic.json 2>/dev/null
cp gpurun_out/traffic_$TAG/traffic.json profiles/traffic_latest.json 2>/dev/null
rm -rf gpurun_out/traffic_$TAG
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc=$?"; tail -2 $OUT/bench_default.err | cut -c1-200
timeout 300 python tools/host_overhead_train.py > $OUT/host_overhead_train.txt 2>&1
grep -v "amdgpu.ids\|Warning\|run_backward" $OUT/host_overhead_train.txt | cut -c1-220
find gpurun_out -size +4M -delete
du -sh gpurun_out; ls -la $OUT

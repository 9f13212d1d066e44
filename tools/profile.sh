#!/bin/bash
# Kernel-trace recipe used for profiles/ (run on the GPU box through gpurun). Usage: tools/profile.sh <tag>
# (HBM traffic counters: tools/profile_traffic.sh, the one recipe for them; SQ counters of the headline: tools/profile_trajectory.sh)
# (every pass under its own `timeout`: a counter pass of round 4 did not come back and ate the rest of the GPU budget)
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also > $OUT/bench_under_trace.json 2> $OUT/trace.log
find $OUT -name "*kernel_trace.csv" -size +4M -delete
python $R/tools/profile_summary.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt

#!/bin/bash
# rocprofv3 passes for the whole-trajectory kernel (run on the GPU box through gpurun).
# Usage: tools/profile_trajectory.sh <tag> [workload]   (default: the headline; every pass under its own timeout)
# SQ counters in two passes of 8 and 6 (8 SQ slots per pass), GRBM_GUI_ACTIVE in the second: no trace domain but --kernel-trace.
set -u
TAG=${1:-r1}
WL=${2:-c2_euler_diag_default_route_b65536_d64_s1000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_traj_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline --no-also --no-stepwise"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/bench_under_trace.json 2> $OUT/trace.log
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -o bench -- $CMD > /dev/null 2> $OUT/pmc_sq.log
timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o bench -- $CMD > /dev/null 2> $OUT/pmc_sq2.log
python $R/tools/trajectory_pmc_summary.py "$OUT" "$R" "$WL" > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +4M -delete

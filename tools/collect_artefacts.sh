#!/bin/bash
# Measurements DESIGN.md argues from, kept as text (run on the GPU box through gpurun; copy the outputs to profiles/).
# Usage: tools/collect_artefacts.sh <tag>      -> gpurun_out/artefacts_<tag>/*.txt, each starting with its command line
set -u
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/artefacts_$TAG
mkdir -p $OUT
cd $R
run() {   # run <outfile> <command...>
  local out=$OUT/$1; shift
  { echo "command: $*"; echo "host: $(rocm-smi --showproductname 2>/dev/null | grep -m1 'Card Series' || echo MI355X) ; $(date -u +%FT%TZ)"; eval "$@"; } > $out 2>&1
  echo "== $out"; tail -n 3 $out
}
run kernels_size_sweep_auto.txt        "python tools/bench_kernels.py"
run kernels_size_sweep_nt_off.txt      "TSDE_FORCE_NT=0 python tools/bench_kernels.py"
run kernels_size_sweep_nt_on.txt       "TSDE_FORCE_NT=1 python tools/bench_kernels.py"
run microbench_step_zero_data.txt      "tools/microbench_step"
run microbench_step_random_data.txt    "TSDE_RANDOM=1 tools/microbench_step"
run microbench_step_1gib_random.txt    "TSDE_RANDOM=1 TSDE_ROWS=1048576 tools/microbench_step"
run c3_lds_vs_registers.txt            "tools/microbench_general"
run brownian_query_timings.txt         "python tools/bench_query.py"
run adaptive_solve_timings.txt         "python tools/bench_adaptive.py"
# HIP API + kernel statistics of one adaptive solve with the control on the device / on the host (who synchronises how often)
for mode in device host; do
  D=$OUT/adaptive_trace_$mode
  ( cd /tmp && export TMPDIR=/tmp && rocprofv3 --hip-trace --kernel-trace --stats --output-format csv -d $D -o t -- python $R/tools/prof_adaptive.py $mode > $OUT/adaptive_trace_$mode.log 2>&1 )
  { echo "command: rocprofv3 --hip-trace --kernel-trace --stats -- python tools/prof_adaptive.py $mode   (3 solves; HIP API statistics, then kernel statistics)"; tail -n 1 $OUT/adaptive_trace_$mode.log; f=$(find $D -name "*hip_api_stats.csv" | head -1); [ -n "$f" ] && head -n 14 $f; f=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-160 $f | head -n 16; } > $OUT/adaptive_trace_$mode.txt
  rm -rf $D
  echo "== $OUT/adaptive_trace_$mode.txt"; head -n 8 $OUT/adaptive_trace_$mode.txt
done

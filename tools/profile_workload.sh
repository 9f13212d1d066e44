#!/bin/bash
# rocprofv3 kernel stats for one bench workload: tools/profile_workload.sh <workload> <tag> [extra bench args]
set -u
W=$1; TAG=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python $R/bench.py --workload $W --steps 2 --warmup 1 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/trace.log
find $OUT -name "*kernel_trace.csv" -delete
python - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("== rocprofv3 --kernel-trace --stats: bench.py --workload $W $* (top kernels by total time) ==")
for r in rows[:14]:
    print(f"{r['Name'][:100]:100s} calls={r['Calls']:>7s} avg_us={float(r['AverageNs'])/1e3:8.2f} pct={float(r['TotalDurationNs'])/tot*100:5.1f}")
PY

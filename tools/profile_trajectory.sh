#!/bin/bash
# rocprofv3 passes for the whole-trajectory kernel (run on the GPU box through gpurun).
# Usage: tools/profile_trajectory.sh <tag> [workload]
set -u
TAG=${1:-r1}
WL=${2:-c2_euler_closed_form_b65536_d64_s1000}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_traj_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --workload $WL --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/bench_under_trace.json 2> $OUT/trace.log
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -o bench -- $CMD > /dev/null 2> $OUT/pmc_sq.log
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/pmc_sq2 -o bench -- $CMD > /dev/null 2> $OUT/pmc_sq2.log
python - "$OUT" <<'EOF'
import csv, glob, os, sys
out = sys.argv[1]
for sub in ("trace",):
    for p in glob.glob(os.path.join(out, sub, "**", "*kernel_stats.csv"), recursive=True):
        print("==", os.path.relpath(p, out))
        for i, row in enumerate(csv.reader(open(p))):
            if i < 8:
                print(",".join(row))
for sub in ("pmc_sq", "pmc_sq2"):
    for p in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        agg = {}
        n = {}
        for row in csv.DictReader(open(p)):
            if "trajectory_kernel" not in row.get("Kernel_Name", ""):
                continue
            key = row["Counter_Name"]
            agg[key] = agg.get(key, 0.0) + float(row["Counter_Value"])
            n[key] = n.get(key, 0) + 1
        print("==", os.path.relpath(p, out), "(trajectory_kernel, mean per launch)")
        for k in sorted(agg):
            print(f"{k},{agg[k] / n[k]:.6g},launches={n[k]}")
EOF
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +4M -delete

#!/bin/bash
# rocprofv3 kernel trace + PMC passes for the perceptron-drift TRAINING step: sampling kernel, reverse sweep, weight sums
# (run on the GPU box through gpurun).
set -u
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_mlp_training_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_mlp_training.py --no-stepwise"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/bench.txt 2> $OUT/trace.log
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc1 -o bench -- $CMD > /dev/null 2> $OUT/pmc1.log
python - "$OUT" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for i, row in enumerate(csv.reader(open(p))):
        if i < 10:
            print(",".join(row)[:220])
for p in glob.glob(os.path.join(out, "pmc1", "**", "*counter_collection.csv"), recursive=True):
    for kernel in ("mlp_backward_kernel", "gram_kernel", "mlp_trajectory_kernel"):
        agg, n = {}, {}
        for row in csv.DictReader(open(p)):
            if kernel not in row.get("Kernel_Name", ""):
                continue
            k = row["Counter_Name"]
            agg[k] = agg.get(k, 0.0) + float(row["Counter_Value"]); n[k] = n.get(k, 0) + 1
        print("==", kernel, "(mean per launch)")
        for k in sorted(agg):
            print(f"{k},{agg[k] / n[k]:.6g},launches={n[k]}")
PY
find $OUT -name "*kernel_trace.csv" -size +4M -delete
find $OUT -name "*counter_collection.csv" -size +4M -delete

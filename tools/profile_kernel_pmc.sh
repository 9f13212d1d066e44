#!/bin/bash
# SQ / GRBM counters of ONE kernel of a bench workload (run on the GPU box through gpurun): a kernel-trace pass and two
# counter passes (8 SQ slots each; GRBM_GUI_ACTIVE rides with the second), every pass under its own timeout, no trace
# domain beside --kernel-trace in the counter passes.
# Usage: tools/profile_kernel_pmc.sh <tag> <workload> <kernel name substring>
set -u
TAG=${1:-r5}
WL=${2:-c3_euler_general_default_route_b16384_d32_m16}
KERNEL=${3:-neural_trajectory_kernel}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_${TAG}_$WL
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-also --no-stepwise"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- $CMD > $OUT/bench_under_trace.json 2> $OUT/trace.log
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $OUT/pmc1 -o bench -- $CMD > /dev/null 2> $OUT/pmc1.log
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_SALU --kernel-trace --output-format csv -d $OUT/pmc2 -o bench -- $CMD > /dev/null 2> $OUT/pmc2.log
python - "$OUT" "$R" "$WL" "$KERNEL" > $OUT/summary.txt 2>&1 <<'PY'
import csv, glob, os, sys
out, root, workload, kernel = sys.argv[1:5]
sys.path.insert(0, root)
import bench
print("workload", workload, "kernel", kernel, "csrc_sha", bench.csrc_digest())
ns = None
for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for i, row in enumerate(csv.reader(open(p))):
        if i < 5:
            print(",".join(row)[:220])
        if i > 0 and kernel in row[0] and ns is None:
            ns = float(row[3])
c = {}
for sub in ("pmc1", "pmc2"):
    for p in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        agg, n = {}, {}
        for row in csv.DictReader(open(p)):
            if kernel not in row.get("Kernel_Name", ""):
                continue
            k = row["Counter_Name"]
            agg[k] = agg.get(k, 0.0) + float(row["Counter_Value"]); n[k] = n.get(k, 0) + 1
        print("==", sub, f"({kernel}, mean per launch)")
        for k in sorted(agg):
            c[k] = agg[k] / n[k]
            print(f"{k},{agg[k] / n[k]:.6g},launches={n[k]}")
if ns and "GRBM_GUI_ACTIVE" in c:
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0
    print("== derived")
    print(f"kernel_avg_us,{ns / 1e3:.1f}")
    print(f"effective_clock_ghz,{cycles / ns:.3f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
        print(f"mfma_busy (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / kernel cycles),{c['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / cycles:.3f}")
    if "SQ_ACTIVE_INST_VALU" in c:
        print(f"valu_busy (SQ_ACTIVE_INST_VALU x 4 / 1024 / kernel cycles),{c['SQ_ACTIVE_INST_VALU'] * 4.0 / 1024.0 / cycles:.3f}")
    if "SQ_INSTS_MFMA" in c and "SQ_WAVES" in c:
        print(f"mfma_instructions_per_wave,{c['SQ_INSTS_MFMA'] / c['SQ_WAVES']:.1f}")
PY
cat $OUT/summary.txt
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete

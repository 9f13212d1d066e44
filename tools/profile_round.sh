#!/bin/bash
# Everything profiles/ holds for one round, in one GPU call (run through gpurun; then copy the text / json / csv files of
# gpurun_out/round_<tag>/ into profiles/ with the tag as prefix). Usage: tools/profile_round.sh <tag>
set -u
TAG=${1:-r2}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/round_$TAG
mkdir -p $OUT
cd $R
# 1. the default bench line (what the driver runs), on its own
python bench.py > $OUT/bench_c2_default.json 2> $OUT/bench_c2_default.err
# 2. default bench under rocprofv3: kernel trace + stats
tools/profile.sh $TAG > $OUT/bench_c2_rocprofv3_summary.txt 2>&1
cp gpurun_out/prof_$TAG/bench_under_trace.json $OUT/bench_c2_under_trace.json 2>/dev/null
f=$(find gpurun_out/prof_$TAG/trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -n 12 $f | cut -c1-400 > $OUT/bench_c2_kernel_stats.csv
# 3. kernel statistics of the other BASELINE configurations (stepwise) and of the closed-form routes
: > $OUT/other_workloads_rocprofv3.txt
for W in c3_euler_general_b16384_d32_m16 c4_midpoint_diag_b32768_d64 c5_adjoint_latent_b32768_d128_s500 c2_euler_expdiff_b65536_d64_s1000 c5_adjoint_mlp_b32768_d128_s500 c2_euler_expdiff_closed_form_b65536_d64_s1000; do
  tools/profile_workload.sh $W ${TAG}_$W >> $OUT/other_workloads_rocprofv3.txt 2>&1
  python - "$R/gpurun_out/prof_${TAG}_$W/bench.json" >> $OUT/other_workloads_rocprofv3.txt <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print("   bench line under trace: value=%.4g traj-steps/s, ms_per_step=%.3f" % (d["value"], d["ms_per_step"]))
except Exception as e:
    print("   (no bench line: %s)" % e)
PY
done
# 4. matrix-core kernels: MFMA busy / LDS counters (sampling kernel; adjoint kernel)
tools/profile_mlp.sh $TAG > $OUT/mlp_sampling_kernel_rocprofv3.txt 2>&1
( cd /tmp && export TMPDIR=/tmp
  CMD="python $R/bench.py --workload c5_adjoint_mlp_b32768_d128_s500 --steps 2 --warmup 1 --no-cpu-baseline"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_adjoint -o bench -- $CMD > /dev/null 2> $OUT/pmc_adjoint.log
  python - "$OUT" > $OUT/mlp_adjoint_kernel_pmc.txt <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
print("rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY -- bench.py --workload c5_adjoint_mlp_b32768_d128_s500")
for p in glob.glob(os.path.join(out, "pmc_adjoint", "**", "*counter_collection.csv"), recursive=True):
    for kernel in ("mlp_adjoint_kernel", "gram_kernel", "mlp_trajectory_kernel"):
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
  rm -rf $OUT/pmc_adjoint )
# 5. the measurements DESIGN.md argues from, as text
tools/collect_artefacts.sh $TAG > $OUT/collect.log 2>&1
cp gpurun_out/artefacts_$TAG/*.txt $OUT/ 2>/dev/null
ls -la $OUT

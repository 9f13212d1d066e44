#!/bin/bash
# round 3, GPU call 1: full GPU suite on the new sources + the measurements items 5 / 6 start from
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3a
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -5 $OUT/pytest_gpu.txt
timeout 120 tools/microbench_shard > $OUT/microbench_shard.txt 2>&1
timeout 60 tools/microbench_rng > $OUT/microbench_rng.txt 2>&1
timeout 120 python tools/bench_query.py > $OUT/bench_query.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/shard_trace -o shard -- $R/tools/microbench_shard > /dev/null 2> $OUT/shard_trace.log
f=$(find $OUT/shard_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cut -c1-300 $f > $OUT/microbench_shard_kernel_stats.csv
rm -rf $OUT/shard_trace
for PMC in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS"; do
  tag=$(echo $PMC | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc_$tag -o q -- python $R/tools/bench_query.py > /dev/null 2> $OUT/pmc_$tag.log
done
python - "$OUT" > $OUT/query_kernel_pmc.txt <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
for p in sorted(glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    agg, n = {}, {}
    for row in csv.DictReader(open(p)):
        k = (row.get("Kernel_Name", "")[:90], row["Counter_Name"])
        if "query_kernel" not in k[0]:
            continue
        agg[k] = agg.get(k, 0.0) + float(row["Counter_Value"]); n[k] = n.get(k, 0) + 1
    for k in sorted(agg):
        print(f"{k[0]} | {k[1]} | mean per launch {agg[k] / n[k]:.6g} | launches {n[k]}")
PY
rm -rf $OUT/pmc_*/
ls -la $OUT

#!/bin/bash
# round 3, GPU call 2: full GPU suite, the new bench line, query-kernel counters, kernel sweep
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r3b
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1
echo "pytest rc=$?" >> $OUT/pytest_gpu.txt
tail -15 $OUT/pytest_gpu.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench rc=$?"; tail -3 $OUT/bench_default.err
timeout 300 python tools/bench_kernels.py > $OUT/kernels_size_sweep.txt 2>&1
timeout 300 python tools/host_overhead_train.py > $OUT/host_overhead_train.txt 2>&1
cd /tmp && export TMPDIR=/tmp
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

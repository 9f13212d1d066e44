#!/bin/bash
# SQ counters of the general-noise contraction kernel and the shared-diffusion MFMA kernel at the shapes VERDICT r3 names:
# two rocprofv3 --pmc passes (8 SQ slots each) + GRBM, each with --kernel-trace only. Usage (GPU box): tools/pmc_general.sh <tag>
set -u
TAG=${1:-r4}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_general_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
P1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"
P2="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM"
P3="GRBM_GUI_ACTIVE GRBM_COUNT"
i=0
for P in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $OUT/pass$i -o g -- python $R/tools/bench_general.py --once > $OUT/pass$i.log 2>&1
done
python - "$OUT" <<'PY'
import csv, glob, os, sys
from collections import defaultdict
out = sys.argv[1]
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for path in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r.get("Kernel_Name", "")
            if "general_rows_kernel" in k or "shared_mfma_kernel" in k or "general_fast_kernel" in k:
                key = (k.split("(")[0][:60], r.get("Grid_Size"), r.get("VGPR_Count") or r.get("Arch_VGPR_Count"))
                c = agg[key][r["Counter_Name"]]
                c[0] += 1
                c[1] += float(r["Counter_Value"])
for key in sorted(agg):
    print("==", key)
    for name, (n, tot) in sorted(agg[key].items()):
        print(f"   {name:24s} launches={n:4d} mean={tot / n:14.1f}")
PY
find $OUT -name "*.csv" -size +1M -delete

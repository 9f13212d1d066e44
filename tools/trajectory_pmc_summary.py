"""Summary of tools/profile_trajectory.sh's rocprofv3 passes: the kernel-trace statistics, the SQ / GRBM counters of the
trajectory kernel (mean per launch) and what follows from them; writes headline_pmc.json next to the summary (copied to
profiles/headline_pmc_latest.json, which bench.py attaches to its `roofline` when the kernel-source digest matches).

    python tools/trajectory_pmc_summary.py <out dir> <repo root> <workload>
"""
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
sys.path.insert(0, sys.argv[2])
import bench  # noqa: E402

workload, digest = sys.argv[3], bench.csrc_digest()
print("workload", workload, "csrc_sha", digest)
cfg = bench.WORKLOADS[workload]
kernel_ns = None
for p in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("==", os.path.relpath(p, out))
    for i, row in enumerate(csv.reader(open(p))):
        if i < 8:
            print(",".join(row))
        if i > 0 and "trajectory" in row[0] and "kernel" in row[0] and kernel_ns is None:
            kernel_ns = float(row[3])
counters = {}
for sub in ("pmc_sq", "pmc_sq2"):
    for p in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        agg, n = {}, {}
        for row in csv.DictReader(open(p)):
            if "trajectory" not in row.get("Kernel_Name", ""):
                continue
            key = row["Counter_Name"]
            agg[key] = agg.get(key, 0.0) + float(row["Counter_Value"])
            n[key] = n.get(key, 0) + 1
        print("==", os.path.relpath(p, out), "(trajectory kernel, mean per launch)")
        for k in sorted(agg):
            counters[k] = agg[k] / n[k]
            print(f"{k},{agg[k] / n[k]:.6g},launches={n[k]}")
need = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE")
if all(k in counters for k in need) and kernel_ns:
    wave_steps = counters["SQ_WAVES"] * cfg["nsteps"]
    cycles = counters["GRBM_GUI_ACTIVE"] / 8.0                  # the counter sums the 8 XCDs
    rec = {"workload": workload, "csrc_sha": digest, "files": bench.csrc_file_digests(), "kernel_avg_us": kernel_ns / 1e3,
           "counters": counters,
           "valu_instructions_per_wave_step": counters["SQ_INSTS_VALU"] / wave_steps,
           "valu_busy": counters["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cycles,
           "effective_clock_ghz": cycles / kernel_ns,
           "cycles_per_wave_step_per_simd": cycles * 1024.0 / wave_steps,
           "formulae": {"valu_busy": "SQ_ACTIVE_INST_VALU x 4 (quad-cycles) / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)",
                        "effective_clock_ghz": "(GRBM_GUI_ACTIVE / 8) / kernel duration of the --kernel-trace pass",
                        "valu_instructions_per_wave_step": "SQ_INSTS_VALU / (SQ_WAVES x solver steps)"},
           "source": "tools/profile_trajectory.sh: rocprofv3 --pmc (two SQ passes, GRBM_GUI_ACTIVE in the second), "
                     "--kernel-trace --stats in a pass of its own"}
    with open(os.path.join(out, "headline_pmc.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    print("== derived")
    for k in ("valu_instructions_per_wave_step", "valu_busy", "effective_clock_ghz", "cycles_per_wave_step_per_simd"):
        print(f"{k},{rec[k]:.4f}")

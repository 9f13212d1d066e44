"""Per-instruction VALU issue cycles from a rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace run of tools/microbench_valu.

The microbenchmark's own two clocks are not the shader clock (s_memtime ticks at a fixed rate on this stack, and wall time
assumes the nominal 2.4 GHz while the chip clocks to its power budget: 2.06-2.44 GHz across these kernels). GRBM_GUI_ACTIVE
counts the cycles the kernel really ran (summed over the 8 XCDs), so

    cycles per wave64 instruction = (GRBM_GUI_ACTIVE / 8) / (8 waves per SIMD x instructions per wave)

is in true SIMD cycles whatever the clock did. Writes profiles/valu_rates.json (what tools/valu_model.py prices the step
loop with) and prints the table.

    python tools/valu_rates_from_pmc.py <dir with mb_counter_collection.csv, mb_kernel_trace.csv> <names json of the plain run>
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, plain = sys.argv[1], sys.argv[2]
with open(plain) as fh:
    base = json.load(fh)
names = list(base["cycles"])                      # in kernel order: issue<0>, issue<1>, ...
per_wave = 8 * 8 * 2000                            # kChains x kUnroll x kIters of tools/microbench_valu.hip
waves_per_simd = base.get("waves_per_simd", 8)
dur, cyc = collections.defaultdict(list), collections.defaultdict(list)
with open(os.path.join(src, "mb_kernel_trace.csv")) as fh:
    for r in csv.DictReader(fh):
        dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
with open(os.path.join(src, "mb_counter_collection.csv")) as fh:
    for r in csv.DictReader(fh):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cyc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
out = {"device": base.get("device"), "waves_per_simd": waves_per_simd, "column": "grbm",
       "unit": "SIMD cycles per wave64 instruction (issue-bound, 8 waves per SIMD); grbm = (GRBM_GUI_ACTIVE / 8 XCDs) / "
               "instructions per SIMD, i.e. true shader cycles",
       "source": "tools/microbench_valu under rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace (tools/valu_rates_from_pmc.py)",
       "cycles": {}}
print(f"{'instruction':22s} {'cycles':>8s} {'clock GHz':>10s} {'us':>9s}   (wall @2.4 GHz of the plain run)")
for kernel in sorted(cyc, key=lambda s: int(s.split("<")[1].split(">")[0])):
    op = int(kernel.split("<")[1].split(">")[0])
    c, d = min(cyc[kernel]) / 8.0, min(dur[kernel])
    rate = c / (per_wave * waves_per_simd)
    out["cycles"][names[op]] = {"grbm": round(rate, 3), "clock_ghz": round(c / d, 3),
                                "wall_2p4ghz": base["cycles"][names[op]]["wall_2p4ghz"]}
    print(f"{names[op]:22s} {rate:8.3f} {c / d:10.3f} {d / 1e3:9.1f}   {base['cycles'][names[op]]['wall_2p4ghz']:.2f}")
with open(os.path.join(ROOT, "profiles", "valu_rates.json"), "w") as fh:
    json.dump(out, fh, indent=1)
    fh.write("\n")

"""Condenses rocprofv3 CSV output (kernel stats + FETCH_SIZE / WRITE_SIZE counter passes) into a short text
summary that is committed under profiles/."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def find(sub, pattern):
    hits = glob.glob(os.path.join(out, sub, "**", pattern), recursive=True)
    return hits[0] if hits else None


kernel_avg_us = {}
stats = find("trace", "*kernel_stats.csv")
if stats:
    print("== rocprofv3 --kernel-trace --stats (bench.py --steps 3 --warmup 1): per-kernel totals ==")
    with open(stats) as f:
        rows = [r for r in csv.DictReader(f) if "delay_kernel" not in r.get("Name", "")]   # (a bench helper, not the path)
    total_ns = sum(float(r.get("TotalDurationNs") or 0) for r in rows) or 1.0
    for r in rows:
        r["Percentage"] = f"{float(r.get('TotalDurationNs') or 0) / total_ns * 100:.2f}"
    for r in rows:
        try:
            kernel_avg_us[r.get("Name", "")] = float(r.get("AverageNs")) / 1e3
        except (TypeError, ValueError):
            pass
    for r in rows[:12]:
        print(f"{r.get('Name','')[:110]:110s} calls={r.get('Calls')} total_ns={r.get('TotalDurationNs')} "
              f"avg_ns={r.get('AverageNs')} pct={r.get('Percentage')}")
traffic = {}
for label, sub, col in (("FETCH_SIZE", "pmc_fetch", "FETCH_SIZE"), ("WRITE_SIZE", "pmc_write", "WRITE_SIZE")):
    path = find(sub, "*counter_collection.csv")
    if not path:
        print(f"== {label}: no counter file ==")
        continue
    agg = defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != col:
                continue
            k = r.get("Kernel_Name", "")
            agg[k][0] += 1
            agg[k][1] += float(r.get("Counter_Value", 0))
    print(f"== {label} per launch (raw counter units as reported by rocprofv3; KiB on this stack) ==")
    for k, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:8]:
        print(f"{k[:110]:110s} launches={n} mean={tot / max(n, 1):.1f}")
        if "tsde::" in k:
            traffic.setdefault(k, {})[label] = tot / max(n, 1)

# HBM traffic per launch of our kernels: FETCH_SIZE is reported in KiB and counts 64 B per 128-B request for wide
# coalesced reads on gfx950 (guides/MI355X_MICROARCH.md section HBM) -> x2; WRITE_SIZE in KiB as reported.
import json
out_json = {}
for k, v in traffic.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        out_json[k] = {"fetch_kib_raw": v["FETCH_SIZE"], "write_kib_raw": v["WRITE_SIZE"],
                       "traffic_bytes_per_launch": (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0,
                       "kernel_avg_us": kernel_avg_us.get(k),
                       "correction": "FETCH_SIZE x2 (gfx950 wide-read under-count), units KiB"}
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (csrc_digest: what these counters are a measurement OF)
out_json = {"csrc_sha": bench.csrc_digest(),
            "collected": "tools/profile.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes of "
                         "`bench.py --steps 1 --warmup 0`, kernel_avg_us from the --kernel-trace --stats pass",
            "kernels": out_json}
with open(os.path.join(out, "traffic.json"), "w") as f:
    json.dump(out_json, f, indent=1)
print("== traffic.json ==")
print(json.dumps(out_json, indent=1))

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
# HBM traffic (FETCH_SIZE / WRITE_SIZE counter passes) has ONE recipe: tools/profile_traffic.sh, which runs the passes per
# workload and writes profiles/traffic_latest.json stamped with the kernel-source digest (bench.py attaches it). This
# summary is the kernel trace only.
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
print(f"== kernel sources: csrc_sha {bench.csrc_digest()}; HBM traffic counters: tools/profile_traffic.sh ==")

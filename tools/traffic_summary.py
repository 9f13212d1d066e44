"""Condenses what tools/profile_traffic.sh collected into traffic.json + a text summary: per workload, for every
tsde:: kernel, launches, average duration (kernel trace) and HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE
passes. Units and correction as guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are reported in KiB,
and FETCH_SIZE counts 64 B per 128-B request of a wide coalesced read, so bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024.
The `pct` column of the kernel table leaves out bench.py's own helper kernels (delay_kernel)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
workloads = sys.argv[2:]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (csrc_digest: what these counters are a measurement OF)


def find(root, pattern):
    hits = glob.glob(os.path.join(root, "**", pattern), recursive=True)
    return hits[0] if hits else None


def counter_means(path, name):
    agg = defaultdict(lambda: [0, 0.0])
    if path:
        with open(path) as f:
            for r in csv.DictReader(f):
                if r.get("Counter_Name") == name:
                    k = r.get("Kernel_Name", "")
                    agg[k][0] += 1
                    agg[k][1] += float(r.get("Counter_Value", 0))
    return {k: (n, tot / max(n, 1)) for k, (n, tot) in agg.items()}


result = {"csrc_sha": bench.csrc_digest(), "files": bench.csrc_file_digests(),
          "collected": "tools/profile_traffic.sh: per workload, rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes "
                       "(separate) and a --kernel-trace --stats pass of `bench.py --workload W --profile-steps 100`; "
                       "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024",
          "workloads": {}}
for w in workloads:
    root = os.path.join(out, w)
    stats = find(os.path.join(root, "trace"), "*kernel_stats.csv")
    avg_us, rows = {}, []
    if stats:
        with open(stats) as f:
            rows = [r for r in csv.DictReader(f) if "delay_kernel" not in r.get("Name", "")]
        for r in rows:
            try:
                avg_us[r["Name"]] = float(r["AverageNs"]) / 1e3
            except (KeyError, TypeError, ValueError):
                pass
    total = sum(float(r.get("TotalDurationNs", 0) or 0) for r in rows) or 1.0
    print(f"== {w}: kernel trace (top 8 by total time; pct excludes bench.py's delay_kernel) ==")
    for r in sorted(rows, key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))[:8]:
        print(f"  {r['Name'][:104]:104s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs']) / 1e3:8.2f} "
              f"pct={float(r['TotalDurationNs']) / total * 100:5.1f}")
    fetch = counter_means(find(os.path.join(root, "pmc_fetch"), "*counter_collection.csv"), "FETCH_SIZE")
    write = counter_means(find(os.path.join(root, "pmc_write"), "*counter_collection.csv"), "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fetch) & set(write)):
        if "tsde::" not in k:
            continue
        kernels[k] = {"launches": fetch[k][0], "fetch_kib_raw": fetch[k][1], "write_kib_raw": write[k][1],
                      "traffic_bytes_per_launch": (2.0 * fetch[k][1] + write[k][1]) * 1024.0,
                      "kernel_avg_us": avg_us.get(k)}
        print(f"  traffic {k[:96]:96s} launches={fetch[k][0]:5d} FETCH={fetch[k][1]:10.1f} KiB WRITE={write[k][1]:10.1f} KiB "
              f"-> {kernels[k]['traffic_bytes_per_launch'] / 1e6:8.3f} MB/launch")
    result["workloads"][w] = {"kernels": kernels}
with open(os.path.join(out, "traffic.json"), "w") as f:
    json.dump(result, f, indent=1)
print("== traffic.json written, csrc", result["csrc_sha"], "==")

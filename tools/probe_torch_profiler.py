"""Does torch.profiler (roctracer / rocprofiler-sdk inside the process) report per-kernel durations for kernels launched
by a HIP-graph replay on this stack, and do they agree with rocprofv3's? Prints the tsde:: kernels of one graph-replayed
and one eagerly issued solve of two bench workloads."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from torch.profiler import ProfilerActivity, profile  # noqa: E402

import bench  # noqa: E402

dev = torch.device("cuda", 0)
for name in ("c2_euler_diag_b65536_d64_s1000", "c4_midpoint_diag_b32768_d64"):
    job = bench.Job(name, dev)
    for graph in (True, False):
        for i in range(3):
            job.solve(i, graph=graph)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            job.solve(10, graph=graph)
            torch.cuda.synchronize()
        print(f"== {name}, {'graph replay' if graph else 'eager'} ==")
        rows = [e for e in prof.key_averages() if e.device_time_total > 0]
        rows.sort(key=lambda e: -e.device_time_total)
        for e in rows[:6]:
            print(f"  {e.key[:100]:100s} calls={e.count:6d} avg_us={e.device_time_total / max(e.count, 1):8.2f}")

"""The REAL reference (dev container, CPU) on the adaptive solve tools/bench_adaptive.py times on the GPU: Milstein on
GBM over [0, 1], 4 output times, dt = 0.05, rtol 1e-3, atol 1e-4, float32, batch 65536 x 64 and 1024 x 16.

    python tools/cpu_adaptive_reference.py > profiles/r2_cpu_adaptive_reference.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden", "_ref_shim"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import torch  # noqa: E402
import torchsde  # noqa: E402  (the reference)

from workloads import problems  # noqa: E402

print("command: python tools/cpu_adaptive_reference.py")
print(f"host: {os.cpu_count()} logical CPUs, torch threads {torch.get_num_threads()}; reference torchsde.sdeint(adaptive=True), CPU")
for (B, d) in ((1024, 16), (65536, 64)):
    sde = problems.make("gbm_ito", d=d)
    y0 = torch.full((B, d), 0.1)
    ts = torch.tensor([0.0, 0.25, 0.5, 0.75, 1.0])
    bm = torchsde.BrownianInterval(t0=0.0, t1=1.0, size=(B, d), dtype=torch.float32, entropy=1)
    calls = {"n": 0}
    inner = bm.__call__

    start = time.perf_counter()
    with torch.no_grad():
        ys = torchsde.sdeint(sde, y0, ts, bm=bm, method="milstein", dt=0.05, adaptive=True, rtol=1e-3, atol=1e-4)
    elapsed = time.perf_counter() - start
    print(f"B={B} d={d} adaptive milstein on the reference (CPU): {elapsed * 1e3:10.1f} ms per solve, final mean {float(ys[-1].mean()):.6f}")

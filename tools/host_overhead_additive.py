"""Host cost of the recognised additive route at a small batch (where the kernel itself takes microseconds): interpretation of f,
the probe call of g, the batched call of g over the stage times (torch.vmap), the launch. Usage: python tools/host_overhead_additive.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import torchsde_amd  # noqa: E402
from workloads import problems  # noqa: E402

DEV = "cuda"
B, d, m, steps = 256, 8, 4, 1000
y0 = torch.full((B, d), 0.1, device=DEV)
ts = torch.tensor([0.0, 1.0], device=DEV)
for name, sde in (("ExAdditive (g of t: table over 2 x 1000 stage times)", problems.AdditiveDecay(d, m)),
                  ("constant matrix (no table over time)", problems.AdditiveShared(d, m)),
                  ("NeuralAdditive (g_net of t, hidden 8)", problems.MLPNetAdditive(d, m, hidden=8))):
    sde = sde.to(DEV)
    for options, label in ((None, "default route"), ({"trajectory_kernel": False}, "stepwise")):
        times = []
        for rep in range(8):
            bm = torchsde_amd.BrownianInterval(0.0, 1.0, size=(B, m), device=DEV, dtype=torch.float32, entropy=rep,
                                               levy_area_approximation="space-time")
            torch.cuda.synchronize()
            start = time.perf_counter()
            with torch.no_grad():
                torchsde_amd.sdeint(sde, y0, ts, bm=bm, dt=1.0 / steps, options=options)
            torch.cuda.synchronize()
            times.append((time.perf_counter() - start) * 1e3)
        print(f"{name:55s} {label:14s} {sorted(times[3:])[2]:8.2f} ms per solve (256 x 8, 1000 SRK steps)")

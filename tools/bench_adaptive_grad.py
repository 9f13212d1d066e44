"""Adaptive solve + backward with autograd recording: the host-driven loop (every attempt recorded, one synchronisation per
attempt: base_solver.py:117-142) against the replay of the accepted steps found by the device-controlled loop
(adaptive.integrate_with_grad). Usage: python tools/bench_adaptive_grad.py"""
import os
import sys
import time
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import torchsde_amd  # noqa: E402
from torchsde_amd import adaptive  # noqa: E402
from workloads import problems  # noqa: E402

DEV = "cuda"
warnings.simplefilter("ignore")
for B, d in ((1024, 16), (65536, 64)):
    sde = problems.make("gbm_ito", d=d).to(DEV)
    ts = torch.tensor([0.0, 0.25, 0.5, 0.75, 1.0], device=DEV)
    for device_control in (True, False):
        times = []
        for rep in range(6):
            y0 = torch.full((B, d), 0.1, device=DEV, requires_grad=True)
            bm = torchsde_amd.BrownianInterval(0.0, 1.0, size=(B, d), dtype=torch.float32, device=DEV, entropy=rep)
            sde.zero_grad()
            torch.cuda.synchronize()
            start = time.perf_counter()
            ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="milstein", dt=0.05, adaptive=True, rtol=1e-3, atol=1e-4,
                                     options={"device_adaptive": device_control, "hip_graph": False, "adaptive_replay": True})
            ys[-1].sum().backward()
            torch.cuda.synchronize()
            times.append((time.perf_counter() - start) * 1e3)
        label = "accepted steps replayed under autograd" if device_control else "host-driven loop, every attempt recorded"
        print(f"B={B} d={d} adaptive milstein + backward, {label:44s}: {sorted(times[1:])[2]:8.2f} ms   "
              f"{adaptive.last_stats if device_control else ''}")

"""Host cost of the recognised route on SMALL solves (where a solve is host-bound): ms per solve of the untouched GBM
module and of the reference's benchmark SDE, default route (recognise.py + one trajectory launch) against the stepwise
route replayed from a HIP graph (`trajectory_kernel=False`, hip_graph="auto") and issued eagerly."""
import os
import sys
import time

import torch
from torch import nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchsde_amd  # noqa: E402
from workloads import problems  # noqa: E402


class Benchmark(nn.Module):
    noise_type, sde_type = "diagonal", "ito"

    def f(self, t, y):
        return y

    def g(self, t, y):
        return torch.exp(-y)


def ms_per_solve(sde, B, d, steps, dt, options, n=50):
    y0 = torch.full((B, d), 0.1, device="cuda")
    ts = torch.tensor([0.0, steps * dt], device="cuda")

    def solve(i):
        bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(B, d), device="cuda", entropy=i)
        with torch.no_grad():
            return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="euler", dt=dt, options=options)
    for i in range(6):
        solve(i)
    torch.cuda.synchronize()
    start = time.perf_counter()
    for i in range(n):
        solve(100 + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - start) / n * 1e3


for name, make in (("gbm", lambda d: problems.make("gbm_ito", d=d).to("cuda")), ("benchmark", lambda d: Benchmark().to("cuda"))):
    for (B, d, steps) in ((128, 8, 16), (1024, 8, 200), (4096, 64, 200), (65536, 64, 1000)):
        dt = 2.0 ** -10
        row = [ms_per_solve(make(d), B, d, steps, dt, opt) for opt in (None, {"trajectory_kernel": False},
                                                                      {"trajectory_kernel": False, "hip_graph": False})]
        print(f"{name:10s} B={B:6d} d={d:3d} steps={steps:5d}: default route {row[0]:8.3f} ms   stepwise, graph replay "
              f"{row[1]:8.3f} ms   stepwise, eager {row[2]:8.3f} ms")

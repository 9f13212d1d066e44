"""Where the host's time goes in a small reversible-Heun training step (the sde_gan example's sizes): cProfile of 40 iterations
of sdeint_adjoint + backward on the kernel route, next to the kernels' own time.  python tools/host_overhead_rheun.py"""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import torchsde_amd  # noqa: E402
from workloads import configs  # noqa: E402

c = configs.WORKLOADS["sdegan_rheun_adjoint_default_route_b1024_d16_m3_s63"]
dev = "cuda"
sde = configs.make_problem(c["problem"], c["d"], c["m"], dev)
y0 = torch.full((c["B"], c["d"]), 0.1, device=dev, requires_grad=True)
ts = torch.arange(c["nsteps"] + 1, device=dev, dtype=torch.float32) * c["dt"]


def step(i):
    bm = torchsde_amd.BrownianInterval(t0=0.0, t1=c["nsteps"] * c["dt"], size=(c["B"], c["m"]), dtype=torch.float32, device=dev,
                                       entropy=100 + i, dt=c["dt"])
    ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=bm, method="reversible_heun", adjoint_method="adjoint_reversible_heun",
                                     dt=c["dt"])
    sde.zero_grad()
    ys.sum().backward()


for i in range(4):
    step(i)
torch.cuda.synchronize()
t = time.perf_counter()
for i in range(40):
    step(10 + i)
torch.cuda.synchronize()
print(f"wall per iteration: {(time.perf_counter() - t) / 40 * 1e3:.3f} ms")
prof = cProfile.Profile()
prof.enable()
for i in range(40):
    step(100 + i)
torch.cuda.synchronize()
prof.disable()
stats = pstats.Stats(prof)
stats.sort_stats("cumulative").print_stats(28)

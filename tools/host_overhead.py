"""Host-side cost per solver step (tiny batch => GPU time negligible) and a cProfile of the driver."""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torchsde_amd  # noqa: E402
from workloads import problems  # noqa: E402

dev = "cuda"
B, d, n, dt = 64, 64, 1000, 2.0 ** -10
ts = torch.tensor([0.0, n * dt], device=dev)
y0 = torch.full((B, d), 0.1, device=dev)


def run(prob, method, levy):
    sde = problems.make(prob, d=d).to(dev)
    bm = torchsde_amd.BrownianInterval(0.0, n * dt, size=(B, d), device=dev, dtype=torch.float32, entropy=1, dt=dt,
                                       levy_area_approximation=levy)
    with torch.no_grad():
        return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=dt)


for prob, method, levy in [("gbm_ito", "euler", "none"), ("gbm_strat", "midpoint", "none"),
                           ("gbm_ito", "milstein", "none"), ("gbm_ito", "srk", "space-time")]:
    run(prob, method, levy)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        run(prob, method, levy)
    torch.cuda.synchronize()
    print(f"{method:9s} host-bound time per solver step: {(time.perf_counter() - t) / 3 / n * 1e6:7.1f} us")

# what a bare torch op costs on this host, for scale
x = torch.ones(64, 64, device=dev)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(5000):
    x = x * 1.0001
torch.cuda.synchronize()
print(f"bare torch elementwise op: {(time.perf_counter() - t) / 5000 * 1e6:.1f} us")

pr = cProfile.Profile()
pr.enable()
run("gbm_ito", "euler", "none")
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)

"""Host-side cost per solver step in the TRAINING regime (small batch, gradients on): backprop through the solver and
sdeint_adjoint, on a latent-SDE-sized diagonal problem. Prints time per step and a cProfile of one iteration."""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torchsde_amd  # noqa: E402
from torchsde_amd import graph  # noqa: E402
from workloads import problems  # noqa: E402

dev = "cuda"
B, d, n, dt = 1024, 8, 200, 2.0 ** -8
ts = torch.tensor([0.0, n * dt], device=dev)


def iteration(fn, method, sde_type, **kw):
    sde = problems.make("mlpdiag_" + ("ito" if sde_type == "ito" else "strat"), d=d).to(dev)
    y0 = torch.full((B, d), 0.1, device=dev, requires_grad=True)
    levy = "space-time" if method == "srk" else "none"

    def go(i):
        bm = torchsde_amd.BrownianInterval(0.0, n * dt, size=(B, d), device=dev, dtype=torch.float32, entropy=i, dt=dt,
                                           levy_area_approximation=levy)
        ys = fn(sde, y0, ts, bm=bm, method=method, dt=dt, **kw)
        ys[-1].sum().backward()
    go.sde = sde
    return go


cases = [("backprop euler", torchsde_amd.sdeint, "euler", "ito", {}),
         ("backprop midpoint", torchsde_amd.sdeint, "midpoint", "stratonovich", {}),
         ("backprop reversible_heun", torchsde_amd.sdeint, "reversible_heun", "stratonovich", {}),
         ("adjoint euler/euler", torchsde_amd.sdeint_adjoint, "euler", "ito", {"adjoint_method": "euler"}),
         ("adjoint midpoint", torchsde_amd.sdeint_adjoint, "midpoint", "stratonovich", {}),
         ("adjoint reversible_heun", torchsde_amd.sdeint_adjoint, "reversible_heun", "stratonovich",
          {"adjoint_method": "adjoint_reversible_heun"})]
OFF = {"hip_graph": False}
for name, fn, method, sde_type, kw in cases:       # every launch issued eagerly (what round 2's default was)
    extra = dict(options=OFF, adjoint_options=OFF) if fn is torchsde_amd.sdeint_adjoint else dict(options=OFF)
    go = iteration(fn, method, sde_type, **extra, **kw)
    go(0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(3):
        go(1 + i)
    torch.cuda.synchronize()
    print(f"{name + ' [eager]':40s} fwd+bwd per solver step: {(time.perf_counter() - t) / 3 / n * 1e6:7.1f} us")

for name, fn, method, sde_type, kw in cases:       # NO options: the drop-in call (hip_graph = "auto")
    go = iteration(fn, method, sde_type, **kw)
    for i in range(5):                             # eager + screened, recording, first replays (on probation)
        go(i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(5):
        go(5 + i)
    torch.cuda.synchronize()
    print(f"{name + ' [no options]':40s} fwd+bwd per solver step: {(time.perf_counter() - t) / 5 / n * 1e6:7.1f} us")
    for line in graph.describe_cache(go.sde):
        print("      ", line[:200])

if len(sys.argv) > 1:
    which = sys.argv[1]
    for name, fn, method, sde_type, kw in cases:
        if name == which:
            go = iteration(fn, method, sde_type, **kw)
            go(0)
            pr = cProfile.Profile()
            pr.enable()
            go(1)
            torch.cuda.synchronize()
            pr.disable()
            pstats.Stats(pr).sort_stats("tottime").print_stats(25)

# the same adjoint cases with forward solve and backward sweep replayed as HIP graphs
for name, fn, method, sde_type, kw in cases:
    if fn is not torchsde_amd.sdeint_adjoint:
        continue
    go = iteration(fn, method, sde_type, options={"hip_graph": True}, adjoint_options={"hip_graph": True}, **kw)
    for i in range(4):                             # recording, probation
        go(20 + i)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(5):
        go(2 + i)
    torch.cuda.synchronize()
    print(f"{name + ' [hip_graph=True]':40s} fwd+bwd per solver step: {(time.perf_counter() - t) / 5 / n * 1e6:7.1f} us")

# back-propagation through the solver with the forward solve and its backward recorded as two HIP graphs
for name, fn, method, sde_type, kw in cases:
    if fn is not torchsde_amd.sdeint:
        continue
    go = iteration(fn, method, sde_type, options={"hip_graph": True}, **kw)
    go(0)
    go(1)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(5):
        go(2 + i)
    torch.cuda.synchronize()
    print(f"{name + ' [hip_graph=True]':40s} fwd+bwd per solver step: {(time.perf_counter() - t) / 5 / n * 1e6:7.1f} us")

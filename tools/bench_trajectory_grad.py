"""Differentiable whole-trajectory solve (sensitivity kernel + reductions) at the BASELINE configs[1] shape."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import torchsde_amd  # noqa: E402

dev = torch.device("cuda")
B, d, n, dt = 65536, 64, 1000, 2.0 ** -10
for method, sde_type in (("euler", "ito"), ("milstein", "ito"), ("midpoint", "stratonovich"), ("srk", "ito")):
    sde = torchsde_amd.AffineDiagonalSDE(torch.full((d,), 0.1), 0.0, torch.full((d,), 0.2), 0.0, sde_type=sde_type,
                                         dtype=torch.float32).to(dev)
    ts = torch.tensor([0.0, n * dt], device=dev)
    levy = "space-time" if method == "srk" else "none"

    def go(i):
        y0 = torch.full((B, d), 0.1, device=dev, requires_grad=True)
        bm = torchsde_amd.BrownianInterval(0.0, n * dt, size=(B, d), dtype=torch.float32, device=dev, entropy=i, dt=dt,
                                           levy_area_approximation=levy)
        ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=dt)
        torch.cuda.synchronize()
        t = time.perf_counter()
        ys[-1].sum().backward()
        torch.cuda.synchronize()
        return time.perf_counter() - t

    go(0)
    torch.cuda.synchronize()
    t = time.perf_counter()
    bwd = sum(go(1 + i) for i in range(3)) / 3
    total = (time.perf_counter() - t) / 3
    print(f"{method:9s} fwd+bwd {total * 1e3:7.2f} ms  (backward reductions {bwd * 1e3:5.2f} ms)  "
          f"{B * n / total:.3e} trajectory-steps/s with gradients")

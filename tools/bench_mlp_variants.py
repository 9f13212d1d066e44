"""Sampling time of the perceptron-drift kernel's other variants (midpoint, sigmoid diffusion, Milstein) at
d = hidden = 128 and 64, 500 steps. Run on the GPU box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import torchsde_amd  # noqa: E402

dev = "cuda"
for method, sde_type, diffusion in (("midpoint", "stratonovich", "affine"), ("euler", "ito", "sigmoid"), ("milstein", "ito", "affine")):
    for d in (128, 64):
        B = 32768 if d == 128 else 65536
        torch.manual_seed(0)
        sde = torchsde_amd.MLPDriftDiagonalSDE(d, d, activation="softplus", sde_type=sde_type, diffusion=diffusion, diff_rate=0.0, diff_shift=0.1).to(dev)
        y0 = torch.full((B, d), 0.1, device=dev); dt = 2.0 ** -9; ts = torch.tensor([0.0, 500 * dt], device=dev)
        def solve(i):
            bm = torchsde_amd.BrownianInterval(0.0, 500 * dt, size=(B, d), dtype=torch.float32, device=dev, entropy=i, dt=dt)
            with torch.no_grad():
                return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=dt)
        for i in range(2): solve(i)
        torch.cuda.synchronize(); t = time.perf_counter()
        for i in range(3): solve(10 + i)
        torch.cuda.synchronize()
        print(method, diffusion, d, f"{(time.perf_counter() - t) / 3 * 1e3:.2f} ms")

"""Training THROUGH `sdeint` (ordinary autograd through the solver) on the reference's scalar-noise problem (ExScalar,
tests/problems.py:75-103): f = -p^2 sin(y) cos(y)^3, g = p cos(y)^2 with one Brownian channel per trajectory. The module is
plain torch code; its drift and diffusion travel to the kernel as small expression programs, and with autograd recording
the same programs run on forward-mode dual numbers -- one launch forward, a few reductions backward, no step stored.

    python examples/scalar_noise_training.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

import torch
from torch import nn

import torchsde_amd as torchsde  # noqa: E402


class ExScalar(nn.Module):
    noise_type, sde_type = "scalar", "ito"

    def __init__(self, d):
        super().__init__()
        self.p = nn.Parameter(torch.sigmoid(torch.randn(d)))

    def f(self, t, y):
        return -self.p ** 2. * torch.sin(y) * torch.cos(y) ** 3.

    def g(self, t, y):
        return (self.p * torch.cos(y) ** 2).unsqueeze(dim=-1)


if __name__ == "__main__":
    device = "cuda"
    batch, d = 8192, 16
    torch.manual_seed(0)
    target = ExScalar(d).to(device)
    model = ExScalar(d).to(device)
    y0 = torch.full((batch, d), 0.5, device=device)
    ts = torch.tensor([0.0, 0.5, 1.0], device=device)
    optimiser = torch.optim.Adam(model.parameters(), lr=5e-2)
    with torch.no_grad():
        bm = torchsde.BrownianInterval(0.0, 1.0, size=(batch, 1), device=device, dtype=torch.float32, entropy=1234,
                                       levy_area_approximation="space-time")
        want = torchsde.sdeint(target, y0, ts, bm=bm, dt=1e-2)[-1].var(0)          # (default method for scalar Ito noise: srk)
    for it in range(30):
        bm = torchsde.BrownianInterval(0.0, 1.0, size=(batch, 1), device=device, dtype=torch.float32, entropy=it,
                                       levy_area_approximation="space-time")
        ys = torchsde.sdeint(model, y0, ts, bm=bm, dt=1e-2)
        loss = ((ys[-1].var(0) - want) ** 2).sum()
        optimiser.zero_grad()
        loss.backward()
        optimiser.step()
        if it % 5 == 0 or it == 29:
            print(f"iteration {it:2d}  loss {loss.item():.3e}  |p - p*| {(model.p - target.p).abs().max().item():.3f}  "
                  f"backward node: {type(ys.grad_fn).__name__}")

"""The reference's README quick example (torchsde README.md:25-54), unchanged except for the import and the device.

    python examples/quickstart.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

import torch

import torchsde_amd as torchsde  # noqa: E402

batch_size, state_size, brownian_size = 32, 3, 2
t_size = 20


class SDE(torch.nn.Module):
    noise_type = "general"
    sde_type = "ito"

    def __init__(self):
        super().__init__()
        self.mu = torch.nn.Linear(state_size, state_size)
        self.sigma = torch.nn.Linear(state_size, state_size * brownian_size)

    def f(self, t, y):                       # drift: (batch_size, state_size)
        return self.mu(y)

    def g(self, t, y):                       # diffusion: (batch_size, state_size, brownian_size)
        return self.sigma(y).view(batch_size, state_size, brownian_size)


if __name__ == "__main__":
    device = "cuda"
    sde = SDE().to(device)
    y0 = torch.full((batch_size, state_size), 0.1, device=device)
    ts = torch.linspace(0, 1, t_size, device=device)
    with torch.no_grad():
        ys = torchsde.sdeint(sde, y0, ts)    # default method (Euler for Ito), default dt = 1e-3
    print("ys", tuple(ys.shape), "finite:", bool(torch.isfinite(ys).all()))

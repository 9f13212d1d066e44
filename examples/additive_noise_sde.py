"""Additive noise the way the reference's tests write it: ExAdditive (tests/problems.py:106-132: drift and diffusion use t,
the diffusion is one value per channel repeated over the m Brownian channels) and NeuralAdditive (tests/problems.py:195-224:
the drift a network of `cat([t, y])`, the diffusion a network of t alone). Nothing of this package is in the modules; `sdeint`
with its default method for additive noise (SRK = SRA1) runs each as ONE kernel launch per solve: the drift travels as an
expression program or runs on the matrix cores, the diffusion is tabulated over the scheme's stage times by one batched call of
`g` -- once the first solve has checked that route against the stepwise one.

    python examples/additive_noise_sde.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

import time

import torch
from torch import nn

import torchsde_amd as torchsde  # noqa: E402


class ExAdditive(nn.Module):
    noise_type, sde_type = "additive", "ito"

    def __init__(self, d, m):
        super().__init__()
        self.m = m
        self.a = nn.Parameter(torch.sigmoid(torch.randn(d)))
        self.b = nn.Parameter(torch.sigmoid(torch.randn(d)))

    def f(self, t, y):
        return self.b / torch.sqrt(1. + t) - y / (2. + 2. * t)

    def g(self, t, y):
        fill_value = self.a * self.b / torch.sqrt(1. + t)
        return fill_value.unsqueeze(dim=0).unsqueeze(dim=-1).repeat(y.size(0), 1, self.m)


class NeuralAdditive(nn.Module):
    noise_type, sde_type = "additive", "ito"

    def __init__(self, d, m, hidden=64):
        super().__init__()
        self.d, self.m = d, m
        self.f_net = nn.Sequential(nn.Linear(d + 1, hidden), nn.Softplus(), nn.Linear(hidden, d))
        self.g_net = nn.Sequential(nn.Linear(1, hidden), nn.Softplus(), nn.Linear(hidden, d * m), nn.Sigmoid())

    def f(self, t, y):
        return self.f_net(torch.cat([t.expand(y.size(0), 1), y], dim=1))

    def g(self, t, y):
        return self.g_net(t.expand(y.size(0), 1)).view(y.size(0), self.d, self.m)


if __name__ == "__main__":
    device = "cuda"
    batch, d, m, steps = 65536, 64, 8, 1000
    torch.manual_seed(0)
    y0 = torch.full((batch, d), 0.1, device=device)
    ts = torch.tensor([0.0, 0.5, 1.0], device=device)
    from torchsde_amd import recognise
    for sde in (ExAdditive(d, m).to(device), NeuralAdditive(d, m).to(device)):
        for options, label in ((None, "default call"), ({"trajectory_kernel": False}, "stepwise route")):
            for rep in range(3):
                bm = torchsde.BrownianInterval(0.0, 1.0, size=(batch, m), device=device, dtype=torch.float32, entropy=rep,
                                               levy_area_approximation="space-time")
                torch.cuda.synchronize()
                t = time.perf_counter()
                with torch.no_grad():
                    ys = torchsde.sdeint(sde, y0, ts, bm=bm, dt=1.0 / steps, options=options)       # method: the default, SRK
                torch.cuda.synchronize()
                elapsed = time.perf_counter() - t
            print(f"{type(sde).__name__:15s} {label:16s} {elapsed * 1e3:8.2f} ms per solve   mean {ys[-1].mean().item():+.4f}  "
                  f"std {ys[-1].std().item():.4f}")
        print("\n".join(recognise.describe(sde)))

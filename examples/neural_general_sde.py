"""A general-noise neural SDE written the way the reference's tests write them (NeuralGeneral, tests/problems.py:226-252):
drift and diffusion are small networks of `cat([t, y])`, the diffusion reshaped to (B, d, m). Nothing of this package is in
the module; `sdeint` with no options runs it as ONE kernel launch per solve (all four layers and the contraction with the
Brownian increments on the f32 matrix cores, weights in LDS) once the first solve has checked that route against the
stepwise one.

    python examples/neural_general_sde.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

import time

import torch
from torch import nn

import torchsde_amd as torchsde  # noqa: E402


class NeuralGeneral(nn.Module):
    noise_type, sde_type = "general", "ito"

    def __init__(self, d, m, hidden=64):
        super().__init__()
        self.d, self.m = d, m
        self.f_net = nn.Sequential(nn.Linear(d + 1, hidden), nn.Softplus(), nn.Linear(hidden, d))
        self.g_net = nn.Sequential(nn.Linear(d + 1, hidden), nn.Softplus(), nn.Linear(hidden, d * m), nn.Sigmoid())

    def f(self, t, y):
        return self.f_net(torch.cat([t.expand(y.size(0), 1), y], dim=1))

    def g(self, t, y):
        return self.g_net(torch.cat([t.expand(y.size(0), 1), y], dim=1)).reshape(y.size(0), self.d, self.m)


if __name__ == "__main__":
    device = "cuda"
    batch, d, m, steps = 16384, 32, 16, 1000
    torch.manual_seed(0)
    sde = NeuralGeneral(d, m).to(device)
    y0 = torch.full((batch, d), 0.1, device=device)
    ts = torch.tensor([0.0, 0.5, 1.0], device=device)
    for options, label in ((None, "default call"), ({"trajectory_kernel": False}, "stepwise route")):
        for rep in range(3):
            bm = torchsde.BrownianInterval(0.0, 1.0, size=(batch, m), device=device, dtype=torch.float32, entropy=rep)
            torch.cuda.synchronize()
            t = time.perf_counter()
            with torch.no_grad():
                ys = torchsde.sdeint(sde, y0, ts, bm=bm, method="euler", dt=1.0 / steps, options=options)
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t
        print(f"{label:16s} {elapsed * 1e3:8.2f} ms per solve   mean {ys[-1].mean().item():+.4f}  std {ys[-1].std().item():.4f}")
    from torchsde_amd import recognise
    print("\n".join(recognise.describe(sde)))

"""Training a neural SDE with the pair the reference recommends for adjoint training -- ``method="reversible_heun"`` with
``adjoint_method="adjoint_reversible_heun"`` -- on an ORDINARY module: drift and diffusion are `nn.Sequential`s of Linear
layers with LipSwish between them and a closing Tanh, general noise, the state read out at 32 times (the shape of the
generator in the reference's `examples/sde_gan.py`, written here from its description). Nothing in the module knows about
this package: `sdeint_adjoint` interprets it at every solve, runs the forward solve as ONE launch on the matrix cores and the
backward pass as the scheme's exact reconstruction sweep (nothing of the trajectory stored), and the gradients land on the
module's own `nn.Linear` tensors. The first iteration solves both ways and compares values and gradients.

    python examples/train_reversible_heun.py [--iters 30] [--stepwise]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

import torch
from torch import nn

import torchsde_amd as torchsde  # noqa: E402


class LipSwish(nn.Module):
    def forward(self, x):
        return 0.909 * torch.nn.functional.silu(x)


def mlp(n_in, n_out, width, layers, closing):
    mods = [nn.Linear(n_in, width), LipSwish()]
    for _ in range(layers - 1):
        mods += [nn.Linear(width, width), LipSwish()]
    mods.append(nn.Linear(width, n_out))
    if closing:
        mods.append(nn.Tanh())
    return nn.Sequential(*mods)


class Generator(nn.Module):
    sde_type = "stratonovich"
    noise_type = "general"

    def __init__(self, hidden, noise, width, layers):
        super().__init__()
        self.hidden, self.noise = hidden, noise
        self.drift = mlp(1 + hidden, hidden, width, layers, closing=True)
        self.diffusion = mlp(1 + hidden, hidden * noise, width, layers, closing=True)

    def f_and_g(self, t, x):
        tx = torch.cat([t.expand(x.size(0), 1), x], dim=1)
        return self.drift(tx), self.diffusion(tx).view(x.size(0), self.hidden, self.noise)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--stepwise", action="store_true", help="the stepwise solver and its adjoint instead of the kernels")
    args = ap.parse_args()
    device = "cuda"
    batch, hidden, noise, steps = 4096, 16, 3, 64
    torch.manual_seed(0)
    sde = Generator(hidden, noise, width=16, layers=2).to(device)
    readout = nn.Linear(hidden, 1).to(device)
    optimizer = torch.optim.Adam(list(sde.parameters()) + list(readout.parameters()), lr=3e-3)
    ts = torch.linspace(0.0, 1.0, 33, device=device)
    target = torch.sin(3.0 * ts)                   # the mean path the read-out should follow
    options = {"trajectory_kernel": False} if args.stepwise else {}
    warm = min(3, args.iters - 1)
    for it in range(args.iters):
        if it == warm:
            torch.cuda.synchronize()
            start = time.perf_counter()
        x0 = 0.1 * torch.randn(batch, hidden, device=device)
        bm = torchsde.BrownianInterval(0.0, 1.0, size=(batch, noise), device=device, dtype=torch.float32, entropy=it)
        xs = torchsde.sdeint_adjoint(sde, x0, ts, bm=bm, method="reversible_heun", dt=1.0 / steps,
                                     adjoint_method="adjoint_reversible_heun", adjoint_params=tuple(sde.parameters()),
                                     options=dict(options), adjoint_options=dict(options))
        ys = readout(xs).squeeze(-1)               # (times, batch)
        loss = ((ys.mean(1) - target) ** 2).mean() + 0.1 * ((ys.std(1) - 0.2) ** 2).mean()
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        if it % 10 == 0 or it == args.iters - 1:
            print(f"iter {it:3d}  loss {loss.item():.5f}")
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - start) / (args.iters - warm) * 1e3:.2f} ms per iteration "
          f"({'stepwise solver and adjoint' if args.stepwise else 'reversible-Heun kernels'})")

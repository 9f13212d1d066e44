"""A latent-SDE-style training loop (posterior drift vs prior drift, KL through `logqp=True`) with the solver's work
replayed as HIP graphs: `options={"hip_graph": True}` records the forward solve together with its autograd graph
and the back-propagation through it; with `sdeint_adjoint`, `adjoint_options={"hip_graph": True}` records the whole
backward sweep. Parameters are read in place by the recorded kernels, so the optimiser just steps.

    python examples/train_latent_sde.py [--adjoint] [--steps 20]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

import argparse
import time

import torch
from torch import nn

import torchsde_amd as torchsde  # noqa: E402


class LatentSDE(nn.Module):
    noise_type, sde_type = "diagonal", "ito"

    def __init__(self, d=8, hidden=64):
        super().__init__()
        self.posterior = nn.Sequential(nn.Linear(d, hidden), nn.Softplus(), nn.Linear(hidden, d))
        self.prior = nn.Sequential(nn.Linear(d, hidden), nn.Softplus(), nn.Linear(hidden, d))
        self.log_sigma = nn.Parameter(torch.full((d,), -1.0))

    def f(self, t, y):
        return self.posterior(y)

    def h(self, t, y):
        return self.prior(y)

    def g(self, t, y):
        return torch.exp(self.log_sigma).expand_as(y)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--adjoint", action="store_true")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--eager", action="store_true")
    args = ap.parse_args()
    device = "cuda"
    batch, d = 1024, 8
    torch.manual_seed(0)
    sde = LatentSDE(d).to(device)
    opt = torch.optim.Adam(sde.parameters(), lr=1e-2)
    ts = torch.linspace(0.0, 1.0, 5, device=device)
    target = torch.linspace(-1.0, 1.0, d, device=device)
    graph = {} if args.eager else {"hip_graph": True}
    solve = torchsde.sdeint_adjoint if args.adjoint else torchsde.sdeint
    extra = {"adjoint_options": dict(graph)} if args.adjoint else {}
    t0 = None
    for it in range(args.steps):
        if it == 2:                          # the first iterations record the graphs
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        y0 = torch.zeros(batch, d, device=device)
        bm = torchsde.BrownianInterval(0.0, 1.0, size=(batch, d + 1), device=device, dtype=torch.float32, entropy=it)
        ys, logqp = solve(sde, y0, ts, bm=bm, method="euler", dt=2.0 ** -6, logqp=True, options=dict(graph), **extra)
        loss = ((ys[-1] - target) ** 2).mean() + 0.1 * logqp.sum(0).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        if it % 5 == 0 or it == args.steps - 1:
            print(f"iteration {it:3d}  loss {loss.item():.4f}")
    torch.cuda.synchronize()
    if t0 is not None and args.steps > 2:
        print(f"{(time.perf_counter() - t0) / (args.steps - 2) * 1e3:.2f} ms per training iteration "
              f"({'adjoint' if args.adjoint else 'backprop'}, {'eager' if args.eager else 'HIP graphs'})")


if __name__ == "__main__":
    main()

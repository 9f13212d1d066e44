"""Training a neural SDE through the solver on the matrix cores: `MLPDriftDiagonalSDE` + `sdeint(method="euler")` with
autograd on. The forward solve is one kernel launch, `loss.backward()` is a reverse-sweep kernel plus the
weight-gradient products -- no autograd tape over the steps. The toy task: learn a drift that carries N(0, I)
samples to a target mean and scale at t = 1.

    python examples/train_neural_sde.py [--iters 40] [--stepwise]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

import time

import torch

import torchsde_amd as torchsde  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--stepwise", action="store_true", help="back-propagate through the stepwise solver instead")
    args = ap.parse_args()
    device = "cuda"
    batch, d, hidden, steps = 8192, 32, 64, 64
    torch.manual_seed(0)
    sde = torchsde.MLPDriftDiagonalSDE(d, hidden, activation="tanh", diff_rate=0.0, diff_shift=0.3).to(device)
    target_mean = torch.linspace(-1.0, 1.0, d, device=device)
    target_std = 0.5
    optimizer = torch.optim.Adam(sde.parameters(), lr=1e-2)
    ts = torch.tensor([0.0, 1.0], device=device)
    options = {"trajectory_kernel": False} if args.stepwise else {}
    warm = min(5, args.iters - 1)          # the first iterations load the library and warm the allocator
    for it in range(args.iters):
        if it == warm:
            torch.cuda.synchronize()
            start = time.perf_counter()
        y0 = torch.randn(batch, d, device=device)
        bm = torchsde.BrownianInterval(0.0, 1.0, size=(batch, d), device=device, dtype=torch.float32, entropy=it)
        ys = torchsde.sdeint(sde, y0, ts, bm=bm, method="euler", dt=1.0 / steps, options=dict(options))
        y1 = ys[-1]
        loss = ((y1.mean(0) - target_mean) ** 2).mean() + ((y1.std(0) - target_std) ** 2).mean()
        optimizer.zero_grad()
        loss.backward()
        optimizer.step()
        if it % 10 == 0 or it == args.iters - 1:
            print(f"iter {it:3d}  loss {loss.item():.5f}  diffusion shift {sde.diff_shift.item():.3f}")
    torch.cuda.synchronize()
    print(f"{(time.perf_counter() - start) / (args.iters - warm) * 1e3:.2f} ms per iteration "
          f"({'stepwise solver + autograd tape' if args.stepwise else 'trajectory kernels'})")

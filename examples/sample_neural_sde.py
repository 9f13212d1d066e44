"""Sampling a trained neural SDE whose drift is a two-layer perceptron: `MLPDriftDiagonalSDE` is an ordinary module for
training (any solver, autograd, sdeint_adjoint); forward solves without autograd run as ONE kernel launch with both
layers on the f32 matrix cores.

    python examples/sample_neural_sde.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

import time

import torch

import torchsde_amd as torchsde  # noqa: E402

if __name__ == "__main__":
    device = "cuda"
    batch, d, hidden, steps = 32768, 128, 128, 500
    torch.manual_seed(0)
    sde = torchsde.MLPDriftDiagonalSDE(d, hidden, activation="softplus", diff_rate=0.0, diff_shift=0.1).to(device)
    y0 = torch.zeros(batch, d, device=device)
    ts = torch.tensor([0.0, 1.0], device=device)
    for options, label in (({}, "matrix-core sampling kernel"), ({"trajectory_kernel": False, "hip_graph": True},
                                                                  "stepwise path, HIP-graph replay")):
        for rep in range(3):
            bm = torchsde.BrownianInterval(0.0, 1.0, size=(batch, d), device=device, dtype=torch.float32, entropy=rep)
            torch.cuda.synchronize()
            t = time.perf_counter()
            with torch.no_grad():
                ys = torchsde.sdeint(sde, y0, ts, bm=bm, method="euler", dt=1.0 / steps, options=dict(options))
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t
        print(f"{label:34s} {elapsed * 1e3:7.2f} ms   mean {ys[-1].mean().item():+.4f}  std {ys[-1].std().item():.4f}")

"""Monte Carlo with closed-form SDEs: the whole solve is one kernel launch, and with autograd on the same launch
carries path-wise sensitivities ("Greeks") -- gradients without a backward sweep through the steps.

Prices a basket of European calls under geometric Brownian motion and differentiates the price with respect to the
spot, the drift and the volatility of every underlying.

    python examples/monte_carlo_closed_form.py
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # run from a checkout

import math
import time

import torch

import torchsde_amd as torchsde  # noqa: E402

if __name__ == "__main__":
    device = "cuda"
    paths, assets, steps = 1 << 18, 16, 256
    rate, strike, maturity = 0.03, 1.0, 1.0
    sigma = torch.linspace(0.1, 0.4, assets, device=device, requires_grad=True)
    mu = torch.full((assets,), rate, device=device, requires_grad=True)
    sde = torchsde.AffineDiagonalSDE(mu, 0.0, sigma, 0.0, dtype=torch.float32).to(device)
    # the module owns copies of the coefficients as parameters: differentiate with respect to those
    spot = torch.ones(paths, assets, device=device, requires_grad=True)
    ts = torch.tensor([0.0, maturity], device=device)
    bm = torchsde.BrownianInterval(0.0, maturity, size=(paths, assets), device=device, dtype=torch.float32, entropy=7)

    torch.cuda.synchronize()
    t = time.perf_counter()
    ys = torchsde.sdeint(sde, spot, ts, bm=bm, method="milstein", dt=maturity / steps)
    payoff = torch.relu(ys[-1] - strike).mean(0) * math.exp(-rate * maturity)       # one price per asset
    payoff.sum().backward()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t
    delta = spot.grad.sum(0)                                                          # d price / d spot
    vega = sde.diff_rate.grad                                                         # d price / d sigma
    print(f"{paths} paths x {assets} assets x {steps} Milstein steps, price + Greeks in {elapsed * 1e3:.1f} ms")
    for i in (0, assets // 2, assets - 1):
        s = sigma[i].item()
        d1 = (math.log(1.0 / strike) + (rate + 0.5 * s * s) * maturity) / (s * math.sqrt(maturity))
        nd1 = 0.5 * (1 + math.erf(d1 / math.sqrt(2)))
        bs_vega = math.exp(-0.5 * d1 * d1) / math.sqrt(2 * math.pi) * math.sqrt(maturity)
        print(f"  asset {i:2d} sigma {s:.2f}: price {payoff[i].item():.4f}  delta {delta[i].item():.4f} "
              f"(Black-Scholes {nd1:.4f})  vega {vega[i].item():.4f} (Black-Scholes {bs_vega:.4f})")

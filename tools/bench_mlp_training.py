"""Training step (forward + backward) of a perceptron-drift diagonal SDE at the BASELINE configs[4] shape: the
trajectory kernels (sampling kernel + reverse sweep + weight-gradient products) vs back-propagation through the stepwise
solver and vs the stochastic adjoint (both as HIP-graph replays). Run on the GPU box.

    python tools/bench_mlp_training.py [--B 32768] [--d 128] [--hidden 128] [--steps 500] [--reps 3] [--no-stepwise]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import torchsde_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=32768)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-stepwise", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda")
    dt = 2.0 ** -9
    torch.manual_seed(0)
    sde = torchsde_amd.MLPDriftDiagonalSDE(args.d, args.hidden, activation="softplus", diff_rate=0.05,
                                           diff_shift=0.1).to(dev)
    ts = torch.tensor([0.0, args.steps * dt], device=dev)

    def step(i, fn, options):
        y0 = torch.full((args.B, args.d), 0.1, device=dev, requires_grad=True)
        bm = torchsde_amd.BrownianInterval(0.0, args.steps * dt, size=(args.B, args.d), dtype=torch.float32,
                                           device=dev, entropy=i, dt=dt)
        sde.zero_grad()
        ys = fn(sde, y0, ts, bm=bm, method="euler", dt=dt, options=options)
        ys[-1].sum().backward()
        return y0.grad

    def timed(fn, options):
        for i in range(2):
            step(i, fn, options)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(args.reps):
            g = step(10 + i, fn, options)
        torch.cuda.synchronize()
        elapsed = (time.perf_counter() - t) / args.reps * 1e3
        assert torch.isfinite(g).all()
        return elapsed

    flops = (2 + 3 + 2) * 2.0 * args.B * args.d * args.hidden * args.steps      # fwd 2, sweep 3, weight sums 2 products
    fast = timed(torchsde_amd.sdeint, {})
    print(f"B={args.B} d={args.d} hidden={args.hidden} steps={args.steps}")
    print(f"  trajectory kernels (fwd + reverse sweep + weight sums): {fast:8.2f} ms   "
          f"{flops / fast / 1e9:6.1f} TFLOP/s f32   {args.B * args.steps / fast * 1e3:.3e} traj-steps/s fwd+bwd")
    if not args.no_stepwise:
        slow = timed(torchsde_amd.sdeint, {"trajectory_kernel": False, "hip_graph": True})
        print(f"  back-propagation through the stepwise solver (graph):   {slow:8.2f} ms   x{slow / fast:.1f}")
        adj = timed(lambda *a, **k: torchsde_amd.sdeint_adjoint(*a, adjoint_method="euler", **k),
                    {"trajectory_kernel": False, "hip_graph": True})
        print(f"  stochastic adjoint, stepwise (graph):                   {adj:8.2f} ms   x{adj / fast:.1f}")


if __name__ == "__main__":
    main()

"""Whole-trajectory kernel vs the stepwise path on the BASELINE configs[1] shape (run on the GPU box).

    python tools/bench_trajectory.py [--B 65536] [--d 64] [--steps 1000] [--reps 5]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

import torchsde_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=65536)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--stepwise", action="store_true", help="also time the stepwise path (HIP-graph replay)")
    args = ap.parse_args()
    dev = torch.device("cuda")
    dt = 2.0 ** -10
    results = []
    for dtype in (torch.float32, torch.float64):
        for method, sde_type in (("euler", "ito"), ("milstein", "ito"), ("midpoint", "stratonovich"), ("srk", "ito")):
            sde = torchsde_amd.AffineDiagonalSDE(0.1, 0.0, 0.2, 0.0, sde_type=sde_type, dtype=dtype, device=dev)
            y0 = torch.full((args.B, args.d), 0.1, dtype=dtype, device=dev)
            ts = torch.tensor([0.0, args.steps * dt], dtype=dtype, device=dev)
            levy = "space-time" if method == "srk" else "none"

            def solve(i, options):
                bm = torchsde_amd.BrownianInterval(0.0, args.steps * dt, size=(args.B, args.d), dtype=dtype, device=dev,
                                                   entropy=100 + i, dt=dt, levy_area_approximation=levy)
                with torch.no_grad():
                    return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=dt, options=options)

            def timed(options):
                for i in range(2):
                    solve(i, options)
                torch.cuda.synchronize()
                t = time.perf_counter()
                for i in range(args.reps):
                    out = solve(10 + i, options)
                torch.cuda.synchronize()
                elapsed = time.perf_counter() - t
                assert torch.isfinite(out).all()
                return elapsed / args.reps * 1e3

            rec = {"dtype": str(dtype).split(".")[-1], "method": method, "trajectory_ms": timed({})}
            if args.stepwise and not (method == "milstein"):
                rec["stepwise_graph_ms"] = timed({"trajectory_kernel": False, "hip_graph": True})
            elif args.stepwise:
                rec["stepwise_eager_ms"] = timed({"trajectory_kernel": False})
            rec["traj_steps_per_s"] = args.B * args.steps / rec["trajectory_ms"] * 1e3
            results.append(rec)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()

"""Perceptron-drift sampling kernel vs the stepwise path (HIP-graph replay) at the BASELINE configs[4] shape, forward."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import torchsde_amd  # noqa: E402

dev = torch.device("cuda")
only = int(sys.argv[sys.argv.index("--only") + 1]) if "--only" in sys.argv else None
stepwise = "--no-stepwise" not in sys.argv
for idx, (B, d, hidden, n) in enumerate(((32768, 128, 128, 500), (65536, 64, 64, 500), (262144, 32, 32, 200), (1024, 64, 64, 500))):
    if only is not None and idx != only:
        continue
    dt = 2.0 ** -9
    torch.manual_seed(0)
    sde = torchsde_amd.MLPDriftDiagonalSDE(d, hidden, activation="softplus", diff_rate=0.0, diff_shift=0.1).to(dev)
    y0 = torch.full((B, d), 0.1, device=dev)
    ts = torch.tensor([0.0, n * dt], device=dev)

    def solve(i, options):
        bm = torchsde_amd.BrownianInterval(0.0, n * dt, size=(B, d), dtype=torch.float32, device=dev, entropy=i, dt=dt)
        with torch.no_grad():
            return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="euler", dt=dt, options=options)

    def timed(options, reps=3):
        for i in range(2):
            solve(i, options)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(reps):
            out = solve(10 + i, options)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t
        assert torch.isfinite(out).all()
        return elapsed / reps * 1e3

    fast = timed({})
    slow = timed({"trajectory_kernel": False, "hip_graph": True}) if stepwise else float("nan")
    flops = 4.0 * B * d * hidden * n
    print(f"B={B} d={d} hidden={hidden} steps={n}: kernel {fast:8.2f} ms ({flops / fast / 1e9:6.1f} TFLOP/s f32, "
          f"{B * n / fast * 1e3:.3e} traj-steps/s)   stepwise graph {slow:8.2f} ms   x{slow / fast:.1f}")

"""Adaptive solves (step doubling, Milstein on GBM over [0, 1], 4 output times): accept / reject decided on the device
(adaptive.py: the host synchronises once per round of attempts; eagerly issued, or with the attempt replayed as a
cached HIP graph) vs decided on the host (one sync per attempt)."""
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torchsde_amd  # noqa: E402
from torchsde_amd import adaptive  # noqa: E402
from workloads import problems  # noqa: E402

dev = "cuda"
for (B, d) in ((1024, 16), (65536, 64)):
    sde = problems.make("gbm_ito", d=d).to(dev)
    y0 = torch.full((B, d), 0.1, device=dev)
    ts = torch.tensor([0.0, 0.25, 0.5, 0.75, 1.0], device=dev)

    def solve(i, device_control, graph):
        bm = torchsde_amd.BrownianInterval(0.0, 1.0, size=(B, d), device=dev, dtype=torch.float32, entropy=i)
        with torch.no_grad():
            return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="milstein", dt=0.05, adaptive=True, rtol=1e-3,
                                       atol=1e-4, options={"device_adaptive": device_control, "hip_graph": graph})
    for device_control, graph, label in ((True, False, "device             "), (True, True, "device, graph replay"),
                                         (False, False, "host               ")):
        solve(0, device_control, graph)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for i in range(5):
            out = solve(1 + i, device_control, graph)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 5 * 1e3
        stats = adaptive.last_stats if device_control else "one sync per attempt"
        print(f"B={B} d={d} adaptive milstein, control on the {label}: {ms:8.2f} ms per solve   {stats}")

"""The Levy-area correction term (m JVPs of the diffusion columns) at the BASELINE configs[2] shape: the reference's two
formulations (base_sde.py:164-209): v1 = m double-backward JVPs, v2 = one JVP over an m-times replicated batch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from workloads import problems  # noqa: E402
from torchsde_amd.sde import ForwardSDE  # noqa: E402

dev = "cuda"
B, d, m = 16384, 32, 16
sde = ForwardSDE(problems.MLPGeneral(d, m, "ito", hidden=64).to(dev))
y = torch.full((B, d), 0.1, device=dev)
a = 0.01 * torch.randn(B, m, m, device=dev)
t = torch.tensor(0.0, device=dev)
with torch.no_grad():
    ref = None
    for name in ("v1", "v2"):
        fn = getattr(sde, "dg_ga_jvp_column_sum_" + name)
        out = fn(t, y, a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            out = fn(t, y, a)
        torch.cuda.synchronize()
        print(name, f"{(time.perf_counter() - t0) / 5 * 1e3:.2f} ms", "max|diff vs v1|",
              0.0 if ref is None else (out - ref).abs().max().item())
        ref = out if ref is None else ref

"""step_general at the shapes VERDICT r3 names, graph-replayed on rotating operands (tools/bench_kernels.py's method)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_kernels import gen_noise, timeit  # noqa: E402
from torchsde_amd import kernels as K  # noqa: E402

dev, dt = "cuda", 2.0 ** -10
EXTERNAL = "--external" in sys.argv      # increments read from memory instead of generated: what the RNG costs
ONCE = "--once" in sys.argv       # (counter passes: a few eager launches per shape instead of graph-replayed timing)
for (B, d, m) in [(16384, 32, 16), (65536, 16, 16), (16384, 32, 64), (16384, 64, 16), (65536, 32, 16), (4096, 32, 16),
                  (262144, 32, 16)]:
    nbytes = 4 * B * (d * m + 3 * d)
    k = max(2, min(8, (256 << 20) // nbytes))
    ys = [torch.rand(B, d, device=dev) for _ in range(k + 1)]
    fs = [torch.randn(B, d, device=dev) for _ in range(k)]
    gs = [torch.rand(B, d, m, device=dev) for _ in range(k)]
    specs = [gen_noise((B, m), i, dt, dev) for i in range(200)]
    if EXTERNAL:
        Ws = [torch.randn(B, m, device=dev) for _ in range(k)]
        specs = [K.NoiseSpec.external(Ws[i % k]) for i in range(200)]
    if ONCE:
        for i in range(4):
            K._raw_step_general(ys[i % k], fs[i % k], gs[i % k], dt, 1.0, specs[i], ys[i % k + 1])
        S = torch.randn(d, m, device=dev)
        for i in range(4):
            K._raw_step_shared(ys[i % k], fs[i % k], S, 1.0, dt, 1.0, 0, 0.0, 0.0, 0.0, specs[i], ys[i % k + 1])
        torch.cuda.synchronize()
        continue
    us = timeit(lambda i: K._raw_step_general(ys[i % k], fs[i % k], gs[i % k], dt, 1.0, specs[i], ys[i % k + 1]))
    print(f"step_general B={B} d={d} m={m} ({k} operand sets) {us:8.2f} us {nbytes / us / 1e3:8.1f} GB/s "
          f"({nbytes / us / 1e3 / 80:.1f} % of 8 TB/s)")

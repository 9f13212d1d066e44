"""Why does the parallel-branch graph of the C5 stepwise adjoint's backward sweep not reproduce the sequential one?
Prints, per output tensor, the largest entry of each graph's result and their largest difference."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from torchsde_amd import graph  # noqa: E402

real = graph._same_tensors


def verbose(xs, ys, exact=True):
    for i, (a, b) in enumerate(zip(xs, ys)):
        print(f"  output {i} {tuple(a.shape)}: max|a| {a.abs().max().item():.3e} max|b| {b.abs().max().item():.3e} "
              f"max|a-b| {(a - b).abs().max().item():.3e} nan {int(a.isnan().sum())}/{int(b.isnan().sum())} exact={exact}")
    return real(xs, ys, exact)


graph._same_tensors = verbose
name = sys.argv[1] if len(sys.argv) > 1 else "c5_adjoint_latent_b32768_d128_s500"
job = bench.Job(name, torch.device("cuda", 0))
job.cfg = dict(job.cfg, nsteps=100)
job.ts = torch.tensor([0.0, 100 * job.cfg["dt"]], device="cuda")
for i in range(3):
    job.solve(i)
torch.cuda.synchronize()
for line in graph.describe_cache(job.sde):
    print(line[:300])

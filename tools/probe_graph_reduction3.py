"""What breaks a recorded graph of several multi-block reductions (tools/probe_graph_reduction2.py)? Twenty column sums
of a 4096 x 128 tensor in one HIP graph, replayed three times, with different kinds of eager work between the replays;
run once per setting of the HIP runtime's graph flags (the shell sets the environment: see tools/r3_probe_flags.sh).
Second part: the adjoint's backward sweep at that size with the row limit of "auto" lifted -- does its graph pass the
replay checks under this setting?"""
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

dev = "cuda"
torch.manual_seed(0)
B, d = 4096, 128
flags = {k: v for k, v in os.environ.items() if k.startswith(("DEBUG_CLR", "DEBUG_HIP", "HIP_FORCE"))}
print("flags:", flags or "(defaults)")

y = torch.randn(B, d, device=dev)
other = torch.randn(B, d, device=dev)


def twenty_sums():
    return [sum((y * float(k)).sum(0) for k in range(1, 21))]


def between_nothing():
    pass


def between_elementwise():
    (other * 2.0).add_(1.0)


def between_reduction():
    other.sum(0)


def between_full_reduction():
    other.sum()


def between_allocation():
    torch.empty(1 << 20, device=dev).fill_(1.0)


def between_memset():
    other.new_empty(4096).zero_()


want = [o.clone() for o in twenty_sums()]
for tag, between in [("nothing", between_nothing), ("elementwise kernels", between_elementwise),
                     ("a column reduction", between_reduction), ("a full reduction", between_full_reduction),
                     ("an allocation + fill", between_allocation), ("a small memset", between_memset)]:
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        twenty_sums()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        outs = twenty_sums()
    kept = []
    for _ in range(4):
        g.replay()
        kept.append(outs[0].clone())          # (a copy kernel: the least eager work that keeps the result)
        between()
    torch.cuda.synchronize()
    errs = [((k - want[0]).abs().max() / want[0].abs().max()).item() for k in kept]
    print(f"  between replays: {tag:24s} relative error of replays 1-4:", " ".join(f"{e:.1e}" for e in errs))
    del g, outs

# ---- the adjoint's backward sweep at 4096 x 128, "auto" allowed to record it -------------------------------------------
import torchsde_amd                                                     # noqa: E402
from torchsde_amd import graph                                          # noqa: E402
from workloads import problems                                          # noqa: E402

graph._AUTO_MAX_BACKWARD_ROWS = 1 << 20
DT = 2.0 ** -6
sde = problems.make("mlpdiag_ito", d=d).to(dev)
ts = torch.tensor([0.0, 8 * DT], device=dev)


def grads(entropy, opts):
    y0 = torch.full((B, d), 0.1, device=dev, requires_grad=True)
    bm = torchsde_amd.BrownianInterval(0.0, 8 * DT, size=(B, d), device=dev, dtype=torch.float32, entropy=entropy)
    ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=bm, method="euler", adjoint_method="euler", dt=DT,
                                     options=opts, adjoint_options=opts)
    sde.zero_grad()
    ys[-1].sum().backward()
    return [y0.grad] + [p.grad.clone() for p in sde.parameters()]


with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    worst = 0.0
    for entropy in range(1, 7):
        got, want_ = grads(entropy, None), grads(entropy, {"hip_graph": False})
        worst = max(worst, max(((a - e).abs().max() / e.abs().max().clamp_min(1e-30)).item() for a, e in zip(got, want_)))
print(f"adjoint 4096 x 128, no options, six iterations: worst relative gradient error {worst:.1e}")
for line in graph.describe_cache(sde):
    print("   ", line[:200])

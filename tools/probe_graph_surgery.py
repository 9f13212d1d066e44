"""Does rewriting a recorded graph's memset nodes as kernel nodes (csrc/graph_nodes.hip, graph._capturing) cure the
fault of tools/probe_graph_reduction4.py?
  A  20 x [memset node, kernel] with an eager memset + host synchronisation between replays, as recorded / rewritten;
  B  twenty multi-block column sums with an eager column sum + synchronisation between replays, as recorded / rewritten;
  C  the adjoint's backward sweep at 4096 x 128 and 32768 x 128 (row limit of "auto" lifted), rewritten: is the graph
     accepted, and are the gradients those of the eager sweep on every iteration?"""
import ctypes
import os
import sys
import warnings

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchsde_amd                                                     # noqa: E402
from torchsde_amd import graph                                          # noqa: E402
from workloads import problems                                          # noqa: E402

dev = "cuda"
torch.manual_seed(0)
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]


def record(fn, rewrite):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    graph._REWRITE_MEMSET_NODES = rewrite
    g = graph.new_graph()
    with graph._capturing(g, torch.device(dev)):
        outs = fn()
    graph._REWRITE_MEMSET_NODES = True
    return g, outs


counter = torch.zeros(256, dtype=torch.int32, device=dev)
elsewhere = torch.zeros(1 << 16, dtype=torch.int32, device=dev)


def memsets_and_kernels():
    acc = torch.zeros(256, device=dev)
    for k in range(20):
        stream = torch.cuda.current_stream().cuda_stream
        assert hip.hipMemsetAsync(counter.data_ptr(), 0, counter.numel() * 4, stream) == 0
        counter.add_(k + 1)
        acc = acc + counter
    return [acc]


def eager_memsets():
    stream = torch.cuda.current_stream().cuda_stream
    for k in range(8):
        assert hip.hipMemsetAsync(elsewhere.data_ptr() + 1024 * k, 1, 512, stream) == 0
    torch.cuda.synchronize()


for rewrite in (False, True):
    g, outs = record(memsets_and_kernels, rewrite)
    seen = []
    for _ in range(4):
        g.replay()
        seen.append(outs[0].max().item())
        eager_memsets()
    print(f"A  memset nodes (found, rewritten) = {g.memset_nodes}: acc after replays 1-4 (210 = right):", seen)
    del g, outs

B, d = 4096, 128
y = torch.randn(B, d, device=dev)


def twenty_sums():
    return [sum((y * float(k)).sum(0) for k in range(1, 21))]


want = twenty_sums()[0].clone()
for rewrite in (False, True):
    g, outs = record(twenty_sums, rewrite)
    kept = []
    for _ in range(4):
        g.replay()
        kept.append(outs[0].clone())
        y.sum(0)
        torch.cuda.synchronize()
    errs = [((k - want).abs().max() / want.abs().max()).item() for k in kept]
    print(f"B  memset nodes (found, rewritten) = {g.memset_nodes}: relative error of replays 1-4:",
          " ".join(f"{e:.1e}" for e in errs))
    del g, outs

graph._AUTO_MAX_BACKWARD_ROWS = 1 << 20
DT = 2.0 ** -6
for rows in (4096, 32768):
    sde = problems.make("mlpdiag_ito", d=d).to(dev)
    ts = torch.tensor([0.0, 8 * DT], device=dev)

    def grads(entropy, opts):
        y0 = torch.full((rows, d), 0.1, device=dev, requires_grad=True)
        bm = torchsde_amd.BrownianInterval(0.0, 8 * DT, size=(rows, d), device=dev, dtype=torch.float32, entropy=entropy)
        ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=bm, method="euler", adjoint_method="euler", dt=DT,
                                         options=opts, adjoint_options=opts)
        sde.zero_grad()
        ys[-1].sum().backward()
        return [y0.grad] + [p.grad.clone() for p in sde.parameters()]

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        worst = 0.0
        for entropy in range(1, 9):
            got, want_ = grads(entropy, None), grads(entropy, {"hip_graph": False})
            worst = max(worst, max(((a - e).abs().max() / e.abs().max().clamp_min(1e-30)).item()
                                   for a, e in zip(got, want_)))
    print(f"C  adjoint {rows} x {d}, no options, eight iterations: worst relative gradient error {worst:.1e}")
    for line in graph.describe_cache(sde):
        print("     ", line[:220])
    for entry in getattr(sde, graph._CACHE_ATTR).values():
        g = getattr(entry, "graph", None)
        if g is not None:
            print("      ", type(entry).__name__, "memset nodes (found, rewritten):", g.memset_nodes)

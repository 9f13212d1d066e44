"""THE reproducer of tools/REPORT_hip_graph_memset_nodes.md (needs only torch). Which part of a recorded graph goes wrong? torch's multi-block reductions
zero their block-counting semaphores with a `hipMemsetAsync` before every launch (ATen/native/cuda/Reduce.cuh): under
capture that is a MEMSET NODE, the only kind of node besides kernels in the graphs this package records.
  A  a graph of 20 x [memset node on a counter buffer, kernel that increments and accumulates it], replayed with a host
     synchronisation between replays: do the memset nodes still do their work?
  B  the twenty column sums with (i) nothing, (ii) a device synchronize, (iii) a device-to-host read between replays;
  C  node types of recorded graphs (hipGraphGetNodes / hipGraphNodeGetType on torch's raw graph): twenty column sums at
     4096 x 128 (multi-block) and at 1024 x 8 (one block per column group)."""
import collections
import ctypes
import os

import torch

dev = "cuda"
torch.manual_seed(0)
flags = {k: v for k, v in os.environ.items() if k.startswith(("DEBUG_CLR", "DEBUG_HIP", "HIP_FORCE"))}
print("flags:", flags or "(defaults)")
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipGraphGetNodes.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
hip.hipGraphNodeGetType.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
NODE_TYPES = {0: "kernel", 1: "memcpy", 2: "memset", 3: "host", 4: "graph", 5: "empty", 10: "alloc", 11: "free"}


def node_types(graph):
    raw = ctypes.c_void_p(graph.raw_cuda_graph())
    n = ctypes.c_size_t(0)
    assert hip.hipGraphGetNodes(raw, None, ctypes.byref(n)) == 0
    nodes = (ctypes.c_void_p * n.value)()
    assert hip.hipGraphGetNodes(raw, nodes, ctypes.byref(n)) == 0
    count = collections.Counter()
    for node in nodes:
        t = ctypes.c_int(-1)
        assert hip.hipGraphNodeGetType(ctypes.c_void_p(node), ctypes.byref(t)) == 0
        count[NODE_TYPES.get(t.value, t.value)] += 1
    return dict(count)


def record(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph(keep_graph=True)
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        outs = fn()
    return g, outs


# ---- A: do memset nodes keep working? -----------------------------------------------------------------------------------
counter = torch.zeros(256, dtype=torch.int32, device=dev)


def memsets_and_kernels():
    acc = torch.zeros(256, device=dev)
    for k in range(20):
        stream = torch.cuda.current_stream().cuda_stream
        assert hip.hipMemsetAsync(counter.data_ptr(), 0, counter.numel() * 4, stream) == 0
        counter.add_(k + 1)
        acc = acc + counter
    return [acc]


g, outs = record(memsets_and_kernels)
print("A  nodes:", node_types(g))
seen = []
for _ in range(4):
    g.replay()
    seen.append(outs[0].max().item())          # (a reduction, a device-to-host copy, a host synchronisation)
print("A  acc after replays 1-4 (210 = every memset node worked):", seen)
del g, outs

# ---- B: what between replays breaks the twenty column sums? -----------------------------------------------------------
B, d = 4096, 128
y = torch.randn(B, d, device=dev)


def twenty_sums():
    return [sum((y * float(k)).sum(0) for k in range(1, 21))]


want = twenty_sums()[0].clone()
scratch = torch.zeros(4, device=dev)
for tag, between in [("nothing", lambda: None), ("torch.cuda.synchronize()", torch.cuda.synchronize),
                     ("a device-to-host read", lambda: scratch.tolist()),
                     ("an eager column sum + synchronize", lambda: (y.sum(0), torch.cuda.synchronize()))]:
    g, outs = record(twenty_sums)
    kept = []
    for _ in range(4):
        g.replay()
        kept.append(outs[0].clone())
        between()
    torch.cuda.synchronize()
    errs = [((k - want).abs().max() / want.abs().max()).item() for k in kept]
    print(f"B  between replays: {tag:36s} relative error of replays 1-4:", " ".join(f"{e:.1e}" for e in errs))
    types = node_types(g)
    del g, outs
print("C  nodes of twenty column sums at 4096 x 128:", types)

small = torch.randn(1024, 8, device=dev)
g, outs = record(lambda: [sum((small * float(k)).sum(0) for k in range(1, 21))])
print("C  nodes of twenty column sums at 1024 x 8:  ", node_types(g))

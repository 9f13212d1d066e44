"""Minimal reproduction attempts of the non-idempotent captured backward: column reductions (B, d) -> (d,) inside a HIP
graph, as plain ops and through autograd, alone and among other allocations; replays compared with eager."""
import torch

dev = "cuda"
torch.manual_seed(0)
B, d = 4096, 128


def check(tag, fn, n_replays=3):
    want = [o.clone() for o in fn()]
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        outs = fn()
    res = []
    for _ in range(n_replays):
        g.replay()
        torch.cuda.synchronize()
        res.append(max(((o - w).abs().max() / w.abs().max().clamp_min(1e-30)).item() for o, w in zip(outs, want)))
    print(f"{tag:60s}", " ".join(f"{r:.1e}" for r in res))


y = torch.randn(B, d, device=dev)
w = torch.randn(d, device=dev, requires_grad=True)
b = torch.randn(d, device=dev, requires_grad=True)

check("plain sum(0), once", lambda: [(y * 2.0).sum(0)])
check("plain sum(0), 20 times in a row", lambda: [sum((y * float(k)).sum(0) for k in range(1, 21))])


def via_autograd(steps):
    def fn():
        acc_w = torch.zeros_like(w)
        acc_b = torch.zeros_like(b)
        yy = y
        for _ in range(steps):
            with torch.enable_grad():
                leaf = yy.detach().requires_grad_(True)
                g = 0.1 * torch.sigmoid(w * leaf + b)
                gy, gw, gb = torch.autograd.grad([g], [leaf, w, b], grad_outputs=[torch.ones_like(g)])
            acc_w = acc_w + gw
            acc_b = acc_b + gb
            yy = yy + 0.01 * gy
        return [acc_w, acc_b, yy]
    return fn


with torch.no_grad():
    check("autograd.grad of sigmoid(w*y+b), 1 step", via_autograd(1))
    check("autograd.grad of sigmoid(w*y+b), 20 steps", via_autograd(20))
torch.autograd.set_multithreading_enabled(False)
with torch.no_grad():
    check("the same, engine single-threaded", via_autograd(20))

"""Timing of BrownianInterval queries (aligned cell, multi-cell, misaligned) at C2 size."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import torchsde_amd  # noqa: E402

dev = "cuda"
B, m, dt = 65536, 64, 2.0 ** -10


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


for levy in ("none", "space-time"):
    bm = torchsde_amd.BrownianInterval(0.0, 1.0, size=(B, m), device=dev, dtype=torch.float32, entropy=1, dt=dt,
                                       levy_area_approximation=levy)
    W = torch.empty(B, m, device=dev)
    U = torch.empty(B, m, device=dev)
    wu = levy != "none"
    print(levy, "aligned 1 cell      %.1f us" % t(lambda: bm.increment(5 * dt, 6 * dt, want_U=wu, out_W=W, out_U=U)))
    print(levy, "aligned 8 cells     %.1f us" % t(lambda: bm.increment(8 * dt, 16 * dt, want_U=wu, out_W=W, out_U=U)))
    print(levy, "half cell (dyadic)  %.1f us" % t(lambda: bm.increment(5 * dt, 5.5 * dt, want_U=wu, out_W=W, out_U=U)))
    print(levy, "misaligned 2 cells  %.1f us" % t(lambda: bm.increment(5.3 * dt + 1e-7, 6.3 * dt + 1e-7, want_U=wu,
                                                                       out_W=W, out_U=U)))
    print(levy, "misaligned in-cell  %.1f us" % t(lambda: bm.increment(5.1 * dt + 1e-7, 5.7 * dt, want_U=wu,
                                                                       out_W=W, out_U=U)))


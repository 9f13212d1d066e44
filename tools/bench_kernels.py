"""Kernel-level timings through the C ABI: `iters` launches recorded into ONE HIP graph and replayed between a pair of
events (best of 3 replays), on random data. Round 2 timed a Python loop of eager launches instead, which measures the
host (6.8 us per ctypes launch) for every kernel shorter than that -- the "0.15 of peak" of the 2 MiB-per-stream rows was
that, not the kernel (profiles/r3_kernels_size_sweep.txt).

Two figures per step kernel: back to back on constant f, g (which a per-XCD L2 partly retains between launches at shard
sizes: an upper bound), and IN SITU -- f = mu*y and g = sigma*y recomputed by torch kernels before every launch, as in
a solve, reported as (triple - producers alone)."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from torchsde_amd import kernels as K  # noqa: E402
from torchsde_amd.kernels import NoiseSpec  # noqa: E402


def timeit(fn, iters=200):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(10):
            fn(i)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for i in range(iters):
            fn(i)
    graph.replay()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e3 / iters


def gen_noise(shape, cell, dt, dev):
    return NoiseSpec(shape, torch.float32, torch.device(dev), entropy=12345, elem0=0, cell=cell, h=dt)


def main():
    dev = "cuda"
    dt = 2.0 ** -10
    rows = []
    # the last two exceed the 256 MiB Infinity Cache (4 x 64 MiB and 4 x 256 MiB live streams): true HBM traffic
    for (B, d) in [(65536, 64), (32768, 64), (32768, 128), (16384, 32), (262144, 64), (1048576, 64)]:
        y = [torch.rand(B, d, device=dev) for _ in range(2)]
        f, g = torch.randn(B, d, device=dev), torch.rand(B, d, device=dev)
        specs = [gen_noise((B, d), i, dt, dev) for i in range(200)]
        us = timeit(lambda i: K._raw_step_diag(y[i & 1], f, g, dt, 1.0, specs[i], y[(i + 1) & 1]))
        rows.append((f"step_diag B={B} d={d}", us, 16 * B * d))
        mu, sigma = -torch.rand(d, device=dev), torch.rand(d, device=dev)

        def producers(i):
            torch.mul(y[i & 1], mu, out=f)
            torch.mul(y[i & 1], sigma, out=g)

        def triple(i):
            producers(i)
            K._raw_step_diag(y[i & 1], f, g, dt, 1.0, specs[i], y[(i + 1) & 1])
        rows.append((f"step_diag B={B} d={d} IN SITU (after f = mu*y, g = sigma*y)", timeit(triple) - timeit(producers),
                     16 * B * d))
        gdg = torch.randn(B, d, device=dev)
        us = timeit(lambda i: K._raw_milstein_diag(y[i & 1], f, g, gdg, dt, specs[i], y[(i + 1) & 1]))
        rows.append((f"milstein_diag B={B} d={d}", us, 20 * B * d))
    for (B, d, m) in [(16384, 32, 16), (16384, 64, 16), (65536, 16, 16), (16384, 32, 64), (65536, 32, 16), (4096, 32, 16)]:
        y = [torch.rand(B, d, device=dev) for _ in range(2)]
        f, g = torch.randn(B, d, device=dev), torch.rand(B, d, m, device=dev)
        specs = [gen_noise((B, m), i, dt, dev) for i in range(200)]
        us = timeit(lambda i: K._raw_step_general(y[i & 1], f, g, dt, 1.0, specs[i], y[(i + 1) & 1]))
        rows.append((f"step_general B={B} d={d} m={m}", us, 4 * B * (d * m + 3 * d)))
    # batch-broadcast diffusion on the matrix cores: reads y0, f, writes y1 (12*d bytes per row); operands rotate over
    # enough copies that no launch finds its inputs in L2
    for (B, d, m) in [(16384, 32, 16), (65536, 64, 16), (262144, 64, 32), (65536, 128, 64), (1048576, 32, 16)]:
        k = max(2, min(16, (128 << 20) // (8 * B * d)))
        ys = [torch.rand(B, d, device=dev) for _ in range(k + 1)]
        fs = [torch.randn(B, d, device=dev) for _ in range(k)]
        S = torch.randn(d, m, device=dev) / m ** 0.5
        specs = [gen_noise((B, m), i, dt, dev) for i in range(200)]
        us = timeit(lambda i: K._raw_step_shared(ys[i % k], fs[i % k], S, 1.0, dt, 1.0, 0, 0.0, 0.0, 0.0, specs[i],
                                                 ys[i % k + 1]))
        rows.append((f"step_shared (MFMA) B={B} d={d} m={m}", us, 12 * B * d))
    B, d = 32768, 128
    s = [torch.rand(B, d, device=dev) for _ in range(4)]
    F = [torch.randn(B, d, device=dev) for _ in range(4)]
    P = [torch.randn(128, 128, device=dev) for _ in range(4)] + [torch.randn(128, device=dev) for _ in range(4)]

    def aug(i):
        segs = [dict(out=s[2], s=s[0], F=F[0], G=F[1], sF=-1.0, sG=-1.0), dict(out=s[3], s=s[1], F=F[2], G=F[3])]
        segs += [dict(out=p, s=p, F=p, G=p) for p in P]
        K.aug_update(segs, dt, 1.0, torch.float32, torch.device(dev))
    us = timeit(aug)
    rows.append((f"aug_update B={B} d={d} (+8 param segments)", us, 32 * B * d))
    for name, us, nbytes in rows:
        print(f"{name:72s} {us:8.2f} us  {nbytes / us / 1e3:8.1f} GB/s  ({nbytes / us / 1e3 / 80:.1f} % of 8 TB/s)")


if __name__ == "__main__":
    main()

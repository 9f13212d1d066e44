"""Weight-gradient GEMM of a batch-32768 Linear(128,128) (K = batch): library default vs TunableOp pick (run on the GPU box).

    python tools/bench_tunable_gemm.py            # default heuristics
    PYTORCH_TUNABLEOP_ENABLED=1 python tools/bench_tunable_gemm.py
"""
import time

import torch

dev = torch.device("cuda")
B, d = 32768, 128
x = torch.randn(B, d, device=dev)
delta = torch.randn(B, d, device=dev)
w = torch.randn(d, d, device=dev)
cases = {
    "dW = delta^T x   (128 x 32768 x 128)": lambda: delta.t().mm(x),
    "dX = delta W     (32768 x 128 x 128)": lambda: delta.mm(w),
    "fwd x W^T + b    (32768 x 128 x 128)": lambda: torch.nn.functional.linear(x, w),
}
for name, fn in cases.items():
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(200):
        fn()
    torch.cuda.synchronize()
    us = (time.perf_counter() - t) / 200 * 1e6
    print(f"{name}: {us:7.1f} us  ({2.0 * B * d * d / us / 1e6:6.1f} TFLOP/s)   tunable={torch.cuda.tunable.is_enabled()}")

"""Host-side cost of one small-batch training iteration through the perceptron-drift trajectory kernels (cProfile)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import torchsde_amd as torchsde  # noqa: E402

device = "cuda"
batch, d, hidden, steps = 1024, 32, 64, 64
torch.manual_seed(0)
sde = torchsde.MLPDriftDiagonalSDE(d, hidden, activation="tanh", diff_rate=0.0, diff_shift=0.3).to(device)
ts = torch.tensor([0.0, 1.0], device=device)
y0 = torch.randn(batch, d, device=device)


def iteration(it):
    bm = torchsde.BrownianInterval(0.0, 1.0, size=(batch, d), device=device, dtype=torch.float32, entropy=it)
    ys = torchsde.sdeint(sde, y0, ts, bm=bm, method="euler", dt=1.0 / steps)
    loss = (ys[-1] ** 2).mean()
    sde.zero_grad()
    loss.backward()


for it in range(10):
    iteration(it)
torch.cuda.synchronize()
t = time.perf_counter()
for it in range(100):
    iteration(it)
host = time.perf_counter() - t
torch.cuda.synchronize()
total = time.perf_counter() - t
print(f"per iteration: host {host * 10:.3f} ms, host+gpu {total * 10:.3f} ms")
prof = cProfile.Profile()
prof.enable()
for it in range(100):
    iteration(it)
prof.disable()
pstats.Stats(prof).sort_stats("cumulative").print_stats(35)

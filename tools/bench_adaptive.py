import sys, time, torch
sys.path.insert(0, ".")
import torchsde_amd
from workloads import problems
dev = "cuda"
for (B, d) in ((1024, 16), (65536, 64)):
    sde = problems.make("gbm_ito", d=d).to(dev)
    y0 = torch.full((B, d), 0.1, device=dev)
    ts = torch.tensor([0.0, 1.0], device=dev)
    def solve(i):
        bm = torchsde_amd.BrownianInterval(0.0, 1.0, size=(B, d), device=dev, dtype=torch.float32, entropy=i)
        with torch.no_grad():
            return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="milstein", dt=0.05, adaptive=True, rtol=1e-3, atol=1e-4)
    solve(0); torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(3): out = solve(1 + i)
    torch.cuda.synchronize()
    print(B, d, "adaptive milstein", (time.perf_counter() - t) / 3 * 1e3, "ms")

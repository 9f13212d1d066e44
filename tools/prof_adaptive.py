import sys, time, torch, cProfile, pstats
sys.path.insert(0, "/root/repo")
import torchsde_amd
from workloads import problems
dev = "cuda"
B, d = 1024, 16
sde = problems.make("gbm_ito", d=d).to(dev)
y0 = torch.full((B, d), 0.1, device=dev)
ts = torch.tensor([0.0, 1.0], device=dev)
def solve(i):
    bm = torchsde_amd.BrownianInterval(0.0, 1.0, size=(B, d), device=dev, dtype=torch.float32, entropy=i)
    with torch.no_grad():
        return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="milstein", dt=0.05, adaptive=True, rtol=1e-3, atol=1e-4)
solve(0); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); solve(1); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)

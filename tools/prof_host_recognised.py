import cProfile, pstats, io, sys, os
sys.path.insert(0, os.getcwd())
import torch, torchsde_amd
from workloads import problems
B,d,steps,dt=1024,8,200,2.0**-10
sde=problems.make("gbm_ito",d=d).to("cuda")
y0=torch.full((B,d),0.1,device="cuda"); ts=torch.tensor([0.0,steps*dt],device="cuda")
def solve(i):
    bm=torchsde_amd.BrownianInterval(0.0,steps*dt,size=(B,d),device="cuda",entropy=i)
    with torch.no_grad(): return torchsde_amd.sdeint(sde,y0,ts,bm=bm,method="euler",dt=dt)
for i in range(10): solve(i)
torch.cuda.synchronize()
pr=cProfile.Profile(); pr.enable()
for i in range(300): solve(100+i)
torch.cuda.synchronize(); pr.disable()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("cumulative").print_stats(45); print(s.getvalue()[:7000])

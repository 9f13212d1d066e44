"""Outputs of the stepwise adjoint's backward sweep (latent SDE, Ito diagonal, Euler) run eagerly and as a sequential HIP
graph, with zero and with all-ones cotangents, replayed several times: are the diffusion parameters' gradients stable?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import torchsde_amd  # noqa: E402
from torchsde_amd import adjoint, graph, timegrid  # noqa: E402
from workloads.configs import make_problem  # noqa: E402

dev = torch.device("cuda", 0)
B, d, n, dt = 4096, 128, 20, 2.0 ** -9
sde = make_problem("latent_diag", d, d, dev)
from torchsde_amd.sde import ForwardSDE  # noqa: E402
fsde = ForwardSDE(sde)
fsde.overlap_f_g = False
y0 = torch.full((B, d), 0.1, device=dev)
ts = torch.tensor([0.0, n * dt], device=dev)
bm = torchsde_amd.BrownianInterval(0.0, n * dt, size=(B, d), device=dev, dtype=torch.float32, entropy=5, dt=dt)
with torch.no_grad():
    ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="euler", dt=dt, options={"hip_graph": False})
params = [p for p in sde.parameters()]


def show(tag, outs):
    print(tag, " ".join(f"{o.abs().max().item():.3e}" for o in outs))


for fill, name in ((torch.zeros_like, "zero cotangents"), (torch.ones_like, "ones cotangents")):
    print("==", name)
    run = adjoint._backward_runner(fsde, bm, dt, "euler", params, timegrid.ts_to_host(ts), dev)
    with torch.no_grad():
        show("eager        ", run(ys, fill(ys)))
        show("eager again  ", run(ys, fill(ys)))
    from torch.nn.utils.stateless import _reparametrize_module
    alias_of = {id(p): p.detach().requires_grad_(True) for p in params}
    swapped = {nm: alias_of[id(p)] for nm, p in fsde.named_parameters(remove_duplicate=False) if id(p) in alias_of}
    run2 = adjoint._backward_runner(fsde, bm, dt, "euler", [alias_of[id(p)] for p in params], timegrid.ts_to_host(ts), dev)
    with torch.no_grad(), _reparametrize_module(fsde, swapped):
        cap = graph._CapturedBackward(run2, bm, [ys, fill(ys)], keepalive=(run2.plan,))
    for k in range(4):
        cap.graph.replay()
        torch.cuda.synchronize()
        show(f"graph replay {k}", cap.out)

"""Bisecting the non-idempotent backward graph (tools/probe_backward_graph.py): variants of the sweep, each captured as
a sequential graph and replayed 3 times with all-ones cotangents; prints max |replay_k - eager| / scale per output."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from torch.nn.utils.stateless import _reparametrize_module  # noqa: E402

import torchsde_amd  # noqa: E402
from torchsde_amd import adjoint, graph, timegrid  # noqa: E402
from torchsde_amd.sde import ForwardSDE  # noqa: E402
from workloads import problems  # noqa: E402

dev = torch.device("cuda", 0)


def trial(tag, B, d, n, sde, kind="euler", params=None, single_thread=False):
    dt = 2.0 ** -9
    fsde = ForwardSDE(sde)
    fsde.overlap_f_g = False
    y0 = torch.full((B, d), 0.1, device=dev)
    ts = torch.tensor([0.0, n * dt], device=dev)
    bm = torchsde_amd.BrownianInterval(0.0, n * dt, size=(B, d), device=dev, dtype=torch.float32, entropy=5, dt=dt)
    with torch.no_grad():
        ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="euler" if sde.sde_type == "ito" else "midpoint", dt=dt,
                                 options={"hip_graph": False})
    params = [p for p in sde.parameters()] if params is None else params
    ctx = torch.autograd.set_multithreading_enabled(False) if single_thread else torch.no_grad()
    with ctx:
        run = adjoint._backward_runner(fsde, bm, dt, kind, params, timegrid.ts_to_host(ts), dev)
        with torch.no_grad():
            want = [o.clone() for o in run(ys, torch.ones_like(ys))]
        alias_of = {id(p): p.detach().requires_grad_(True) for p in params}
        swapped = {nm: alias_of[id(p)] for nm, p in fsde.named_parameters(remove_duplicate=False) if id(p) in alias_of}
        run2 = adjoint._backward_runner(fsde, bm, dt, kind, [alias_of[id(p)] for p in params], timegrid.ts_to_host(ts), dev)
        with torch.no_grad(), _reparametrize_module(fsde, swapped):
            cap = graph._CapturedBackward(run2, bm, [ys, torch.ones_like(ys)], keepalive=(run2.plan,))
        rows = []
        for k in range(3):
            cap.graph.replay()
            torch.cuda.synchronize()
            rows.append(" ".join(f"{((o - w).abs().max() / w.abs().max().clamp_min(1e-30)).item():.1e}"
                                 for o, w in zip(cap.out, want)))
    print(f"{tag:44s} replay0 [{rows[0]}]\n{'':44s} replay1 [{rows[1]}]\n{'':44s} replay2 [{rows[2]}]")


from workloads.configs import make_problem  # noqa: E402
latent = make_problem("latent_diag", 128, 128, dev)
trial("latent d128 B4096 (repro)", 4096, 128, 20, latent)
trial("latent, engine single-threaded", 4096, 128, 20, latent, single_thread=True)
trial("latent, only the net's parameters", 4096, 128, 20, latent, params=list(latent.net.parameters()))
trial("latent, only w and b", 4096, 128, 20, latent, params=[latent.w, latent.b])
trial("latent B512", 512, 128, 20, latent)
trial("latent 2 steps", 4096, 128, 2, latent)
trial("latent 1 step", 4096, 128, 1, latent)
trial("mlpdiag_ito d128 B4096", 4096, 128, 20, problems.make("mlpdiag_ito", d=128).to(dev))
trial("mlpdiag_strat d128 B4096 midpoint", 4096, 128, 20, problems.make("mlpdiag_strat", d=128).to(dev), kind="midpoint")
trial("gbm_ito d128 B4096", 4096, 128, 20, problems.make("gbm_ito", d=128).to(dev))

"""Back-propagation THROUGH the solver (plain `sdeint` with tensors that require grad): the fused forward kernels
carry autograd wrappers whose backward is written from the step formulas. Checked against torch autograd through
the oracle's restatement of the reference's steps (CPU, float64) on the same Brownian path (C twin of the generator)."""
import numpy as np
import pytest
import torch

from oracle import counter, solvers_ref
from workloads import problems

pytestmark = pytest.mark.gpu
DEV = "cuda"

CASES = [
    ("gbm_ito", "euler", None, (16, 4, 4)),
    ("gbm_ito", "milstein", None, (16, 4, 4)),
    ("gbm_ito", "milstein", {"grad_free": True}, (16, 4, 4)),
    ("gbm_strat", "midpoint", None, (16, 4, 4)),
    ("gbm_strat", "heun", None, (16, 4, 4)),
    ("gbm_strat", "euler_heun", None, (16, 4, 4)),
    ("gbm_strat", "reversible_heun", None, (16, 4, 4)),
    ("general_ito", "euler", None, (12, 4, 4)),
    ("general_strat", "midpoint", None, (12, 4, 4)),
    ("general_strat", "reversible_heun", None, (12, 4, 4)),
    ("scalar_ito", "milstein", None, (12, 4, 1)),
    ("additive_ito", "euler", None, (12, 4, 3)),
    ("gbm_ito", "srk", None, (16, 4, 4)),            # SRID2 stages: differentiable torch twin of the stage kernels
    ("scalar_ito", "srk", None, (12, 4, 1)),
    ("additive_ito", "srk", None, (12, 4, 3)),       # SRA1
]


@pytest.mark.parametrize("prob,method,options,shape", CASES)
def test_backprop_through_solver_matches_oracle(prob, method, options, shape):
    import torchsde_amd
    B, d, m = shape
    dtype = torch.float64
    steps, dt = 8, 2.0 ** -4
    ts_list = [0.0, 3 * dt, steps * dt]
    edges = np.arange(steps + 1) * dt
    levy = method == "srk"

    def loss_and_grads(device):
        sde = problems.make(prob, dtype=dtype, d=d, m=m).to(device)
        y0 = torch.full((B, d), 0.1, dtype=dtype, device=device, requires_grad=True)
        ts = torch.tensor(ts_list, dtype=dtype, device=device)
        if device == DEV:
            bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(B, m), dtype=dtype, device=DEV, entropy=31,
                                               dt=dt, levy_area_approximation="space-time" if levy else "none")
            ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=dt,
                                     options=None if options is None else dict(options))
        else:
            def bm_cpu(ta, tb, return_U=False):
                W, U, _ = counter.query(B * m, 31, edges, float(ta), float(tb), dtype=np.float64, have_h=levy)
                W = torch.from_numpy(W).reshape(B, m)
                return (W, torch.from_numpy(U).reshape(B, m)) if return_U else W
            if method == "reversible_heun":
                ys, _ = solvers_ref.integrate_reversible_heun(sde, bm_cpu, y0, ts, dt)
            else:
                ys = solvers_ref.integrate(sde, bm_cpu, y0, ts, dt, method, options)
        weights = torch.linspace(-1, 1, ys.numel(), dtype=dtype, device=device).reshape(ys.shape)
        (ys * weights).sum().backward()
        return ys.detach().cpu(), y0.grad.cpu(), [p.grad.cpu() for p in sde.parameters()]

    ys_g, gy_g, gp_g = loss_and_grads(DEV)
    ys_c, gy_c, gp_c = loss_and_grads("cpu")
    torch.testing.assert_close(ys_g, ys_c, rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(gy_g, gy_c, rtol=1e-8, atol=1e-10)
    for a, b in zip(gp_g, gp_c):
        torch.testing.assert_close(a, b, rtol=1e-8, atol=1e-9)


def test_shape_edge_cases():
    """Single output time, batch of one, state sizes that are not multiples of 4, non-contiguous y0, ts given as a list."""
    import torchsde_amd
    with torch.no_grad():
        sde = problems.make("gbm_ito", d=5).to(DEV)
        y0 = torch.full((1, 5), 0.1, device=DEV)
        ys = torchsde_amd.sdeint(sde, y0, torch.tensor([0.3], device=DEV), method="euler", dt=0.1)
        assert ys.shape == (1, 1, 5) and torch.equal(ys[0], y0)
        ys = torchsde_amd.sdeint(sde, y0, [0.0, 0.25], method="srk", dt=0.1)
        assert ys.shape == (2, 1, 5) and torch.isfinite(ys).all()
        sde = problems.make("gbm_ito", d=6).to(DEV)
        wide = torch.rand(7, 12, device=DEV) * 0.1
        y0 = wide[:, ::2]                                   # non-contiguous view
        a = torchsde_amd.sdeint(sde, y0, [0.0, 0.5], method="milstein", dt=0.125,
                                bm=torchsde_amd.BrownianInterval(0.0, 0.5, size=(7, 6), device=DEV,
                                                                 dtype=torch.float32, entropy=2))
        b = torchsde_amd.sdeint(sde, y0.contiguous(), [0.0, 0.5], method="milstein", dt=0.125,
                                bm=torchsde_amd.BrownianInterval(0.0, 0.5, size=(7, 6), device=DEV,
                                                                 dtype=torch.float32, entropy=2))
        assert torch.equal(a, b)

"""The whole-trajectory kernel for elementwise-expression SDEs (``tsde_trajectory_expr_diag``,
torchsde_amd.ElementwiseDiagonalSDE; ``-m gpu``) against

* the REAL reference solving the same module on the same Brownian path in float64
  (tests/golden/closed_form_expr_*.npz -- three of them the SDE of the reference's own benchmark, f = y, g = exp(-y));
* the stepwise path of this package on the same module (`options={"trajectory_kernel": False}`: torch evaluates f and g
  between the per-step kernels), for every scheme x function pair.
"""
import os

import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cases():
    return sorted(f[len("closed_form_expr_"):-4] for f in os.listdir(helpers.GOLDEN) if f.startswith("closed_form_expr_"))


def _module(z, dtype):
    import torchsde_amd
    coefs = [torch.tensor(z["param__" + n]) for n in torchsde_amd.ElementwiseDiagonalSDE._NAMES]
    return torchsde_amd.ElementwiseDiagonalSDE(str(z["drift"]), str(z["diffusion"]), coefs[:4], coefs[4:],
                                               sde_type=str(z["sde_type"]), dtype=dtype).to(DEV)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("name", _cases())
def test_matches_the_reference(name, dtype):
    import torchsde_amd
    z = helpers.load(f"closed_form_expr_{name}.npz")
    B, d, steps = (int(v) for v in z["shape"])
    dt, levy = float(z["dt"]), str(z["levy"])
    sde = _module(z, dtype)
    y0 = torch.tensor(z["y0"], dtype=dtype, device=DEV)
    ts = torch.tensor(z["ts"], dtype=dtype, device=DEV)

    def solve(trajectory):
        bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(B, d), dtype=dtype, device=DEV,
                                           entropy=int(z["entropy"]), dt=dt, levy_area_approximation=levy)
        with torch.no_grad():
            return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=str(z["method"]), dt=dt,
                                       options={"trajectory_kernel": trajectory})
    fast, stepwise = solve(True), solve(False)
    want = torch.tensor(z["ys"])
    tol = dict(rtol=1e-9, atol=1e-11) if dtype == torch.float64 else dict(rtol=2e-4, atol=2e-5)
    torch.testing.assert_close(fast.double().cpu(), want, **tol)
    torch.testing.assert_close(stepwise.double().cpu(), want, **tol)
    # and the two routes agree far more closely with each other: same increments, same operation order, phi evaluated
    # by the same library functions
    close = dict(rtol=1e-13, atol=1e-14) if dtype == torch.float64 else dict(rtol=2e-6, atol=2e-7)
    torch.testing.assert_close(fast, stepwise, **close)


FUNCTIONS = ["identity", "exp", "sigmoid", "tanh", "softplus", "sin", "cos"]


@pytest.mark.parametrize("method,sde_type,levy", [("euler", "ito", "none"), ("milstein", "ito", "none"),
                                                  ("milstein", "stratonovich", "none"),
                                                  ("midpoint", "stratonovich", "none"), ("srk", "ito", "space-time")])
@pytest.mark.parametrize("diffusion", FUNCTIONS)
def test_every_function_and_scheme_against_the_stepwise_path(diffusion, method, sde_type, levy):
    """Each phi as the diffusion (Milstein also exercises phi'), with the next one of the list as the drift; vector and
    scalar lanes (d = 8 / d = 5), outputs on and off the step grid, float32."""
    import torchsde_amd
    drift = FUNCTIONS[(FUNCTIONS.index(diffusion) + 1) % len(FUNCTIONS)]
    for (B, d) in ((8192, 8), (33, 5)):
        gen = torch.Generator().manual_seed(d)
        rnd = lambda lo, hi: lo + (hi - lo) * torch.rand(d, generator=gen)   # noqa: E731
        # (an exponential diffusion with a steep rate runs away under explicit schemes: keep its argument gentle)
        steep = 0.4 if "exp" in (drift, diffusion) else 1.0
        sde = torchsde_amd.ElementwiseDiagonalSDE(drift, diffusion, (rnd(-0.5, 0.5), rnd(0.5, 1.2), rnd(-0.2, 0.2), 0.05),
                                                  (rnd(0.1, 0.4), rnd(-steep, steep), rnd(-0.2, 0.2), 0.05),
                                                  sde_type=sde_type).to(DEV)
        y0 = (0.4 * torch.rand(B, d, generator=gen) - 0.2).to(DEV)
        ts = torch.tensor([0.0, 0.1, 0.26, 0.5], device=DEV)

        def solve(trajectory, rows=slice(None), row_offset=0):
            bm = torchsde_amd.BrownianInterval(0.0, 0.5, size=tuple(y0[rows].shape), dtype=torch.float32, device=DEV,
                                               entropy=6, levy_area_approximation=levy, row_offset=row_offset)
            with torch.no_grad():
                return torchsde_amd.sdeint(sde, y0[rows], ts, bm=bm, method=method, dt=0.05,
                                           options={"trajectory_kernel": trajectory})
        fast, stepwise = solve(True), solve(False)
        assert torch.isfinite(fast).all()
        torch.testing.assert_close(fast, stepwise, rtol=5e-6, atol=5e-7)
        if B > 64:      # rows solved alone, with their global offset, are the rows of the full solve
            part = solve(True, rows=slice(4096, 4096 + 512), row_offset=4096)
            assert torch.equal(part, fast[:, 4096:4096 + 512])


def test_autograd_and_subclasses_take_the_stepwise_path():
    import torchsde_amd
    sde = torchsde_amd.ElementwiseDiagonalSDE("tanh", "sigmoid", (0.5, 1.0, 0.0, 0.0), (0.3, 1.0, 0.0, 0.1)).to(DEV)
    y0 = torch.full((64, 8), 0.1, device=DEV, requires_grad=True)
    ts = torch.tensor([0.0, 0.5], device=DEV)
    bm = torchsde_amd.BrownianInterval(0.0, 0.5, size=(64, 8), dtype=torch.float32, device=DEV, entropy=2)
    ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="euler", dt=0.05)
    ys[-1].sum().backward()
    assert torch.isfinite(y0.grad).all() and sde.f_scale.grad is not None and torch.isfinite(sde.g_rate.grad).all()

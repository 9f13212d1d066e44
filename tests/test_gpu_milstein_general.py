"""General-noise Milstein (opt-in extension; the reference rejects it, milstein.py:25). No reference oracle exists,
so it is pinned by the reduction tests of SURVEY.md section 8, note N1."""
import math

import pytest
import torch
from scipy.stats import linregress
from torch import nn

from workloads import problems

pytestmark = pytest.mark.gpu
DEV = "cuda"
F64 = torch.float64


class DiagAsGeneral(nn.Module):
    """GBM written as a general-noise SDE: g[b, i, j] = delta_ij sigma_i y_i."""
    noise_type = "general"

    def __init__(self, base):
        super().__init__()
        self.base, self.sde_type = base, base.sde_type

    def f(self, t, y):
        return self.base.f(t, y)

    def g(self, t, y):
        return torch.diag_embed(self.base.g(t, y))


@pytest.mark.parametrize("sde_type", ["ito", "stratonovich"])
def test_reduces_to_diagonal_milstein(sde_type):
    """(i) diagonal g embedded as general == the reference-backed diagonal Milstein on the same increments."""
    import torchsde_amd
    B, d, steps, dt = 32, 4, 16, 2.0 ** -5
    base = problems.GBMDiag(d, sde_type, dtype=F64).to(DEV)
    y0 = torch.full((B, d), 0.1, dtype=F64, device=DEV)
    ts = torch.tensor([0.0, steps * dt], dtype=F64, device=DEV)
    kw = dict(t0=0.0, t1=steps * dt, size=(B, d), dtype=F64, device=DEV, entropy=5, dt=dt)
    with torch.no_grad():
        ref = torchsde_amd.sdeint(base, y0, ts, bm=torchsde_amd.BrownianInterval(**kw), method="milstein", dt=dt)
        gen = torchsde_amd.sdeint(DiagAsGeneral(base), y0, ts, bm=torchsde_amd.BrownianInterval(**kw),
                                  method="milstein", dt=dt, options={"general_noise": True})
    torch.testing.assert_close(gen, ref, rtol=1e-10, atol=1e-12)


def test_requires_opt_in():
    import torchsde_amd
    sde = problems.make("general_ito").to(DEV)
    with pytest.raises(ValueError, match="only supports noise types"):
        torchsde_amd.sdeint(sde, torch.zeros(2, 4, device=DEV), [0.0, 0.1], method="milstein", dt=0.05)


def test_levy_term_equals_dg_ga_jvp_column_sum():
    """(iii) one step: y1 - Euler step == dg_ga_jvp_column_sum(t0, y0, I) with I = (W W^T - dt Id)/2 + A."""
    import torchsde_amd
    from torchsde_amd.sde import ForwardSDE
    B, d, m, dt = 16, 4, 4, 2.0 ** -4
    sde = problems.make("general_ito", dtype=F64, d=d, m=m).to(DEV)
    y0 = torch.full((B, d), 0.1, dtype=F64, device=DEV)
    ts = torch.tensor([0.0, dt], dtype=F64, device=DEV)
    kw = dict(t0=0.0, t1=dt, size=(B, m), dtype=F64, device=DEV, entropy=9, dt=dt, levy_area_approximation="foster")
    with torch.no_grad():
        mil = torchsde_amd.sdeint(sde, y0, ts, bm=torchsde_amd.BrownianInterval(**kw), method="milstein", dt=dt,
                                  options={"general_noise": True})
        eul = torchsde_amd.sdeint(sde, y0, ts, bm=torchsde_amd.BrownianInterval(**kw), method="euler", dt=dt)
    W, _, A = torchsde_amd.BrownianInterval(**kw).increment_with_levy_area(0.0, dt)
    integrals = 0.5 * (W.unsqueeze(-1) * W.unsqueeze(-2) - dt * torch.eye(m, dtype=F64, device=DEV)) + A
    expected = ForwardSDE(sde).dg_ga_jvp_column_sum(ts[0], y0, integrals)
    torch.testing.assert_close(mil[-1] - eul[-1], expected.detach(), rtol=1e-9, atol=1e-12)


def test_strong_order_beats_euler():
    """(iv) on a commutative general-noise SDE (diagonal embedded) the slope is ~1.0 vs ~0.5 for Euler."""
    import torchsde_amd
    B, d, t1 = 4096, 4, 1.0
    base = problems.GBMDiag(d, "ito", dtype=F64).to(DEV)
    sde = DiagAsGeneral(base)
    y0 = torch.full((B, d), 0.1, dtype=F64, device=DEV)
    ts = torch.tensor([0.0, t1], dtype=F64, device=DEV)
    bm = torchsde_amd.BrownianInterval(0.0, t1, size=(B, d), dtype=F64, device=DEV, entropy=271, dt=2.0 ** -8)
    exact = base.exact(y0, t1, bm(0.0, t1))
    slopes = {}
    for method, opts in (("euler", None), ("milstein", {"general_noise": True})):
        xs, ys_ = [], []
        with torch.no_grad():
            for k in range(3, 9):
                dt = 2.0 ** -k
                out = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=dt, options=opts)
                xs.append(math.log(dt))
                ys_.append(0.5 * math.log(((out[-1] - exact) ** 2).sum(1).mean().item()))
        slopes[method] = linregress(xs, ys_).slope
    assert abs(slopes["milstein"] - 1.0) < 0.15 and slopes["euler"] < 0.75, slopes

"""The remaining tests of the reference's own suite, re-pointed at this package (run with ``-m gpu``):

* tests/test_adjoint.py:45-94   `test_against_numerical` -- adjoint gradients vs finite differences;
* tests/test_adjoint.py:157-185 `test_basic`             -- adjoint on SDEs with unused / frozen parameters;
* tests/test_sdeint.py:160-215  `test_sdeint_dependencies` -- solvers on SDEs that ignore states / parameters.
"""
import pytest
import torch

from workloads import problems

pytestmark = pytest.mark.gpu
DEV = "cuda"
KINDS = ["state", "params", "frozen", "constant"]


@pytest.mark.parametrize("prob,method", [
    ("gbm_ito", "milstein"), ("gbm_ito", "srk"), ("gbm_strat", "midpoint"), ("gbm_strat", "reversible_heun"),
    ("scalar_ito", "milstein"), ("scalar_ito", "srk"), ("scalar_strat", "midpoint"),
    ("scalar_strat", "reversible_heun"),
    ("additive_ito", "milstein"), ("additive_ito", "srk"), ("additive_strat", "midpoint"),
    ("additive_strat", "reversible_heun"),
    ("general_strat", "midpoint"), ("general_strat", "reversible_heun"),
])
def test_adjoint_against_numerical(prob, method):
    """Directional finite differences of L(theta) = mean_b sum_i y_T^2 on ONE Brownian path (the generator is
    re-queried, so every evaluation sees the same path) vs <grad_adjoint, direction>. Tolerances of the reference:
    1e-2 for the continuous adjoint, 1e-6 for the exact reversible-Heun adjoint (here 1e-5 against an O(eps^2)
    central difference)."""
    import torchsde_amd
    B, d = 4, 3
    m = {"gbm": d, "scalar": 1, "additive": 2, "general": 2}[prob.split("_")[0]]
    dtype = torch.float64
    sde = problems.make(prob, dtype=dtype, d=d, m=m).to(DEV)
    ts = torch.tensor([0.0, 0.5], dtype=dtype, device=DEV)
    y0 = torch.full((B, d), 0.1, dtype=dtype, device=DEV)
    levy = "space-time" if method == "srk" else "none"
    bm = torchsde_amd.BrownianInterval(0.0, 0.5, size=(B, m), dtype=dtype, device=DEV, entropy=21,
                                       levy_area_approximation=levy)
    exact = method == "reversible_heun"
    adjoint_method = "adjoint_reversible_heun" if exact else None
    dt = 2.0 ** -9

    def loss():
        ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=bm, dt=dt, method=method, adjoint_method=adjoint_method)
        return (ys[-1] ** 2).sum(dim=1).mean(dim=0)

    params = [p for p in sde.parameters() if p.requires_grad]
    value = loss()
    grads = torch.autograd.grad(value, params, allow_unused=True)
    grads = [torch.zeros_like(p) if g is None else g for g, p in zip(grads, params)]
    gen = torch.Generator().manual_seed(0)
    for _ in range(2):
        direction = [torch.randn(p.shape, generator=gen, dtype=dtype).to(DEV) for p in params]
        eps = 1e-5
        with torch.no_grad():
            for p, v in zip(params, direction):
                p.add_(eps * v)
            up = loss()
            for p, v in zip(params, direction):
                p.sub_(2 * eps * v)
            down = loss()
            for p, v in zip(params, direction):
                p.add_(eps * v)
        numerical = ((up - down) / (2 * eps)).item()
        analytic = sum((g * v).sum() for g, v in zip(grads, direction)).item()
        tol = 1e-5 if exact else 1e-2
        assert abs(numerical - analytic) <= tol + tol * abs(numerical), (numerical, analytic)


@pytest.mark.parametrize("adaptive", [False, True])
@pytest.mark.parametrize("method,options", [("euler", {}), ("milstein", {}), ("milstein", {"grad_free": True}),
                                            ("srk", {})])
@pytest.mark.parametrize("kind", KINDS)
def test_sdeint_dependencies(kind, method, options, adaptive):
    import warnings
    import torchsde_amd
    B, d, T = 16, 10, 5
    sde = problems.PartialDependence(d, kind).to(DEV)
    y0 = torch.ones(B, d, device=DEV)
    ts = torch.linspace(0.0, 0.5, T, device=DEV)
    for names in (None, {"drift": "h"}):
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")          # adaptive Euler with non-additive noise warns, like the reference
            ys = torchsde_amd.sdeint(sde, y0, ts, method=method, dt=1e-2, adaptive=adaptive, options=dict(options),
                                     names=names)
        assert ys.shape == (T, B, d) and torch.isfinite(ys).all()


@pytest.mark.parametrize("adaptive", [False, True])
@pytest.mark.parametrize("method", ["milstein", "srk"])
@pytest.mark.parametrize("kind", KINDS)
def test_adjoint_basic(kind, method, adaptive):
    """Gradients land exactly on the trainable parameters; frozen / unused ones get none (or zeros)."""
    import torchsde_amd
    B, d = 128, 10
    sde = problems.PartialDependence(d, kind).to(DEV)
    y0 = torch.full((B, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, 0.5], device=DEV)
    trainable_before = sum(p.requires_grad for p in sde.parameters())
    sde.zero_grad()
    _, yt = torchsde_amd.sdeint_adjoint(sde, y0, ts, method=method, dt=1e-2, adaptive=adaptive)
    yt.sum(dim=1).mean(dim=0).backward()
    assert sum(p.requires_grad for p in sde.parameters()) == trainable_before
    for name, p in sde.named_parameters():
        if not p.requires_grad:
            assert p.grad is None, name
        elif p.grad is not None:
            assert torch.isfinite(p.grad).all(), name
    if kind in ("state", "params"):
        assert sde.scale.grad is not None and sde.scale.grad.abs().sum() > 0
    assert sde.spare_trainable.grad is None or sde.spare_trainable.grad.abs().sum() == 0

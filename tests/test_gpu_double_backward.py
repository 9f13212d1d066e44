"""Second derivatives through ``sdeint_adjoint`` (``create_graph=True``), run with ``-m gpu``.

The reference nests a second adjoint solve (adjoint.py:97-112); it gets through for Stratonovich SDEs only -- for an
Ito SDE its nested solve stops with "Adjoint `f_and_g` not defined" (adjoint_sde.py:308,318 -> :271; see
tests/golden/make_golden.py::gen_double_backward). Here the backward sweep is re-stated as differentiable torch
operations (torchsde_amd/adjoint_double.py). Pinned three ways:
  * the differentiable sweep returns the kernel sweep's gradients (to float64 rounding), for every adjoint method;
  * second derivatives against the REAL reference's (golden fixtures, Stratonovich). The two programs discretise the
    same continuous second derivative differently (adjoint of the adjoint vs. derivative of the discrete adjoint), so
    the tolerance is a discretisation one, stated below;
  * second derivatives against central finite differences of the (kernel) first-order gradients, including the Ito
    cases the reference cannot run.
"""
import numpy as np
import pytest
import torch

from tests import helpers
from workloads import problems

pytestmark = pytest.mark.gpu
DEV = "cuda"
F64 = torch.float64

KIND_CASES = [
    # problem, forward method, adjoint method, m
    ("gbm_ito", "euler", "euler", 4),
    ("gbm_ito", "milstein", None, 4),            # default adjoint method for Ito diagonal: milstein
    ("mlpdiag_ito", "srk", "milstein", 4),
    ("mlpdiag_ito", "euler", "euler", 4),
    ("gbm_strat", "midpoint", None, 4),
    ("gbm_strat", "heun", "heun", 4),
    ("mlpdiag_strat", "euler_heun", "euler_heun", 4),
    ("mlpdiag_strat", "milstein", "milstein", 4),
    ("general_ito", "euler", None, 3),
    ("general_strat", "midpoint", "heun", 3),
    ("scalar_ito", "euler", None, 1),
    ("additive_ito", "euler", None, 3),
]


def _setup(prob, method, m, B=6, d=4, steps=16, dt=2.0 ** -6, entropy=31):
    import torchsde_amd
    sde = problems.make(prob, dtype=F64, d=d, m=m).to(DEV)
    levy = "space-time" if method == "srk" else "none"
    bm = torchsde_amd.BrownianInterval(t0=0.0, t1=steps * dt, size=(B, m), dtype=F64, device=DEV, entropy=entropy,
                                       levy_area_approximation=levy, dt=dt)
    ts = torch.tensor([0.0, 5 * dt, steps * dt], dtype=F64, device=DEV)
    y0 = torch.full((B, d), 0.1, dtype=F64, device=DEV) + 0.01 * torch.arange(d, dtype=F64, device=DEV)
    wt = torch.tensor(np.random.default_rng(3).standard_normal((3, B, d)), dtype=F64, device=DEV)
    return sde, bm, ts, y0, wt, dt


def _first_order(sde, bm, ts, y0, wt, dt, method, adjoint_method, create_graph):
    import torchsde_amd
    y0 = y0.detach().requires_grad_(True)
    params = [p for p in sde.parameters() if p.requires_grad]
    ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=bm, method=method, adjoint_method=adjoint_method, dt=dt)
    loss = (ys ** 2 * wt).sum()
    grads = torch.autograd.grad(loss, [y0] + params, create_graph=create_graph, allow_unused=True)
    return y0, params, [torch.zeros_like(x) if g is None else g for g, x in zip(grads, [y0] + params)]


@pytest.mark.parametrize("prob,method,adjoint_method,m", KIND_CASES)
def test_differentiable_sweep_returns_the_kernel_sweeps_gradients(prob, method, adjoint_method, m):
    sde, bm, ts, y0, wt, dt = _setup(prob, method, m)
    _, _, kernel = _first_order(sde, bm, ts, y0, wt, dt, method, adjoint_method, create_graph=False)
    _, _, graph = _first_order(sde, bm, ts, y0, wt, dt, method, adjoint_method, create_graph=True)
    assert any(g.requires_grad for g in graph)
    for a, b in zip(kernel, graph):
        torch.testing.assert_close(b.detach(), a, rtol=1e-9, atol=1e-11)


def _second_order(sde, bm, ts, y0, wt, dt, method, adjoint_method, mix):
    y0, params, first = _first_order(sde, bm, ts, y0, wt, dt, method, adjoint_method, create_graph=True)
    phi = sum((g * w).sum() for g, w in zip(first, mix))
    second = torch.autograd.grad(phi, [y0] + params, allow_unused=True)
    return [torch.zeros_like(x) if h is None else h for h, x in zip(second, [y0] + params)]


@pytest.mark.parametrize("prob,method,adjoint_method,m", [c for c in KIND_CASES if c[0] != "mlpdiag_ito" or c[1] == "euler"])
def test_second_derivative_against_finite_differences(prob, method, adjoint_method, m):
    """d/d(y0, theta) of Phi = <mix, dL/d(y0, theta)> along a random direction. The finite difference is taken on the
    KERNEL first-order gradients. The derivative of the stored forward states w.r.t. (y0, theta) is itself taken by the
    stochastic adjoint (a discretisation of the continuous one, not the derivative of the forward solver), so the
    agreement is to discretisation error: 3% of the directional derivative at dt = 2^-9 (measured: <= 1% except the
    trigonometric scalar-noise problem, 6%, which gets 10%)."""
    steps, dt = 64, 2.0 ** -9
    sde, bm, ts, y0, wt, dt = _setup(prob, method, m, steps=steps, dt=dt)
    rng = np.random.default_rng(17)
    params = [p for p in sde.parameters() if p.requires_grad]
    mix = [torch.tensor(rng.standard_normal(tuple(x.shape)), dtype=F64, device=DEV) for x in [y0] + params]
    direction = [torch.tensor(rng.standard_normal(tuple(x.shape)), dtype=F64, device=DEV) for x in [y0] + params]
    second = _second_order(sde, bm, ts, y0, wt, dt, method, adjoint_method, mix)
    analytic = float(sum((h * u).sum() for h, u in zip(second, direction)))

    def phi_at(eps):
        with torch.no_grad():
            for p, u in zip(params, direction[1:]):
                p.add_(eps * u)
        try:
            _, _, first = _first_order(sde, bm, ts, y0 + eps * direction[0], wt, dt, method, adjoint_method, False)
            return float(sum((g * w).sum() for g, w in zip(first, mix)))
        finally:
            with torch.no_grad():
                for p, u in zip(params, direction[1:]):
                    p.sub_(eps * u)

    eps = 1e-5
    numeric = (phi_at(eps) - phi_at(-eps)) / (2 * eps)
    assert np.isfinite(analytic) and np.isfinite(numeric)
    tol = 1e-1 if prob == "scalar_ito" else 3e-2
    print(f"{prob}/{method}/{adjoint_method}: analytic {analytic:.6f} finite differences {numeric:.6f}")
    assert abs(analytic - numeric) <= tol * max(abs(numeric), abs(analytic)) + 1e-6, (analytic, numeric)


@pytest.mark.parametrize("prob,method,adjoint_method,m", KIND_CASES[:5])
def test_derivative_with_respect_to_the_cotangent_is_exact(prob, method, adjoint_method, m):
    """The gradients are linear in the loss weights; the derivative of <mix, gradients> w.r.t. those weights does not
    involve the forward solve at all, so here the graph of the sweep must agree with differences of the kernel sweep to
    rounding."""
    import torchsde_amd
    sde, bm, ts, y0, wt, dt = _setup(prob, method, m)
    params = [p for p in sde.parameters() if p.requires_grad]
    rng = np.random.default_rng(23)
    mix = [torch.tensor(rng.standard_normal(tuple(x.shape)), dtype=F64, device=DEV) for x in [y0] + params]
    direction = torch.tensor(rng.standard_normal(tuple(wt.shape)), dtype=F64, device=DEV)

    def phi(weights, create_graph):
        y = y0.detach().requires_grad_(True)
        ys = torchsde_amd.sdeint_adjoint(sde, y, ts, bm=bm, method=method, adjoint_method=adjoint_method, dt=dt)
        loss = (ys * weights).sum()
        grads = torch.autograd.grad(loss, [y] + params, create_graph=create_graph, allow_unused=True)
        return sum((g * w).sum() for g, w in zip(grads, mix) if g is not None)

    weights = wt.clone().requires_grad_(True)
    analytic = float((torch.autograd.grad(phi(weights, True), weights)[0] * direction).sum())
    numeric = float(phi(wt + direction, False) - phi(wt, False))     # linear: the difference IS the derivative
    assert abs(analytic - numeric) <= 1e-8 * max(1.0, abs(numeric)), (analytic, numeric)


def _golden_cases():
    import os
    return sorted(f[len("double_backward_"):-4] for f in os.listdir(helpers.GOLDEN) if f.startswith("double_backward_"))


@pytest.mark.parametrize("name", _golden_cases())
def test_second_derivative_against_reference_golden(name):
    """First-order gradients to rounding; second-order ones to the discretisation difference between the reference's
    nested adjoint solve and the derivative of the discrete sweep (dt = 2^-7): 1% of the largest entry per tensor
    (measured 2e-5 .. 1.5e-3)."""
    import torchsde_amd
    case = helpers.Case(name, "f64", prefix="double_backward_")
    z = case.z
    sde = case.sde(DEV)
    params = list(sde.parameters())
    y0 = case.y0(DEV).requires_grad_(True)
    bm = helpers.make_replay_bm(case.table(DEV), (case.B, case.m), F64, DEV, "none")
    ys = torchsde_amd.sdeint_adjoint(sde, y0, case.ts.to(DEV), bm=bm, method=case.method,
                                     adjoint_method=str(z["adjoint_method"]) or None, dt=case.dt)
    torch.testing.assert_close(ys.detach().cpu(), case.ys, rtol=1e-9, atol=1e-11)
    wt = torch.tensor(z["f64__loss_weights"], dtype=F64, device=DEV)
    loss = (ys ** 2 * wt).sum()
    first = torch.autograd.grad(loss, [y0] + params, create_graph=True, allow_unused=True)
    phi = 0.0
    for j, (g, x) in enumerate(zip(first, [y0] + params)):
        ref = torch.tensor(z[f"f64__first{j}"], dtype=F64)
        got = torch.zeros_like(ref) if g is None else g.detach().cpu()
        torch.testing.assert_close(got, ref, rtol=1e-8, atol=1e-9)
        if g is not None:
            phi = phi + (g * torch.tensor(z[f"f64__mix{j}"], dtype=F64, device=DEV)).sum()
    second = torch.autograd.grad(phi, [y0] + params, allow_unused=True)
    worst = 0.0
    for j, (h, x) in enumerate(zip(second, [y0] + params)):
        ref = torch.tensor(z[f"f64__second{j}"], dtype=F64)
        got = torch.zeros_like(ref) if h is None else h.cpu()
        scale = float(ref.abs().max())
        if scale == 0.0:
            assert float(got.abs().max()) < 1e-9
            continue
        worst = max(worst, float((got - ref).abs().max()) / scale)
    print(f"{name}: worst second-derivative deviation from the reference {worst:.3e} of the tensor's largest entry")
    assert worst < 1e-2


def test_second_derivative_through_the_perceptron_module():
    """The matrix-core route of sdeint_adjoint keeps its kernels for the first backward pass and hands a
    create_graph=True pass to the differentiable sweep."""
    import torchsde_amd
    d, B, steps, dt = 32, 64, 16, 2.0 ** -6
    sde = torchsde_amd.MLPDriftDiagonalSDE(d, d, activation="softplus", diffusion="sigmoid").to(DEV)
    bm = torchsde_amd.BrownianInterval(t0=0.0, t1=steps * dt, size=(B, d), dtype=torch.float32, device=DEV, entropy=8,
                                       dt=dt)
    ts = torch.tensor([0.0, steps * dt], device=DEV)
    params = [p for p in sde.parameters() if p.requires_grad]

    def first(create_graph):
        y0 = torch.full((B, d), 0.1, device=DEV, requires_grad=True)
        ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=bm, method="euler", adjoint_method="euler", dt=dt)
        assert type(ys.grad_fn).__name__.startswith("_MlpAdjointFn")
        grads = torch.autograd.grad((ys[-1] ** 2).sum(), [y0] + params, create_graph=create_graph)
        return y0, grads

    _, kernel = first(False)
    y0, graph = first(True)
    for a, b in zip(kernel, graph):
        scale = float(a.abs().max()) + 1e-12
        assert float((a - b.detach()).abs().max()) <= 2e-4 * scale + 1e-6
    penalty = sum((g ** 2).sum() for g in graph)
    second = torch.autograd.grad(penalty, [y0] + params, allow_unused=True)
    assert all(h is not None and bool(torch.isfinite(h).all()) for h in second)
    assert float(sum(h.abs().sum() for h in second)) > 0.0

"""The differentiable statement of the backward sweep (torchsde_amd/adjoint_double.py), CPU part.

One augmented step per adjoint method against the oracle's step (oracle/adjoint_ref.py, pinned to the reference by
tests/test_oracle_adjoint.py), and its autograd derivative against central differences of the ORACLE's step -- the
check that nothing in the differentiable statement is silently held constant (the Milstein mixed-partial term is the
one place where the first-order code detaches a weight)."""
import numpy as np
import pytest
import torch

from oracle import adjoint_ref
from torchsde_amd import adjoint, adjoint_double
from torchsde_amd.sde import ForwardSDE
from workloads import problems

F64 = torch.float64
CASES = [("gbm_ito", "euler", 4), ("gbm_ito", "milstein", 4), ("mlpdiag_ito", "milstein", 4),
         ("mlpdiag_ito", "euler", 4), ("mlpdiag_strat", "midpoint", 4), ("mlpdiag_strat", "milstein", 4),
         ("gbm_strat", "heun", 4), ("mlpdiag_strat", "euler_heun", 4), ("general_ito", "euler", 3),
         ("general_strat", "midpoint", 3), ("general_strat", "heun", 3), ("scalar_ito", "euler", 1),
         ("scalar_strat", "euler_heun", 1), ("additive_ito", "euler", 3)]


def _problem(prob, m, B=3, d=4, seed=0):
    sde = problems.make(prob, dtype=F64, d=d, m=m)
    params = [p for p in sde.parameters() if p.requires_grad]
    rng = np.random.default_rng(seed)
    y = torch.tensor(0.3 + 0.1 * rng.standard_normal((B, d)), dtype=F64)
    a = torch.tensor(rng.standard_normal((B, d)), dtype=F64)
    acc = [torch.tensor(0.1 * rng.standard_normal(tuple(p.shape)), dtype=F64) for p in params]
    v = torch.tensor(0.2 * rng.standard_normal((B, m)), dtype=F64)
    return sde, params, y, a, acc, v


def _oracle_step(sde, params, kind, state, v, t0, h):
    shapes = [t.size() for t in state]
    adj = adjoint_ref.AdjointSDERef(sde, params, shapes)
    flat = adjoint_ref._flatten([t.detach() for t in state]).unsqueeze(0)
    t0 = torch.tensor(t0, dtype=F64)
    out = adjoint_ref._aug_step(adj, kind, lambda ta, tb: v, t0, t0 + h, flat)
    return adjoint_ref._flat_to_shape(out.squeeze(0).detach(), shapes)


def _graph_step(sde, params, kind, state, v, t0, h):
    fwd = ForwardSDE(sde)
    adj = adjoint_double._GraphAdjoint(adjoint.AdjointSDE(fwd, params))
    t = [torch.tensor(x, dtype=F64) for x in (-t0, -(t0 + 0.5 * h), -(t0 + h))]    # forward times of the stages
    return adjoint_double._step(adj, kind, fwd.sde_type == "ito", state, t[0], t[1], t[2], h, v)


@pytest.mark.parametrize("prob,kind,m", CASES)
def test_step_equals_oracle_step(prob, kind, m):
    sde, params, y, a, acc, v = _problem(prob, m)
    t0, h = -0.75, 0.0625                                   # backward time: forward time 0.75 -> 0.6875
    want = _oracle_step(sde, params, kind, [y, a] + acc, v, t0, h)
    with torch.enable_grad():
        got = _graph_step(sde, params, kind, [y.clone().requires_grad_(True), a.clone().requires_grad_(True)] + acc, v,
                          t0, h)
    for g, w in zip(got, want):
        torch.testing.assert_close(g.detach().reshape(w.shape), w, rtol=1e-11, atol=1e-13)


@pytest.mark.parametrize("prob,kind,m", CASES)
def test_step_derivative_equals_differences_of_the_oracle_step(prob, kind, m):
    sde, params, y, a, acc, v = _problem(prob, m, seed=1)
    t0, h = -0.5, 0.03125
    rng = np.random.default_rng(2)
    mix = [torch.tensor(rng.standard_normal(tuple(x.shape)), dtype=F64) for x in [y, a] + acc]
    direction = [torch.tensor(rng.standard_normal(tuple(x.shape)), dtype=F64) for x in [y, a] + params]

    y_, a_ = y.clone().requires_grad_(True), a.clone().requires_grad_(True)
    with torch.enable_grad():
        out = _graph_step(sde, params, kind, [y_, a_] + acc, v, t0, h)
        phi = sum((o.reshape(w.shape) * w).sum() for o, w in zip(out, mix))
    grads = torch.autograd.grad(phi, [y_, a_] + params, allow_unused=True)
    analytic = float(sum((g * u).sum() for g, u in zip(grads, direction) if g is not None))

    def phi_at(eps):
        with torch.no_grad():
            for p, u in zip(params, direction[2:]):
                p.add_(eps * u)
        try:
            out = _oracle_step(sde, params, kind, [y + eps * direction[0], a + eps * direction[1]] + acc, v, t0, h)
            return float(sum((o * w).sum() for o, w in zip(out, mix)))
        finally:
            with torch.no_grad():
                for p, u in zip(params, direction[2:]):
                    p.sub_(eps * u)

    eps = 1e-6
    numeric = (phi_at(eps) - phi_at(-eps)) / (2 * eps)
    assert abs(analytic - numeric) <= 1e-6 * max(1.0, abs(numeric)), (analytic, numeric)


@pytest.mark.parametrize("prob,kind,m", [c for c in CASES if c[0] in ("mlpdiag_ito", "mlpdiag_strat", "general_strat")])
def test_three_chained_steps_carry_the_history(prob, kind, m):
    """From the second step on, (y, a) are functions of theta; the terms of the adjoint must stay PARTIAL derivatives at
    fixed (y, a) (values equal to the oracle's) while the second-order pass sees the whole history (derivative equal
    to differences of the oracle's three steps)."""
    sde, params, y, a, acc, _ = _problem(prob, m, seed=4)
    rng = np.random.default_rng(5)
    vs = [torch.tensor(0.2 * rng.standard_normal((y.shape[0], m)), dtype=F64) for _ in range(3)]
    t0, h = -0.5, 0.03125
    mix = [torch.tensor(rng.standard_normal(tuple(x.shape)), dtype=F64) for x in [y, a] + acc]
    direction = [torch.tensor(rng.standard_normal(tuple(x.shape)), dtype=F64) for x in [y, a] + params]

    def oracle(y_, a_):
        state = [y_, a_] + acc
        for k, v in enumerate(vs):
            state = _oracle_step(sde, params, kind, state, v, t0 + k * h, h)
        return state

    y_, a_ = y.clone().requires_grad_(True), a.clone().requires_grad_(True)
    with torch.enable_grad():
        state = [y_, a_] + acc
        for k, v in enumerate(vs):
            state = _graph_step(sde, params, kind, state, v, t0 + k * h, h)
        phi = sum((o.reshape(w.shape) * w).sum() for o, w in zip(state, mix))
    for g, w in zip(state, oracle(y, a)):
        torch.testing.assert_close(g.detach().reshape(w.shape), w, rtol=1e-10, atol=1e-12)
    grads = torch.autograd.grad(phi, [y_, a_] + params, allow_unused=True)
    analytic = float(sum((g * u).sum() for g, u in zip(grads, direction) if g is not None))

    def phi_at(eps):
        with torch.no_grad():
            for p, u in zip(params, direction[2:]):
                p.add_(eps * u)
        try:
            out = oracle(y + eps * direction[0], a + eps * direction[1])
            return float(sum((o * w).sum() for o, w in zip(out, mix)))
        finally:
            with torch.no_grad():
                for p, u in zip(params, direction[2:]):
                    p.sub_(eps * u)

    eps = 1e-6
    numeric = (phi_at(eps) - phi_at(-eps)) / (2 * eps)
    assert abs(analytic - numeric) <= 1e-6 * max(1.0, abs(numeric)), (analytic, numeric)

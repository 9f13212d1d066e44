"""GPU parity of ``sdeint_adjoint`` (run with ``-m gpu``): gradients vs the REAL reference under replayed
increments (golden fixtures), native vs materialised increments, and adjoint vs backprop-through-the-solver."""
import numpy as np
import pytest
import torch

from tests import helpers
from workloads import problems

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("name", helpers.adjoint_cases())
def test_adjoint_matches_reference_golden(name, tag):
    import torchsde_amd
    case = helpers.Case(name, tag, prefix="adjoint_")
    z = case.z
    adjoint_method = str(z["adjoint_method"]) or None
    sde = case.sde(DEV)
    y0 = case.y0(DEV).requires_grad_(True)
    bm = helpers.make_replay_bm(case.table(DEV), (case.B, case.m), case.dtype, DEV, case.levy)
    ys = torchsde_amd.sdeint_adjoint(sde, y0, case.ts.to(DEV), bm=bm, method=case.method,
                                     adjoint_method=adjoint_method, dt=case.dt)
    wt = torch.tensor(z[f"{tag}__loss_weights"], dtype=case.dtype, device=DEV)
    (ys * wt).sum().backward()
    rtol, atol = (3e-4, 3e-5) if tag == "f32" else (1e-9, 1e-11)
    torch.testing.assert_close(ys.detach().cpu(), case.ys, rtol=rtol, atol=atol)
    torch.testing.assert_close(y0.grad.cpu(), torch.tensor(z[f"{tag}__grad_y0"], dtype=case.dtype), rtol=rtol,
                               atol=atol)
    for j, p in enumerate(sde.parameters()):
        ref = torch.tensor(z[f"{tag}__grad_p{j}"], dtype=case.dtype)
        got = torch.zeros_like(ref) if p.grad is None else p.grad.cpu()
        # parameter gradients are sums over the batch and the steps: scale the absolute tolerance
        torch.testing.assert_close(got, ref, rtol=rtol, atol=atol * 10)


@pytest.mark.parametrize("prob,method,adjoint_method", [
    ("gbm_ito", "euler", "euler"), ("gbm_ito", "milstein", "milstein"), ("gbm_strat", "midpoint", "midpoint"),
    ("general_ito", "euler", "euler"), ("mlpdiag_ito", "srk", "milstein"),
])
def test_adjoint_native_equals_materialised(prob, method, adjoint_method):
    """Backward sweep on the counter-RNG path: generated cells == the same increments served as tensors."""
    import torchsde_amd
    B, d, m, steps, dt = 32, 4, 4, 16, 2.0 ** -5
    levy = "space-time" if method == "srk" else "none"
    kw = dict(t0=0.0, t1=steps * dt, size=(B, m), dtype=torch.float64, device=DEV, entropy=99,
              levy_area_approximation=levy, dt=dt)
    ts = torch.tensor([0.0, 6 * dt, steps * dt], dtype=torch.float64, device=DEV)

    def run(make_bm):
        sde = problems.make(prob, dtype=torch.float64, d=d, m=m).to(DEV)
        y0 = torch.full((B, d), 0.1, dtype=torch.float64, device=DEV, requires_grad=True)
        ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=make_bm(), method=method, adjoint_method=adjoint_method,
                                         dt=dt)
        (ys ** 2).sum().backward()
        return ys.detach(), y0.grad, [p.grad for p in sde.parameters()]

    def foreign():
        inner = torchsde_amd.BrownianInterval(**kw)

        class Foreign(torchsde_amd.BaseBrownian):
            def __call__(self, ta, tb=None, return_U=False, return_A=False):
                return inner(ta, tb, return_U=return_U)

            def __repr__(self):
                return "Foreign"
            dtype = property(lambda s: inner.dtype)
            device = property(lambda s: inner.device)
            shape = property(lambda s: inner.shape)
            levy_area_approximation = property(lambda s: inner.levy_area_approximation)
        return Foreign()

    a = run(lambda: torchsde_amd.BrownianInterval(**kw))
    b = run(foreign)
    assert torch.equal(a[0], b[0])
    assert torch.equal(a[1], b[1])
    for ga, gb in zip(a[2], b[2]):
        assert torch.equal(ga, gb)


@pytest.mark.parametrize("prob,method", [("gbm_strat", "midpoint"), ("gbm_ito", "euler"), ("mlpdiag_ito", "milstein")])
def test_adjoint_close_to_backprop_through_solver(prob, method):
    """reference tests/test_adjoint.py:100-154 (`test_against_sdeint`): same outputs, gradients agree loosely."""
    import torchsde_amd
    B, d, steps, dt = 64, 4, 256, 2.0 ** -8
    kw = dict(t0=0.0, t1=1.0, size=(B, d), dtype=torch.float64, device=DEV, entropy=5, dt=dt)
    ts = torch.tensor([0.0, 0.5, 1.0], dtype=torch.float64, device=DEV)

    def run(fn):
        sde = problems.make(prob, dtype=torch.float64, d=d).to(DEV)
        y0 = torch.full((B, d), 0.1, dtype=torch.float64, device=DEV, requires_grad=True)
        ys = fn(sde, y0, ts, bm=torchsde_amd.BrownianInterval(**kw), method=method, dt=dt)
        ys.sum().backward()
        return ys.detach(), y0.grad, [p.grad for p in sde.parameters()]

    ys_a, gy_a, gp_a = run(torchsde_amd.sdeint_adjoint)
    ys_b, gy_b, gp_b = run(torchsde_amd.sdeint)
    assert torch.equal(ys_a, ys_b)
    # The continuous adjoint is not the gradient of the discretisation: agreement is O(sqrt(dt)) for Ito
    # Euler (the reference's own tolerance is 1e-2 at dt=1e-3 with a 12-element state, tests/test_adjoint.py:151).
    torch.testing.assert_close(gy_a, gy_b, rtol=5e-2, atol=5e-2)
    for p, q in zip(gp_a, gp_b):
        scale = max(1.0, q.abs().max().item())
        assert ((p - q).abs().max() / scale).item() < 5e-2


@pytest.mark.parametrize("prob,shape", [("gbm_strat", (32, 8, 8)), ("general_strat", (32, 4, 4))])
def test_reversible_heun_is_reversible(prob, shape):
    """Solving forwards and then backwards in time with the carried (f, g, z) state reproduces the trajectory
    (reference tests/test_sdeint.py:219-252 `test_reversibility`, tolerance 1e-6)."""
    import torchsde_amd
    B, d, m = shape
    dtype = torch.float64
    steps, dt = 16, 2.0 ** -5
    ts = torch.linspace(0, steps * dt, 5, dtype=dtype, device=DEV)
    sde = problems.make(prob, dtype=dtype, d=d, m=m).to(DEV)
    y0 = torch.full((B, d), 0.1, dtype=dtype, device=DEV)
    bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(B, m), dtype=dtype, device=DEV, entropy=13, dt=dt)
    with torch.no_grad():
        ys, (f, g, z) = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="reversible_heun", dt=dt, extra=True)

        class NegatedTime(torch.nn.Module):       # the same SDE run backwards: t -> -t, drift and diffusion negated
            noise_type, sde_type = sde.noise_type, sde.sde_type

            def f(self, t, y):
                return -sde.f(-t, y)

            def g(self, t, y):
                return -sde.g(-t, y)

        rev_bm = torchsde_amd.ReverseBrownian(bm)
        back = torchsde_amd.sdeint(NegatedTime(), ys[-1], -ts.flip(0), bm=rev_bm, method="reversible_heun", dt=dt,
                                   extra_solver_state=(-f, -g, z))
    torch.testing.assert_close(back.flip(0), ys, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("prob,method,adjoint_method,shape", [
    ("gbm_ito", "euler", "euler", (128, 8, 8)),
    ("gbm_ito", "srk", None, (128, 8, 8)),
    ("mlpdiag_ito", "milstein", None, (96, 8, 8)),
    ("gbm_strat", "midpoint", None, (128, 8, 8)),
    ("general_ito", "euler", None, (64, 4, 4)),
    ("general_strat", "midpoint", None, (64, 4, 4)),
    ("additive_ito", "euler", None, (64, 4, 3)),
    ("scalar_ito", "euler", None, (64, 4, 1)),
])
def test_adjoint_matches_oracle_on_counter_rng_path(prob, method, adjoint_method, shape):
    """sdeint_adjoint on the generator's own path (fused forward, re-materialised reverse sweep) vs the oracle's
    restatement of the reference's adjoint (pinned to the real reference in tests/test_oracle_adjoint.py) driven
    by the C twin of the generator; float64, at shapes beyond the golden fixtures."""
    import torchsde_amd
    from oracle import adjoint_ref, counter
    B, d, m = shape
    dtype = torch.float64
    steps, dt = 16, 2.0 ** -5
    levy = "space-time" if method == "srk" else "none"
    ts_list = [0.0, 6 * dt, steps * dt]
    edges = np.arange(steps + 1) * dt
    sde = problems.make(prob, dtype=dtype, d=d, m=m)
    wt = torch.linspace(-1, 1, 3 * B * d, dtype=dtype).reshape(3, B, d)

    def bm_cpu(ta, tb, return_U=False):
        W, U, _ = counter.query(B * m, 404, edges, float(ta), float(tb), dtype=np.float64, have_h=(levy != "none"))
        W = torch.from_numpy(W).reshape(B, m)
        return (W, torch.from_numpy(U).reshape(B, m)) if return_U else W

    ys_ref, gy_ref, gp_ref = adjoint_ref.adjoint_gradients(sde, torch.full((B, d), 0.1, dtype=dtype),
                                                           torch.tensor(ts_list, dtype=dtype), bm_cpu, dt, method,
                                                           adjoint_method, wt)
    sde_g = problems.make(prob, dtype=dtype, d=d, m=m).to(DEV)
    y0 = torch.full((B, d), 0.1, dtype=dtype, device=DEV, requires_grad=True)
    bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(B, m), dtype=dtype, device=DEV, entropy=404, dt=dt,
                                       levy_area_approximation=levy)
    ys = torchsde_amd.sdeint_adjoint(sde_g, y0, torch.tensor(ts_list, dtype=dtype, device=DEV), bm=bm, method=method,
                                     adjoint_method=adjoint_method, dt=dt)
    (ys * wt.to(DEV)).sum().backward()
    torch.testing.assert_close(ys.detach().cpu(), ys_ref, rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(y0.grad.cpu(), gy_ref, rtol=1e-8, atol=1e-10)
    for p, ref in zip(sde_g.parameters(), gp_ref):
        torch.testing.assert_close(p.grad.cpu(), ref, rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("prob,method,adjoint_method", [
    ("gbm_ito", "milstein", None), ("gbm_ito", "euler", "euler"), ("gbm_strat", "midpoint", None),
    ("general_strat", "midpoint", None),
])
def test_adaptive_adjoint(prob, method, adjoint_method):
    """`adjoint_adaptive=True`: step doubling on the augmented state in the backward sweep. On a tight tolerance the
    gradients agree with the fixed-step backward sweep of a fine grid; the forward pass is unchanged."""
    import warnings
    import torchsde_amd
    B, d, m = 64, 4, 4
    dtype = torch.float64
    ts = torch.tensor([0.0, 0.3, 0.5], dtype=dtype, device=DEV)

    def run(**kw):
        sde = problems.make(prob, dtype=dtype, d=d, m=m).to(DEV)
        y0 = torch.full((B, d), 0.1, dtype=dtype, device=DEV, requires_grad=True)
        bm = torchsde_amd.BrownianInterval(0.0, 0.5, size=(B, m), dtype=dtype, device=DEV, entropy=77)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=bm, method=method, adjoint_method=adjoint_method,
                                             dt=2.0 ** -7, **kw)
        (ys ** 2).sum().backward()
        return ys.detach(), y0.grad, [p.grad for p in sde.parameters()]

    ys_f, gy_f, gp_f = run()
    ys_a, gy_a, gp_a = run(adjoint_adaptive=True, adjoint_rtol=1e-4, adjoint_atol=1e-5)
    assert torch.equal(ys_a, ys_f)
    scale = gy_f.abs().max().item()
    assert ((gy_a - gy_f).abs().max() / scale).item() < 3e-2
    for a, b in zip(gp_a, gp_f):
        assert torch.isfinite(a).all()
        assert ((a - b).abs().max() / max(1.0, b.abs().max().item())).item() < 5e-2


def _adaptive_adjoint_cases():
    import os
    return sorted(f[len("adjoint_adaptive_"):-4] for f in os.listdir(helpers.GOLDEN)
                  if f.startswith("adjoint_adaptive_"))


@pytest.mark.parametrize("name", _adaptive_adjoint_cases())
def test_adaptive_adjoint_matches_reference_golden(name):
    """`adjoint_adaptive=True` against the REAL reference under replayed increments: the replay table only holds the
    intervals the reference queried, so the accept / reject sequence (and the restart from `dt` on every output
    interval) has to be the reference's for the lookups to succeed at all; then the gradients to rounding. Includes
    the reversible-Heun pair."""
    import warnings
    import torchsde_amd
    case = helpers.Case(name, "f64", prefix="adjoint_adaptive_")
    z = case.z
    sde = case.sde(DEV)
    y0 = case.y0(DEV).requires_grad_(True)
    bm = helpers.make_replay_bm(case.table(DEV), (case.B, case.m), case.dtype, DEV, case.levy)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ys = torchsde_amd.sdeint_adjoint(sde, y0, case.ts.to(DEV), bm=bm, method=case.method,
                                         adjoint_method=str(z["adjoint_method"]) or None, dt=case.dt,
                                         adjoint_adaptive=True, adjoint_rtol=float(z["adjoint_rtol"]),
                                         adjoint_atol=float(z["adjoint_atol"]), dt_min=float(z["dt_min"]))
        wt = torch.tensor(z["f64__loss_weights"], dtype=case.dtype, device=DEV)
        (ys * wt).sum().backward()
    torch.testing.assert_close(ys.detach().cpu(), case.ys, rtol=1e-9, atol=1e-11)
    # the step sizes carry the last bits of a device-side reduction (see helpers.make_replay_bm): 1e-7, not rounding
    torch.testing.assert_close(y0.grad.cpu(), torch.tensor(z["f64__grad_y0"]), rtol=1e-7, atol=1e-9)
    for j, p in enumerate(sde.parameters()):
        ref = torch.tensor(z[f"f64__grad_p{j}"])
        got = torch.zeros_like(ref) if p.grad is None else p.grad.cpu()
        torch.testing.assert_close(got, ref, rtol=1e-7, atol=1e-8)

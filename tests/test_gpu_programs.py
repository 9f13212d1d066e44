"""Elementwise user code that no single-function form holds -- several functions of the state summed or multiplied, powers,
quotients, and SCALAR noise -- on `tsde_trajectory_prog_diag` (``-m gpu``): drift and diffusion travel as small postfix programs
(recognise.RecognisedProgram) and the whole solve is one launch, for every scheme with an in-register form.

Pinned against the ORACLE's restatement of the reference's loops on the same Brownian path (the reference's own ExScalar,
tests/problems.py:75-103) and against this package's stepwise route."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from tests import helpers
from workloads import problems

pytestmark = pytest.mark.gpu
DEV = "cuda"
B, D, STEPS, DT = 256, 8, 32, 2.0 ** -7


def _solve(sde, entropy, method, levy="none", stepwise=False, dtype=torch.float32, rows=B, d=D, m=None, row_offset=0):
    import torchsde_amd
    m = (1 if sde.noise_type == "scalar" else d) if m is None else m
    y0 = torch.full((rows, d), 0.3, device=DEV, dtype=dtype)
    ts = torch.tensor([0.0, 11.5 * DT, STEPS * DT], device=DEV, dtype=dtype)
    bm = torchsde_amd.BrownianInterval(0.0, STEPS * DT, size=(rows, m), device=DEV, dtype=dtype, entropy=entropy, dt=DT,
                                       levy_area_approximation=levy, row_offset=row_offset)
    options = {"hip_graph": False}
    if stepwise:
        options["trajectory_kernel"] = False
    with torch.no_grad():
        return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=DT, options=options)


def _book(sde):
    from torchsde_amd import solvers
    return getattr(sde, solvers.BaseSDESolver._RECOGNISED_ATTR, {"trusted": {}, "refused": {}})


def _launches(fn):
    from torchsde_amd import kernels as K
    K.prof_begin(8, 64)
    out = fn()
    torch.cuda.synchronize()
    return out, K.prof_end()[1]


SCHEMES = [("euler", "ito", "none"), ("milstein", "ito", "none"), ("srk", "ito", "space-time"),
           ("midpoint", "stratonovich", "none"), ("milstein", "stratonovich", "none")]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("method,sde_type,levy", SCHEMES)
def test_the_references_scalar_noise_problem_is_one_launch(method, sde_type, levy, dtype):
    """The SDE of the reference's ExScalar test problem (workloads.problems.ScalarTrig: f = -p^2 sin(y) cos(y)^3 -- zeros for Stratonovich --, g = p cos(y)^2 of
    shape (B, d, 1), one Brownian channel per row)."""
    sde = problems.ScalarTrig(D, sde_type, dtype=dtype).to(DEV)
    first = _solve(sde, 1, method, levy, dtype=dtype)
    assert torch.equal(first, _solve(sde, 1, method, levy, stepwise=True, dtype=dtype))
    assert list(_book(sde)["trusted"].values()) == [True], _book(sde)
    tol = dict(rtol=2e-5, atol=2e-6) if dtype == torch.float32 else dict(rtol=1e-11, atol=1e-12)
    for entropy in (2, 3):
        fast, launches = _launches(lambda: _solve(sde, entropy, method, levy, dtype=dtype))
        assert launches == 1
        torch.testing.assert_close(fast, _solve(sde, entropy, method, levy, stepwise=True, dtype=dtype), **tol)
        assert not torch.equal(fast[-1], fast[0])


class _Mixed(nn.Module):
    """Diagonal noise, drift and diffusion mixing several functions of the state."""
    noise_type = "diagonal"

    def __init__(self, sde_type, which):
        super().__init__()
        self.sde_type, self.which = sde_type, which
        gen = torch.Generator().manual_seed(7)
        self.mu = nn.Parameter(0.5 + torch.rand(D, generator=gen))
        self.sigma = nn.Parameter(0.1 + 0.3 * torch.rand(D, generator=gen))

    def f(self, t, y):
        if self.which == "sum":
            return torch.tanh(y) * self.mu - y
        if self.which == "rational":
            return -y / (1.0 + y ** 2) * self.mu
        if self.which == "compositions":          # single ATen operators that run as compositions of the machine's functions
            return F.silu(y) * self.mu - F.mish(y) - torch.clamp(y, min=0)
        return y - y ** 4 * self.mu - F.softplus(y) * 0.1

    def g(self, t, y):
        if self.which == "compositions":
            return self.sigma * torch.rsqrt(1.0 + y * y) + 0.05 * (2.0 + y * y) ** -2
        if self.which == "sum":
            return self.sigma * torch.sigmoid(y) + 0.05 * torch.cos(y)
        if self.which == "rational":
            return self.sigma / (2.0 + torch.cos(y))
        return self.sigma * torch.sqrt(1.0 + y * y)


@pytest.mark.parametrize("which", ["sum", "rational", "quartic", "compositions"])
@pytest.mark.parametrize("method,sde_type,levy", SCHEMES)
def test_mixed_elementwise_code_takes_the_program_kernel(which, method, sde_type, levy):
    sde = _Mixed(sde_type, which).to(DEV)
    _solve(sde, 1, method, levy)
    assert list(_book(sde)["trusted"].values()) == [True], _book(sde)
    assert [key[0][0][0] for key in _book(sde)["trusted"]] == ["program"]
    fast, launches = _launches(lambda: _solve(sde, 2, method, levy))
    assert launches == 1
    torch.testing.assert_close(fast, _solve(sde, 2, method, levy, stepwise=True), rtol=2e-5, atol=2e-6)
    # odd widths take the one-element-per-lane form; rows are global
    odd = _Mixed(sde_type, which).to(DEV)
    odd.mu, odd.sigma = nn.Parameter(odd.mu[:5].detach()), nn.Parameter(odd.sigma[:5].detach())
    whole = [_solve(odd, 4, method, levy, rows=64, d=5) for _ in range(2)][1]
    part = [_solve(odd, 4, method, levy, rows=40, d=5, row_offset=24) for _ in range(2)][1]
    assert torch.equal(part, whole[:, 24:])
    torch.testing.assert_close(whole, _solve(odd, 4, method, levy, rows=64, d=5, stepwise=True), rtol=2e-5, atol=2e-6)


def test_live_parameters_and_refusals():
    sde = _Mixed("ito", "sum").to(DEV)
    _solve(sde, 1, "euler")
    a = _solve(sde, 2, "euler")
    with torch.no_grad():
        sde.sigma.mul_(1.5)
    b, launches = _launches(lambda: _solve(sde, 2, "euler"))
    assert launches == 1 and not torch.equal(a, b)
    torch.testing.assert_close(b, _solve(sde, 2, "euler", stepwise=True), rtol=2e-5, atol=2e-6)

    class ReadsTimeOnTheHost(_Mixed):
        def f(self, t, y):
            return torch.tanh(y) * float(t) - y ** 3 * torch.sin(y)
    timed = ReadsTimeOnTheHost("ito", "sum").to(DEV)
    for entropy in (1, 2):
        got, launches = _launches(lambda: _solve(timed, entropy, "euler"))
        assert launches == 0
    assert any("expression program" in r for r in _book(timed)["refused"].values()), _book(timed)


class _TimeInTheArithmetic(nn.Module):
    """t takes part in arithmetic that is not affine in the state (the reference's ExAdditive drift, tests/problems.py:
    119-121, plus a nonlinear term; a diffusion that decays with t and depends on the state)."""
    noise_type = "diagonal"

    def __init__(self, sde_type):
        super().__init__()
        self.sde_type = sde_type
        gen = torch.Generator().manual_seed(9)
        self.a = nn.Parameter(0.2 + 0.3 * torch.rand(D, generator=gen))
        self.b = nn.Parameter(0.2 + 0.3 * torch.rand(D, generator=gen))

    def f(self, t, y):
        return self.b / torch.sqrt(1. + t) - y / (2. + 2. * t) + torch.tanh(y) * torch.cos(3.0 * t)

    def g(self, t, y):
        return (self.a * self.b / torch.sqrt(1. + t)).expand_as(y) * (1.0 + torch.sigmoid(y))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("method,sde_type,levy", SCHEMES)
def test_time_as_an_operand_of_the_programs(method, sde_type, levy, dtype):
    """Every scheme evaluates f and g at its own stage times (t_k; t_k + dt/2; t_k + dt/4, t_k + dt/2, t_k + dt): the programs
    read the time of the evaluation they are part of, and the solve agrees with the stepwise route, which hands the user's
    code those times one call at a time (base_solver.py:114-149, srk.py:66-72)."""
    sde = _TimeInTheArithmetic(sde_type).to(DEV).to(dtype)
    _solve(sde, 1, method, levy, dtype=dtype)
    assert list(_book(sde)["trusted"].values()) == [True], _book(sde)
    fast, launches = _launches(lambda: _solve(sde, 2, method, levy, dtype=dtype))
    assert launches == 1
    tol = dict(rtol=2e-5, atol=2e-6) if dtype == torch.float32 else dict(rtol=1e-11, atol=1e-12)
    torch.testing.assert_close(fast, _solve(sde, 2, method, levy, stepwise=True, dtype=dtype), **tol)


@pytest.mark.parametrize("method,levy", [("euler", "none"), ("milstein", "none"), ("srk", "space-time")])
def test_scalar_noise_program_rows_vs_oracle(method, levy):
    """16384 x 32, 500 steps of the reference's scalar-noise problem: sampled rows against the oracle's restatement of the
    reference's loop (euler.py:29-37, milstein.py:52-74, srk.py:57-88 on g of shape (B, d, 1)) on the same Brownian path, with
    the bound of tests/test_gpu_full_size_oracle.py."""
    import torchsde_amd
    from tests.test_gpu_full_size_oracle import _oracle_forward
    Bf, d, n, dt = 16384, 32, 500, 2.0 ** -9
    sde = problems.ScalarTrig(d, "ito").to(DEV)
    y0 = torch.full((Bf, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, n * dt], device=DEV)

    def bm(entropy):
        return torchsde_amd.BrownianInterval(0.0, n * dt, size=(Bf, 1), dtype=torch.float32, device=DEV, entropy=entropy, dt=dt,
                                             levy_area_approximation=levy)
    before = torch.get_num_threads()
    torch.set_num_threads(min(8, before))
    try:
        with torch.no_grad():
            torchsde_amd.sdeint(sde, y0, ts, bm=bm(5), method=method, dt=dt)
            ys, launches = _launches(lambda: torchsde_amd.sdeint(sde, y0, ts, bm=bm(20240601), method=method, dt=dt))
        assert launches == 1 and list(_book(sde)["trusted"].values()) == [True], _book(sde)
        rows = helpers.sampled_rows(Bf, 48, seed=8, seams=(2, 8, Bf - 2))
        ref32, ref64 = _oracle_forward(sde, rows, d, 1, 20240601, n, dt, method, 0.1, levy=levy != "none")
        helpers.assert_within_reference_rounding(ys[-1][torch.from_numpy(rows).to(DEV)], ref32[-1], ref64[-1],
                                                 f"scalar noise, {method}, program kernel")
    finally:
        torch.set_num_threads(before)


# ---- training THROUGH sdeint (autograd on): the programs on dual numbers --------------------------------------------------
def _train(sde, entropy, method, levy, stepwise, dtype, d=D):
    import torchsde_amd
    m = 1 if sde.noise_type == "scalar" else d
    y0 = torch.full((B, d), 0.3, device=DEV, dtype=dtype, requires_grad=True)
    ts = torch.tensor([0.0, 11.5 * DT, STEPS * DT], device=DEV, dtype=dtype)
    bm = torchsde_amd.BrownianInterval(0.0, STEPS * DT, size=(B, m), device=DEV, dtype=dtype, entropy=entropy, dt=DT,
                                       levy_area_approximation=levy)
    options = {"hip_graph": False}
    if stepwise:
        options["trajectory_kernel"] = False
    sde.zero_grad()
    ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=DT, options=options)
    weights = torch.cos(torch.arange(ys.numel(), device=DEV, dtype=dtype)).reshape(ys.shape)
    (ys * weights).sum().backward()
    return ys, y0.grad.clone(), {n: p.grad.clone() for n, p in sde.named_parameters()}


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("method,sde_type,levy", SCHEMES)
@pytest.mark.parametrize("which", ["scalar", "sum", "logistic"])
def test_gradients_through_sdeint_take_the_program_sensitivity_kernel(which, method, sde_type, levy, dtype):
    """`sdeint` with autograd recording, on modules whose code is not affine: values, dL/dy0 and the gradients of the module's
    own parameters (through whatever the user's code derives from them: `-p ** 2`) from tsde_trajectory_prog_diag_sens agree
    with back-propagation through the stepwise solver (the reference's behaviour, _core/sdeint.py:27-112)."""
    if which == "scalar":
        sde = problems.ScalarTrig(D, sde_type, dtype=dtype).to(DEV)
    elif which == "sum":
        sde = _Mixed(sde_type, "sum").to(DEV).to(dtype)
    else:
        sde = problems.Logistic(D, sde_type).to(DEV).to(dtype)
    first = _train(sde, 1, method, levy, False, dtype)                       # earns trust: the stepwise result, with its graph
    assert type(first[0].grad_fn).__name__ != "_ProgTrajectoryFnBackward"
    assert list(_book(sde)["trusted"].values()) == [True], _book(sde)
    fast = _train(sde, 2, method, levy, False, dtype)
    assert type(fast[0].grad_fn).__name__ == "_ProgTrajectoryFnBackward", type(fast[0].grad_fn).__name__
    slow = _train(sde, 2, method, levy, True, dtype)
    tol = dict(rtol=2e-3, atol=2e-4) if dtype == torch.float32 else dict(rtol=1e-8, atol=1e-10)
    torch.testing.assert_close(fast[0], slow[0], **(dict(rtol=2e-5, atol=2e-6) if dtype == torch.float32 else tol))
    torch.testing.assert_close(fast[1], slow[1], **tol)
    assert set(fast[2]) == set(slow[2]) and fast[2]
    for name in fast[2]:
        scale = slow[2][name].abs().max().item() + 1e-12
        err = (fast[2][name] - slow[2][name]).abs().max().item()
        assert err <= (2e-3 if dtype == torch.float32 else 1e-8) * scale, (name, err, scale)


def test_more_than_four_trainable_constants_stay_stepwise():
    class Many(nn.Module):
        noise_type, sde_type = "diagonal", "ito"

        def __init__(self):
            super().__init__()
            self.c = nn.ParameterList([nn.Parameter(0.1 * torch.ones(D) * (k + 1)) for k in range(5)])

        def f(self, t, y):
            return torch.tanh(y) * self.c[0] - y * self.c[1] + self.c[2]

        def g(self, t, y):
            return self.c[3] * torch.sigmoid(y) + self.c[4]
    sde = Many().to(DEV)
    for entropy in (1, 2):
        ys, _, grads = _train(sde, entropy, "euler", "none", False, torch.float32)
        assert type(ys.grad_fn).__name__ != "_ProgTrajectoryFnBackward" and len(grads) == 5


@pytest.mark.parametrize("method,levy", [("euler", "none"), ("milstein", "none"), ("srk", "space-time")])
@pytest.mark.parametrize("seed", range(0, 120, 5))
def test_random_expression_trees_on_the_program_kernel(seed, method, levy):
    """The seeded random trees of tests/test_recognise.py (drift depth <= 5: unary functions, powers, the four operations,
    constants, parameters, t) as user modules through the real kernel: the stack machine of csrc/trajectory.hip -- operand
    order of the reversed operators, constants beyond the eight register rows, the derivative program -- against the stepwise
    route, which runs the user's own torch code."""
    import random

    import torch.nn.functional as F

    from tests.test_recognise import _M, _random_tree
    rng = random.Random(seed)
    src_f, src_g = _random_tree(rng, 5 if seed % 3 == 0 else 4), _random_tree(rng, 4 if seed % 3 == 0 else 3)
    if "y" not in src_f:
        src_f = f"({src_f}) * y"
    if "y" not in src_g:
        src_g = f"({src_g}) + torch.sin(y)"
    env = {"torch": torch, "F": F}
    # (bounded dynamics: the random drift and diffusion are squashed, so that 32 steps stay finite whatever the tree)
    f = eval(f"lambda s, t, y: torch.tanh({src_f}) - y", env)
    g = eval(f"lambda s, t, y: 0.3 * torch.tanh({src_g})", env)
    sde = _M(f, g).to(DEV)
    sde.b = sde.b.to(DEV)
    d = sde.mu.numel()
    _solve(sde, 1, method, levy, d=d)
    book = _book(sde)
    if book["refused"]:
        assert any("more than" in r or "derivative" in r for r in book["refused"].values()), (src_f, src_g, book)
        return
    assert list(book["trusted"].values()) == [True], (src_f, src_g, book)
    fast, launches = _launches(lambda: _solve(sde, 2, method, levy, d=d))
    assert launches == 1
    torch.testing.assert_close(fast, _solve(sde, 2, method, levy, stepwise=True, d=d), rtol=5e-5, atol=5e-6,
                               msg=f"f: {src_f}\ng: {src_g}")

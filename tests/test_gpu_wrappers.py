"""Legacy Brownian wrappers and adjoint parameter handling on the GPU (reference tests/test_brownian_path.py,
test_brownian_tree.py, test_sdeint.py:160-179, test_adjoint.py:157-177)."""
import math

import pytest
import torch
from scipy.stats import kstest
from torch import nn

pytestmark = pytest.mark.gpu
DEV = "cuda"
F64 = torch.float64


@pytest.mark.parametrize("kind", ["path", "tree"])
def test_brownian_path_and_tree(kind):
    import torchsde_amd
    B = 65536
    w0 = torch.zeros(B, dtype=F64, device=DEV)
    if kind == "path":
        bm = torchsde_amd.BrownianPath(t0=0.0, w0=w0)
    else:
        bm = torchsde_amd.BrownianTree(t0=0.0, w0=w0, t1=1.0, entropy=5, tol=1e-6)
    with pytest.warns(UserWarning):
        w_a = bm(0.3)
    with pytest.warns(UserWarning):
        w_a2 = bm(0.3)
    assert w_a.shape == (B,) and torch.equal(w_a, w_a2)          # shape + repeat determinism
    inc = bm(0.3, 0.7)
    with pytest.warns(UserWarning):
        w_b = bm(0.7)
    torch.testing.assert_close(w_a + inc, w_b, rtol=1e-6, atol=1e-6)
    _, pval = kstest((inc / math.sqrt(0.4)).cpu().numpy(), "norm")
    assert pval > 1e-5
    if kind == "tree":    # pinned terminal value
        w1 = torch.ones(B, dtype=F64, device=DEV)
        pinned = torchsde_amd.BrownianTree(t0=0.0, w0=w0, t1=1.0, w1=w1, entropy=6, tol=1e-6)
        with pytest.warns(UserWarning):
            assert torch.allclose(pinned(1.0), w1)


def test_brownian_interval_like():
    import torchsde_amd
    y = torch.zeros(8, 3, dtype=F64, device=DEV)
    bm = torchsde_amd.brownian_interval_like(y, t0=0.0, t1=2.0, entropy=1)
    assert bm.shape == (8, 3) and bm.dtype == F64 and bm.device.type == "cuda"
    assert bm(0.5, 1.5).shape == (8, 3)


class _WithUnused(nn.Module):
    noise_type, sde_type = "diagonal", "stratonovich"

    def __init__(self):
        super().__init__()
        self.a = nn.Parameter(torch.tensor(-0.5, dtype=F64))
        self.frozen = nn.Parameter(torch.tensor(0.3, dtype=F64), requires_grad=False)
        self.unused = nn.Parameter(torch.tensor(1.0, dtype=F64))

    def f(self, t, y):
        return self.a * y

    def g(self, t, y):
        return self.frozen * torch.ones_like(y)


@pytest.mark.parametrize("adjoint", [False, True])
def test_params_with_without_grad_and_unused(adjoint):
    import torchsde_amd
    sde = _WithUnused().to(DEV)
    y0 = torch.full((16, 4), 0.1, dtype=F64, device=DEV, requires_grad=True)
    ts = torch.tensor([0.0, 0.5], dtype=F64, device=DEV)
    fn = torchsde_amd.sdeint_adjoint if adjoint else torchsde_amd.sdeint
    ys = fn(sde, y0, ts, method="midpoint", dt=2.0 ** -4)
    ys.sum().backward()
    assert y0.grad is not None and sde.a.grad is not None and torch.isfinite(sde.a.grad)
    assert sde.frozen.grad is None and sde.frozen.requires_grad is False
    assert sde.unused.grad is None or float(sde.unused.grad) == 0.0
    assert sde.a.requires_grad and y0.requires_grad           # flags untouched (test_adjoint.py:157-177)


def test_explicit_adjoint_params_and_no_module():
    import torchsde_amd
    theta = torch.tensor(-0.4, dtype=F64, device=DEV, requires_grad=True)

    class Plain:
        noise_type, sde_type = "diagonal", "ito"

        def f(self, t, y):
            return theta * y

        def g(self, t, y):
            return 0.2 * y

    y0 = torch.full((8, 4), 0.1, dtype=F64, device=DEV)
    ts = torch.tensor([0.0, 0.25], dtype=F64, device=DEV)
    ys = torchsde_amd.sdeint_adjoint(Plain(), y0, ts, method="euler", dt=2.0 ** -5, adjoint_params=(theta,))
    ys[-1].sum().backward()
    assert theta.grad is not None and torch.isfinite(theta.grad)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("n", [1, 7, 4096, 65536 * 64 + 3])
def test_error_norm_matches_reference_formula(n, dtype):
    """tsde_error_norm vs adaptive_stepping.py:42-76 evaluated with torch ops (sum order differs: tolerance)."""
    from torchsde_amd import kernels as K
    gen = torch.Generator(device=DEV).manual_seed(n)
    a = torch.randn(n, generator=gen, device=DEV, dtype=dtype)
    b = a + 1e-3 * torch.randn(n, generator=gen, device=DEV, dtype=dtype)
    rtol, atol, eps = 1e-3, 1e-4, 1e-7
    tol = (rtol * torch.max(a.abs(), b.abs()) + atol).clamp_min(eps)
    ref = torch.sqrt((((a - b) / tol) ** 2.).double().sum() / n).clamp_min(eps).item()
    got = K.error_norm(a, b, rtol, atol, eps)
    assert got.dtype == torch.float64 and got.dim() == 0
    assert abs(got.item() - ref) <= (1e-5 if dtype == torch.float32 else 1e-12) * ref
    assert K.error_norm(a, b, rtol, atol, eps).item() == got.item()          # fixed summation tree: reproducible
    assert K.error_norm(a, a, rtol, atol, eps).item() == pytest.approx(eps)  # clamped from below
    if n > 1:
        b[n // 2] = float("nan")
        assert math.isnan(K.error_norm(a, b, rtol, atol, eps).item())


@pytest.mark.parametrize("method,sde_type", [("euler", "ito"), ("midpoint", "stratonovich"), ("heun", "stratonovich")])
def test_batch_broadcast_diffusion_uses_one_gemm(method, sde_type):
    """Additive noise whose g is the same (d, m) matrix for every row, returned as an expanded view: the product
    g.dW is taken as one GEMM; the result matches the materialised-g path to GEMM rounding."""
    import torchsde_amd
    B, d, m = 4096, 16, 8
    gen = torch.Generator().manual_seed(0)
    sigma = (0.3 * torch.rand(d, m, generator=gen)).to(DEV)

    class Shared(nn.Module):
        noise_type = "additive"

        def __init__(self, expand):
            super().__init__()
            self.sde_type, self.expand = sde_type, expand

        def f(self, t, y):
            return -0.5 * y

        def g(self, t, y):
            if self.expand:
                return sigma.unsqueeze(0).expand(y.size(0), d, m)          # stride-0 batch dimension
            return sigma.unsqueeze(0).repeat(y.size(0), 1, 1)              # materialised copies

    y0 = torch.full((B, d), 0.2, device=DEV)
    ts = torch.tensor([0.0, 0.25, 0.5], device=DEV)
    outs = []
    for expand in (True, False):
        bm = torchsde_amd.BrownianInterval(0.0, 0.5, size=(B, m), device=DEV, dtype=torch.float32, entropy=9)
        with torch.no_grad():
            outs.append(torchsde_amd.sdeint(Shared(expand), y0, ts, bm=bm, method=method, dt=2.0 ** -5))
    torch.testing.assert_close(outs[0], outs[1], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("levy,method", [("none", "milstein"), ("space-time", "srk")])
def test_step_doubling_increment_is_the_merge_of_its_halves(levy, method):
    """Adaptive stepping queries the generator twice per attempt; the whole step's (W, U) is merged from the halves
    and equals a direct query of the whole step up to rounding."""
    import torchsde_amd
    from torchsde_amd import solvers
    from torchsde_amd.sde import ForwardSDE
    from workloads import problems
    B, d = 64, 8
    dtype = torch.float64
    bm = torchsde_amd.BrownianInterval(0.0, 1.0, size=(B, d), dtype=dtype, device=DEV, entropy=12,
                                       levy_area_approximation=levy)
    sde = ForwardSDE(problems.make("gbm_ito", dtype=dtype, d=d).to(DEV))
    solver = solvers.select(method, "ito")(sde=sde, bm=bm, dt=0.1, adaptive=True, rtol=1e-3, atol=1e-4, dt_min=1e-5,
                                           options={})
    solver._state_dtype = dtype
    ta, tm, tb = 0.1234, 0.1234 + 0.5 * 0.0377, 0.1234 + 0.0377
    whole, first, second = solver._step_doubling_noise(ta, tm, tb)
    W, U = bm(ta, tb, return_U=True) if levy != "none" else (bm(ta, tb), None)
    torch.testing.assert_close(whole.W, W, rtol=1e-12, atol=1e-14)
    assert torch.equal(first.W, bm(ta, tm)) and torch.equal(second.W, bm(tm, tb))
    if U is not None:
        torch.testing.assert_close(whole.U, U, rtol=1e-11, atol=1e-14)


@pytest.mark.parametrize("kind", ["path", "tree"])
def test_bridge_law_of_path_and_tree(kind):
    """The reference's `test_normality` for BrownianPath / BrownianTree (tests/test_brownian_path.py:73-100,
    test_brownian_tree.py:80-108): query the end first, then an interior time; given both ends the interior value is
    Brownian-bridge distributed. Also: t0 / t1 given as 0-d device tensors, a (batch, d) shape."""
    import numpy as np
    import torchsde_amd
    rng = np.random.default_rng(7)
    B = 32768
    for rep in range(3):
        w0_ = float(rng.standard_normal())
        w0 = torch.full((B,), w0_, dtype=F64, device=DEV)
        if kind == "path":
            bm = torchsde_amd.BrownianPath(t0=torch.tensor(0.0, device=DEV), w0=w0)
        else:
            bm = torchsde_amd.BrownianTree(t0=torch.tensor(0.0, device=DEV), w0=w0, t1=torch.tensor(1.0, device=DEV),
                                           entropy=40 + rep)
        with pytest.warns(UserWarning):
            w1 = bm(1.0).cpu().numpy()
        t = float(rng.uniform(0.01, 0.99))
        with pytest.warns(UserWarning):
            sample = bm(t).cpu().numpy()
        mean = ((1.0 - t) * w0_ + t * w1)
        std = math.sqrt((1.0 - t) * t)
        _, pval = kstest((sample - mean) / std, "norm")
        assert pval >= 1e-5, (kind, rep, t, pval)
    two_d = torchsde_amd.BrownianPath(t0=0.0, w0=torch.zeros(64, 5, device=DEV))
    with pytest.warns(UserWarning):
        assert two_d(0.4).shape == (64, 5)


@pytest.mark.parametrize("kind", ["path", "tree"])
def test_legacy_brownian_objects_reach_the_one_launch_route(kind):
    """`sdeint(..., bm=BrownianPath(...))` / `BrownianTree(...)` (derived.py:52-191): interval queries of these objects ARE
    their BrownianInterval's increments, so the solver talks to that interval; with a BrownianPath an unchanged user module
    takes the trajectory kernel exactly as with a BrownianInterval -- same path, same values as the stepwise route."""
    import torchsde_amd
    from tests.test_gpu_programs import _book, _launches
    from workloads import problems
    B, d, dt = 256, 8, 2.0 ** -6
    sde = problems.make("gbm_ito", d=d).to(DEV)
    y0 = torch.full((B, d), 0.3, device=DEV)
    ts = torch.tensor([0.0, 0.25, 0.5], device=DEV)
    w0 = torch.zeros(B, d, device=DEV)
    bm = torchsde_amd.BrownianPath(t0=0.0, w0=w0) if kind == "path" else \
        torchsde_amd.BrownianTree(t0=0.0, w0=w0, t1=1.0, entropy=5, tol=1e-6)

    def solve(options=None):
        with torch.no_grad():
            return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="euler", dt=dt, options=options)
    first = solve({"hip_graph": False})                                 # both routes, compared; the stepwise result
    fast, launches = _launches(lambda: solve({"hip_graph": False}))
    stepwise = solve({"hip_graph": False, "trajectory_kernel": False})
    assert torch.equal(first, stepwise)
    if kind == "path":
        assert launches == 1 and list(_book(sde)["trusted"].values()) == [True], _book(sde)
    else:
        # a BrownianTree's path is its dyadic tree resolved to `tol`: its cells are not the solver's steps, the kernels'
        # per-step generator cannot reproduce it, and the solve stays stepwise (on the same increments)
        assert launches == 0 and not _book(sde)["trusted"], _book(sde)
    torch.testing.assert_close(fast, stepwise, rtol=2e-5, atol=2e-6)
    # ... and the increments the solver consumed are the object's own
    direct = torch.stack([bm(k * dt, (k + 1) * dt) for k in range(4)])
    again = torch.stack([bm._interval(k * dt, (k + 1) * dt) for k in range(4)])
    assert torch.equal(direct, again)

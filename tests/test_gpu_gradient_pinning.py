"""Gradient parity of the autograd kernel routes is pinned AT RUN TIME (``-m gpu``).

The reference differentiates the user's code as it stands: ordinary autograd through the solver loop
(torchsde/_core/base_solver.py:143-149 under autograd; adjoint_sde.py:111-128 for ``sdeint_adjoint``). So

* a stop-gradient in the user's code -- ``y.detach() * mu``, ``sigma * y.data``, state arithmetic inside ``torch.no_grad()`` --
  is honoured: with autograd recording, the interpretation (recognise.check_stop_gradient) ends on it and solves 1, 2, 3 all
  give the gradients of autograd through the stepwise solve;
* the verifying solve of the sensitivity routes compares GRADIENTS too (solvers._both_routes_agree): a kernel wrapper that
  returns right values and wrong sensitivities never earns trust;
* ``TSDE_VERIFY_EVERY`` re-runs that comparison periodically and fails loudly; ``options={"assume_pure": True}`` is the one
  documented switch for modules with Python-side state.
"""
import pytest
import torch
from torch import nn

from workloads import problems

pytestmark = pytest.mark.gpu
DEV = "cuda"
B, D, STEPS, DT = 128, 8, 24, 2.0 ** -7


def _book(sde):
    from torchsde_amd import solvers
    return getattr(sde, solvers.BaseSDESolver._RECOGNISED_ATTR, {"trusted": {}, "refused": {}})


class _StopGradient(nn.Module):
    noise_type, sde_type = "diagonal", "ito"

    def __init__(self, kind):
        super().__init__()
        self.kind = kind
        self.mu = nn.Parameter(torch.linspace(-0.6, -0.1, D))
        self.sigma = nn.Parameter(torch.linspace(0.2, 0.5, D))

    def f(self, t, y):
        if self.kind == "detach":
            return y.detach() * self.mu
        if self.kind == "no_grad":
            with torch.no_grad():
                z = 2.0 * y
            return 0.5 * z * self.mu
        if self.kind == "mixed":
            return y.detach() * self.mu + 0.25 * y
        if self.kind == "tanh_detach":                     # (an expression program, not an affine form)
            return torch.tanh(y).detach() * self.mu + torch.sin(y) * 0.1
        return y * self.mu

    def g(self, t, y):
        if self.kind == "data":
            return self.sigma * y.data
        if self.kind == "tanh_detach":
            return self.sigma * torch.sigmoid(y)
        return self.sigma * y


def _train(sde, entropy, options, method="euler", levy="none", dtype=torch.float32):
    import torchsde_amd
    ts = torch.tensor([0.0, 7 * DT, STEPS * DT], device=DEV, dtype=dtype)
    gen = torch.Generator(device=DEV).manual_seed(7)
    weights = torch.randn(3, B, D, device=DEV, dtype=dtype, generator=gen)
    y0 = torch.full((B, D), 0.3, device=DEV, dtype=dtype, requires_grad=True)
    sde.zero_grad()
    bm = torchsde_amd.BrownianInterval(0.0, STEPS * DT, size=(B, D), device=DEV, dtype=dtype, entropy=entropy,
                                       levy_area_approximation=levy)
    ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=DT, options=dict(options, hip_graph=False))
    (ys * weights).sum().backward()
    grads = {n: (torch.zeros_like(p) if p.grad is None else p.grad.clone()) for n, p in sde.named_parameters()}
    return ys.detach(), y0.grad.clone(), grads, type(ys.grad_fn).__name__


def _same_gradients(a, b, rtol):
    for x, y, what in [(a[0], b[0], "ys"), (a[1], b[1], "dL/dy0")] + [(a[2][n], b[2][n], n) for n in a[2]]:
        scale = y.abs().max().item() + 1e-12
        err = (x - y).abs().max().item()
        assert err <= rtol * scale, (what, err, scale)


@pytest.mark.parametrize("kind", ["detach", "data", "no_grad", "mixed", "tanh_detach"])
def test_stop_gradients_in_user_code_are_honoured_on_solves_1_2_and_3(kind):
    """`detach` / `.data` / `no_grad` on the state used to be followed as the identity also with autograd
    recording -- first solve right (stepwise), every later solve silently different. Now every solve agrees with autograd through
    the stepwise solver, and the gradients really are the stop-gradient ones (they differ from the plain module's)."""
    sde = _StopGradient(kind).to(DEV)
    for entropy in (1, 2, 3):
        got = _train(sde, entropy, {})
        want = _train(sde, entropy, {"trajectory_kernel": False})
        _same_gradients(got, want, 2e-4)
        assert not got[3].startswith(("_TrajectoryFn", "_ProgTrajectoryFn")), got[3]       # never a sensitivity kernel
    assert not any(v is True for k, v in _book(sde)["trusted"].items() if k[-1] == "autograd"), _book(sde)
    if kind in ("detach", "data", "no_grad"):
        # ... and the stop-gradient matters: the module without it has another dL/dy0
        plain = _StopGradient("plain").to(DEV)
        other = _train(plain, 3, {"trajectory_kernel": False})
        assert (other[1] - want[1]).abs().max().item() > 1e-3 * other[1].abs().max().item()
    # without autograd a stop-gradient is the identity: the forward route still takes the kernel
    import torchsde_amd
    with torch.no_grad():
        ts = torch.tensor([0.0, STEPS * DT], device=DEV)
        y0 = torch.full((B, D), 0.3, device=DEV)
        for entropy in (1, 2):
            bm = torchsde_amd.BrownianInterval(0.0, STEPS * DT, size=(B, D), device=DEV, entropy=entropy)
            torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="euler", dt=DT)
    assert any(v is True for k, v in _book(sde)["trusted"].items() if k[-1] != "autograd"), _book(sde)


def test_stop_gradient_on_a_parameter_is_not_a_stop_gradient_on_the_state():
    """`self.mu.detach() * y`: the parameter gets no gradient, the state does -- on both routes."""
    class M(nn.Module):
        noise_type, sde_type = "diagonal", "ito"

        def __init__(self):
            super().__init__()
            self.mu = nn.Parameter(torch.linspace(-0.6, -0.1, D))
            self.sigma = nn.Parameter(torch.linspace(0.2, 0.5, D))

        def f(self, t, y):
            return self.mu.detach() * y

        def g(self, t, y):
            return self.sigma * y
    sde = M().to(DEV)
    _train(sde, 1, {})
    got = _train(sde, 2, {})
    want = _train(sde, 2, {"trajectory_kernel": False})
    assert got[3].startswith("_TrajectoryFn"), got[3]
    _same_gradients(got, want, 5e-4)
    assert got[2]["mu"].abs().max().item() == 0.0 and got[2]["sigma"].abs().max().item() > 0.0


@pytest.mark.parametrize("which", ["affine", "program"])
def test_a_wrong_sensitivity_is_refused_at_the_verifying_solve(which, monkeypatch):
    """Right values, wrong gradients: the monkey-patched kernel wrapper scales the sensitivities by 1.5. The verifying solve
    compares d<ys, r>/d(y0, parameters) with autograd through the stepwise graph, so the form never earns trust and every solve
    returns the stepwise result with the stepwise gradients."""
    from torchsde_amd import kernels as K
    sde = (problems.make("gbm_ito", d=D) if which == "affine" else problems.Logistic(D, "ito")).to(DEV)
    name = "_TrajectoryFn" if which == "affine" else "_ProgTrajectoryFn"
    fn = getattr(K, name)
    true_backward = fn.backward

    def wrong_backward(ctx, *grads):
        out = true_backward(ctx, *grads)
        return tuple(1.5 * g if torch.is_tensor(g) else g for g in out)
    monkeypatch.setattr(fn, "backward", staticmethod(wrong_backward))
    for entropy in (1, 2, 3):
        got = _train(sde, entropy, {})
        assert not got[3].startswith(name), got[3]
        want = _train(sde, entropy, {"trajectory_kernel": False})
        _same_gradients(got, want, 1e-6)
    verdicts = [v for k, v in _book(sde)["trusted"].items() if k[-1] == "autograd"]
    assert len(verdicts) == 1 and verdicts[0] is not True and "gradient" in verdicts[0], verdicts
    monkeypatch.undo()
    # the true kernel on a fresh object of the same class earns trust
    fresh = (problems.make("gbm_ito", d=D) if which == "affine" else problems.Logistic(D, "ito")).to(DEV)
    _train(fresh, 1, {})
    got = _train(fresh, 2, {})
    assert got[3].startswith(name), got[3]
    _same_gradients(got, _train(fresh, 2, {"trajectory_kernel": False}), 2e-3)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_the_verifying_solve_passes_gradients_of_honest_modules(dtype):
    """The gradient comparison does not refuse what it should accept: every scheme of the affine sensitivity kernel, f32 / f64."""
    for method, levy, sde_type in (("euler", "none", "ito"), ("milstein", "none", "ito"), ("srk", "space-time", "ito"),
                                   ("midpoint", "none", "stratonovich"), ("heun", "none", "stratonovich")):
        class M(nn.Module):
            noise_type = "diagonal"

            def __init__(self):
                super().__init__()
                self.sde_type = sde_type
                self.a = nn.Parameter(torch.linspace(-0.6, -0.1, D, dtype=dtype))
                self.b = nn.Parameter(torch.linspace(0.2, 0.5, D, dtype=dtype))

            def f(self, t, y):
                return self.a * y + 0.1

            def g(self, t, y):
                return self.b * y
        sde = M().to(DEV)
        _train(sde, 1, {}, method, levy, dtype)
        verdicts = [v for k, v in _book(sde)["trusted"].items() if k[-1] == "autograd"]
        assert verdicts == [True], (method, verdicts)
        got = _train(sde, 2, {}, method, levy, dtype)
        assert got[3].startswith("_TrajectoryFn"), (method, got[3])


def test_periodic_reverification_fails_loudly(monkeypatch):
    """TSDE_VERIFY_EVERY=N: every N-th solve of a trusted form runs both routes again; a kernel that has gone wrong since
    (here: its backward monkey-patched AFTER trust was earned) raises instead of training on."""
    from torchsde_amd import kernels as K
    from torchsde_amd import solvers
    sde = problems.make("gbm_ito", d=D).to(DEV)
    monkeypatch.setattr(solvers, "VERIFY_EVERY", 3)
    assert not _train(sde, 1, {})[3].startswith("_TrajectoryFn")        # verifying solve
    assert _train(sde, 2, {})[3].startswith("_TrajectoryFn")            # solve 1 of the trusted form
    assert _train(sde, 3, {})[3].startswith("_TrajectoryFn")            # solve 2
    third = _train(sde, 4, {})                                           # solve 3: verified again -> the stepwise result
    assert not third[3].startswith("_TrajectoryFn")
    _same_gradients(third, _train(sde, 4, {"trajectory_kernel": False}), 1e-6)
    true_backward = K._TrajectoryFn.backward
    monkeypatch.setattr(K._TrajectoryFn, "backward",
                        staticmethod(lambda ctx, *g: tuple(2.0 * x if torch.is_tensor(x) else x for x in true_backward(ctx, *g))))
    _train(sde, 5, {})
    _train(sde, 6, {})
    with pytest.raises(RuntimeError, match="re-verification"):
        _train(sde, 7, {})
    # ... on the forward route too (values)
    import torchsde_amd
    fwd = problems.make("gbm_ito", d=D).to(DEV)
    ts = torch.tensor([0.0, STEPS * DT], device=DEV)
    y0 = torch.full((B, D), 0.3, device=DEV)

    def solve(entropy):
        bm = torchsde_amd.BrownianInterval(0.0, STEPS * DT, size=(B, D), device=DEV, entropy=entropy)
        with torch.no_grad():
            return torchsde_amd.sdeint(fwd, y0, ts, bm=bm, method="euler", dt=DT, options={"hip_graph": False})
    for entropy in range(1, 4):
        solve(entropy)
    true_launch = K.trajectory_affine_diag

    def wrong_launch(ys, *args, **kwargs):
        true_launch(ys, *args, **kwargs)
        ys.mul_(1.01)
    monkeypatch.setattr(K, "trajectory_affine_diag", wrong_launch)
    with pytest.raises(RuntimeError, match="re-verification"):
        for entropy in range(4, 8):
            solve(entropy)


class _Logs(nn.Module):
    """Python-side state that changes at every call and is NOT a pure call counter (a list grows)."""
    noise_type, sde_type = "diagonal", "ito"

    def __init__(self):
        super().__init__()
        self.mu = nn.Parameter(torch.linspace(-0.6, -0.1, D))
        self.sigma = nn.Parameter(torch.linspace(0.2, 0.5, D))
        self.log = []

    def f(self, t, y):
        self.log.append("f")
        return self.mu * y

    def g(self, t, y):
        return self.sigma * y


@pytest.mark.parametrize("how", ["option", "attribute"])
def test_assume_pure_is_the_one_switch_for_python_side_state(how):
    import torchsde_amd
    from torchsde_amd import kernels as K
    ts = torch.tensor([0.0, STEPS * DT], device=DEV)
    y0 = torch.full((B, D), 0.3, device=DEV)

    def solve(sde, entropy, options):
        bm = torchsde_amd.BrownianInterval(0.0, STEPS * DT, size=(B, D), device=DEV, entropy=entropy)
        with torch.no_grad():
            K.prof_begin(8, 64)
            ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="euler", dt=DT, options=dict(options, hip_graph=False))
            torch.cuda.synchronize()
            return ys, K.prof_end()[1]
    plain = _Logs().to(DEV)
    for entropy in (1, 2, 3):
        assert solve(plain, entropy, {})[1] == 0                      # stepwise: the module changes its own state
    sde = _Logs().to(DEV)
    options = {"assume_pure": True} if how == "option" else {}
    if how == "attribute":
        sde.tsde_assume_pure = True
    first, _ = solve(sde, 1, options)
    assert torch.equal(first, solve(plain, 1, {})[0])
    for entropy in (2, 3):
        ys, launches = solve(sde, entropy, options)
        assert launches == 1
        assert torch.equal(ys, solve(plain, entropy, {})[0])


class _LatentStopGradient(nn.Module):
    """The latent-SDE shape of tests/test_gpu_recognise.py::_Latent with a stop-gradient in front of the drift network."""
    noise_type, sde_type = "diagonal", "ito"

    def __init__(self, d, hidden, stop):
        super().__init__()
        self.stop = stop
        self.net = nn.Sequential(nn.Linear(d, hidden), nn.Softplus(), nn.Linear(hidden, d))
        self.w = nn.Parameter(torch.full((d,), 0.5))
        self.b = nn.Parameter(torch.zeros(d))

    def f(self, t, y):
        return self.net(y.detach() if self.stop else y)

    def g(self, t, y):
        return 0.1 * torch.sigmoid(self.w * y + self.b)


@pytest.mark.parametrize("stop", [False, True])
def test_sdeint_adjoint_honours_a_stop_gradient_in_front_of_the_drift_network(stop):
    """adjoint_sde.py:111-128 asks autograd for a^T df/dy of the user's code: with `net(y.detach())` that is zero. The matrix-core
    adjoint differentiates the recognised network, so such a module must stay on the stepwise adjoint (and the honest one must
    still take the kernels)."""
    import torchsde_amd
    d, hidden, Bn, steps, dt = 32, 32, 96, 16, 2.0 ** -6
    sde = _LatentStopGradient(d, hidden, stop).to(DEV)
    ts = torch.tensor([0.0, steps * dt], device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(3)
    weights = torch.randn(2, Bn, d, device=DEV, generator=gen)
    results = []
    for stepwise in (False, False, True):
        y0 = torch.full((Bn, d), 0.1, device=DEV, requires_grad=True)
        sde.zero_grad()
        bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(Bn, d), device=DEV, entropy=9)
        opts = {"trajectory_kernel": False} if stepwise else None
        ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=bm, method="euler", adjoint_method="euler", dt=dt, options=opts,
                                         adjoint_options=opts)
        took_kernel = type(ys.grad_fn).__name__.startswith("_MlpAdjointFn")
        assert took_kernel == (not stop and not stepwise), (stop, stepwise, type(ys.grad_fn).__name__)
        (ys * weights).sum().backward()
        results.append((ys.detach(), y0.grad.clone(), {n: p.grad.clone() for n, p in sde.named_parameters()}, ""))
    _same_gradients(results[1], results[2], 2e-3 if not stop else 1e-6)

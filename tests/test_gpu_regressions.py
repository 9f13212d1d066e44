"""Regression tests (``-m gpu``) for defects found in review: silent wrong answers that no parity test covered."""
import pytest
import torch

from workloads import problems

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _bm(B, d, t1, entropy=3, dtype=torch.float32):
    import torchsde_amd
    return torchsde_amd.BrownianInterval(0.0, t1, size=(B, d), dtype=dtype, device=DEV, entropy=entropy)


def test_subclass_overriding_the_drift_is_integrated_with_ITS_drift():
    """A subclass of a closed-form module that overrides `f` (time-dependent drift) without restating `closed_form`
    must take the stepwise path: the parent's coefficients are not its dynamics."""
    import torchsde_amd

    class TimeDependent(torchsde_amd.AffineDiagonalSDE):
        def f(self, t, y):
            return torch.cos(3.0 * t) * y

    B, d, dt = 64, 8, 2.0 ** -5
    sub = TimeDependent(-0.5, 0.0, 0.3, 0.0, dtype=torch.float32).to(DEV)

    class Plain(torch.nn.Module):       # the same dynamics as a module that never had a closed form
        noise_type, sde_type = "diagonal", "ito"

        def f(self, t, y):
            return torch.cos(3.0 * t) * y

        def g(self, t, y):
            return 0.3 * y

    y0 = torch.full((B, d), 0.5, device=DEV)
    ts = torch.tensor([0.0, 0.5, 1.0], device=DEV)
    with torch.no_grad():
        got = torchsde_amd.sdeint(sub, y0, ts, bm=_bm(B, d, 1.0), method="euler", dt=dt)
        want = torchsde_amd.sdeint(Plain(), y0, ts, bm=_bm(B, d, 1.0), method="euler", dt=dt)
        parent = torchsde_amd.sdeint(torchsde_amd.AffineDiagonalSDE(-0.5, 0.0, 0.3, 0.0, dtype=torch.float32).to(DEV), y0,
                                     ts, bm=_bm(B, d, 1.0), method="euler", dt=dt)
    torch.testing.assert_close(got, want, rtol=1e-6, atol=1e-7)
    assert (got - parent).abs().max().item() > 1e-2          # and NOT the parent's solution
    # the differentiable route has the same guard
    y0g = y0.clone().requires_grad_(True)
    ys = torchsde_amd.sdeint(sub, y0g, ts, bm=_bm(B, d, 1.0), method="euler", dt=dt)
    torch.testing.assert_close(ys.detach(), want, rtol=1e-6, atol=1e-7)
    ys[-1].sum().backward()
    assert torch.isfinite(y0g.grad).all()


def test_hip_graph_cache_distinguishes_names_and_rebound_parameters():
    """One SDE object, `hip_graph=True`: a second call with another `names=` mapping, or after a parameter was
    re-bound to new storage, must not replay the first call's graph."""
    import torchsde_amd

    class TwoDrifts(torch.nn.Module):
        noise_type, sde_type = "diagonal", "ito"

        def __init__(self):
            super().__init__()
            self.rate = torch.nn.Parameter(torch.tensor(-0.5))

        def f(self, t, y):
            return self.rate * y

        def other(self, t, y):
            return torch.sin(y)

        def g(self, t, y):
            return 0.2 + 0.0 * y

    sde = TwoDrifts().to(DEV)
    B, d, dt = 32, 4, 2.0 ** -4
    y0 = torch.full((B, d), 0.3, device=DEV)
    ts = torch.tensor([0.0, 1.0], device=DEV)

    def solve(graph, **kw):
        with torch.no_grad():
            return torchsde_amd.sdeint(sde, y0, ts, bm=_bm(B, d, 1.0), method="euler", dt=dt,
                                       options={"hip_graph": graph}, **kw)

    a = solve(True)
    b = solve(True, names={"drift": "other"})
    assert torch.equal(a, solve(False)) and torch.equal(b, solve(False, names={"drift": "other"}))
    assert not torch.equal(a, b)
    assert torch.equal(solve(True), a)                       # and the first graph is still valid for the first call
    with torch.no_grad():
        sde.rate.data = torch.tensor(-2.0, device=DEV)       # new storage behind the same Parameter object
    c = solve(True)
    assert torch.equal(c, solve(False)) and not torch.equal(c, a)


@pytest.mark.parametrize("adjoint_method", ["reversible_heun", "log_ode", "srk"])
def test_unusable_adjoint_method_raises_when_sdeint_adjoint_is_called(adjoint_method):
    import torchsde_amd
    ito = adjoint_method == "srk"
    sde = problems.make("gbm_ito" if ito else "gbm_strat", d=4).to(DEV)
    y0 = torch.full((8, 4), 0.1, device=DEV, requires_grad=True)
    ts = torch.tensor([0.0, 0.5], device=DEV)
    levy = "foster" if adjoint_method == "log_ode" else "space-time" if ito else "none"
    bm = torchsde_amd.BrownianInterval(0.0, 0.5, size=(8, 4), device=DEV, dtype=torch.float32, entropy=1,
                                       levy_area_approximation=levy)
    with pytest.raises((ValueError, RuntimeError)):
        torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=bm, method="euler" if ito else "midpoint",
                                    adjoint_method=adjoint_method, dt=0.1)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs in one process")
def test_tensors_on_a_device_that_is_not_current():
    """Tensors on cuda:1 while cuda:0 is current: launches must go to cuda:1's stream with cuda:1 current."""
    import torchsde_amd
    assert torch.cuda.current_device() == 0
    dev = torch.device("cuda", 1)
    sde = problems.make("gbm_ito", d=8).to(dev)
    y0 = torch.full((128, 8), 0.1, device=dev)
    ts = torch.tensor([0.0, 1.0], device=dev)

    def solve(device):
        bm = torchsde_amd.BrownianInterval(0.0, 1.0, size=(128, 8), dtype=torch.float32, device=device, entropy=5)
        with torch.no_grad():
            return torchsde_amd.sdeint(sde.to(device), y0.to(device), ts.to(device), bm=bm, method="euler", dt=2.0 ** -5)

    on_1 = solve(dev)
    assert torch.cuda.current_device() == 0 and on_1.device == dev
    assert torch.equal(on_1.cpu(), solve(torch.device("cuda", 0)).cpu())


def test_module_is_copyable_after_a_graph_solve():
    """`hip_graph=True` keeps its captured graphs on the SDE object; `copy.deepcopy` / pickling of that module (EMA
    copies, checkpoints) must keep working and the copy must capture graphs of its own."""
    import copy
    import pickle
    import torchsde_amd
    sde = problems.make("gbm_ito", d=8).to(DEV)
    B, d, dt = 64, 8, 2.0 ** -5
    y0 = torch.full((B, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, 1.0], device=DEV)

    def solve(module):
        with torch.no_grad():
            return torchsde_amd.sdeint(module, y0, ts, bm=_bm(B, d, 1.0), method="euler", dt=dt,
                                       options={"hip_graph": True})
    a = solve(sde)
    twin = copy.deepcopy(sde)
    restored = pickle.loads(pickle.dumps(sde))
    assert torch.equal(solve(twin), a) and torch.equal(solve(restored.to(DEV)), a)
    with torch.no_grad():
        twin.mu.mul_(2.0)                    # the copy has its own parameters and its own graph
    assert not torch.equal(solve(twin), a) and torch.equal(solve(sde), a)


def test_graph_with_drift_and_diffusion_as_parallel_branches_is_bit_identical():
    """Inside a captured solve f and g are recorded as parallel branches of the HIP graph (sde.py: _f_beside_g); the
    result is the eager solve's, bit for bit, also over many replays with new seeds, and `overlap_f_g=False` restores
    the sequential recording."""
    import torchsde_amd
    B, d, n, dt = 4096, 64, 64, 2.0 ** -8
    sde = problems.make("gbm_ito", d=d).to(DEV)
    y0 = torch.full((B, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, n * dt], device=DEV)

    def solve(entropy, **options):
        bm = torchsde_amd.BrownianInterval(0.0, n * dt, size=(B, d), dtype=torch.float32, device=DEV, entropy=entropy,
                                           dt=dt)
        with torch.no_grad():
            return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="midpoint" if sde.sde_type != "ito" else "euler", dt=dt,
                                       options=options or None)
    for entropy in (1, 2, 3, 4):
        eager = solve(entropy)
        assert torch.equal(solve(entropy, hip_graph=True), eager)
        assert torch.equal(solve(entropy, hip_graph=True, overlap_f_g=False), eager)

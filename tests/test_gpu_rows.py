"""Small row-coupled systems -- channels that read each other -- take ONE launch (torchsde_amd/recognise_rows.py +
specialise.source_rows; ``-m gpu``): the stochastic Lorenz system of the reference's examples/latent_sde_lorenz.py:56-86,
written with `torch.split` / `torch.cat` the way that example writes it (workloads/problems.py), and modules written with `y[:, c]` / `torch.stack`, parameters and t in the arithmetic. The kernel is the
program kernel with one lane per ROW and a model generated from the user's code; pinned against the stepwise route, which runs
the user's own torch code (and replays the reference's goldens)."""
import pytest
import torch
from torch import nn

from workloads.problems import StochasticLorenz

pytestmark = pytest.mark.gpu
DEV = "cuda"
B, STEPS, DT = 512, 40, 2.0 ** -8


@pytest.fixture(autouse=True)
def _compile_in_the_calling_thread(monkeypatch, tmp_path_factory):
    from torchsde_amd import specialise
    monkeypatch.setattr(specialise, "MODE", "sync")
    monkeypatch.setenv("TSDE_SPECIALISE_CACHE", str(tmp_path_factory.getbasetemp() / "specialised"))
    yield


class ForcedVanDerPol(nn.Module):
    """x' = v, v' = mu (1 - x^2) v - x + A sin(t); noise proportional to a bounded function of the other channel."""
    noise_type = "diagonal"

    def __init__(self, sde_type):
        super().__init__()
        self.sde_type = sde_type
        self.mu = nn.Parameter(torch.tensor(1.5))
        self.amp = nn.Parameter(torch.tensor([0.4]))
        self.sigma = nn.Parameter(torch.tensor([0.2, 0.3]))

    def f(self, t, y):
        x, v = y[:, 0], y[:, 1]
        return torch.stack([v, self.mu * (1 - x ** 2) * v - x + self.amp * torch.sin(t)], dim=1)

    def g(self, t, y):
        x, v = y.unbind(dim=1)
        return torch.stack([torch.tanh(v), torch.sigmoid(x)], dim=1) * self.sigma


def _solve(sde, d, entropy, method, levy="none", stepwise=False, dtype=torch.float32, y0=None):
    import torchsde_amd
    gen = torch.Generator(device=DEV).manual_seed(3)
    y0 = torch.randn(B, d, device=DEV, dtype=dtype, generator=gen) if y0 is None else y0
    ts = torch.tensor([0.0, 13.5 * DT, STEPS * DT], device=DEV, dtype=dtype)
    bm = torchsde_amd.BrownianInterval(0.0, STEPS * DT, size=(B, d), device=DEV, dtype=dtype, entropy=entropy,
                                       levy_area_approximation=levy)
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


@pytest.mark.parametrize("method,levy", [("euler", "none"), ("srk", "space-time")])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_the_reference_examples_stochastic_lorenz_is_one_launch(method, levy, dtype):
    sde = StochasticLorenz()
    first = _solve(sde, 3, 1, method, levy, dtype=dtype)
    assert torch.equal(first, _solve(sde, 3, 1, method, levy, stepwise=True, dtype=dtype))     # the verifying solve
    assert list(_book(sde)["trusted"].values()) == [True], _book(sde)
    for entropy in (2, 3):
        fast, launches = _launches(lambda: _solve(sde, 3, entropy, method, levy, dtype=dtype))
        assert launches == 1
        slow = _solve(sde, 3, entropy, method, levy, stepwise=True, dtype=dtype)
        # + - * only, in the user's order, one rounding per operation: the stepwise bits
        assert torch.equal(fast, slow), (fast - slow).abs().max()


@pytest.mark.parametrize("method,levy,sde_type", [("euler", "none", "ito"), ("srk", "space-time", "ito"),
                                                  ("midpoint", "none", "stratonovich"), ("heun", "none", "stratonovich"),
                                                  ("euler_heun", "none", "stratonovich")])
def test_indexing_stack_parameters_and_time(method, levy, sde_type):
    sde = ForcedVanDerPol(sde_type).to(DEV)
    _solve(sde, 2, 1, method, levy)
    assert list(_book(sde)["trusted"].values()) == [True], _book(sde)
    fast, launches = _launches(lambda: _solve(sde, 2, 2, method, levy))
    assert launches == 1
    torch.testing.assert_close(fast, _solve(sde, 2, 2, method, levy, stepwise=True), rtol=2e-5, atol=2e-6)
    with torch.no_grad():                       # an optimiser step: the constants are this solve's live values
        sde.mu.mul_(0.5)
        sde.sigma.add_(0.1)
    changed, launches = _launches(lambda: _solve(sde, 2, 2, method, levy))
    assert launches == 1 and not torch.equal(changed, fast)
    torch.testing.assert_close(changed, _solve(sde, 2, 2, method, levy, stepwise=True), rtol=2e-5, atol=2e-6)


def test_what_stays_stepwise():
    from torchsde_amd import specialise

    class RowMean(StochasticLorenz):            # a reduction over the channels is not column arithmetic
        def f(self, t, y):
            return super().f(t, y) - y.mean(dim=1, keepdim=True)

    class Wide(nn.Module):                      # more than 8 channels
        noise_type, sde_type = "diagonal", "ito"

        def f(self, t, y):
            return torch.cat([y[:, 1:], y[:, :1]], dim=1) - y

        def g(self, t, y):
            return 0.1 * y
    for sde, d in ((RowMean(), 3), (Wide(), 12)):
        for entropy in (1, 2):
            out, launches = _launches(lambda: _solve(sde, d, entropy, "euler"))
            assert launches == 0
        assert _book(sde)["refused"]
    # Milstein needs the diffusion's derivative: stepwise
    sde = StochasticLorenz()
    for entropy in (1, 2):
        out, launches = _launches(lambda: _solve(sde, 3, entropy, "milstein"))
        assert launches == 0
    # no compiler: there is no interpreter for these systems, so nothing changes
    sde = StochasticLorenz()
    import unittest.mock as mock
    with mock.patch.object(specialise, "compiler", lambda: None):
        for entropy in (1, 2):
            out, launches = _launches(lambda: _solve(sde, 3, entropy, "euler"))
            assert launches == 0
    assert torch.equal(out, _solve(sde, 3, 2, "euler", stepwise=True))


def test_background_compilation_keeps_the_solve_stepwise_until_the_unit_is_there(monkeypatch):
    import time

    from torchsde_amd import specialise
    monkeypatch.setattr(specialise, "MODE", "1")
    sde = StochasticLorenz(drift_constants=(9., 27., 2.5))     # (other constants: another unit than the tests above)
    first, launches = _launches(lambda: _solve(sde, 3, 1, "euler"))
    assert launches == 0
    deadline = time.time() + 120
    while time.time() < deadline and any(v == "pending" for v in specialise.status().values()):
        time.sleep(0.5)
    _solve(sde, 3, 2, "euler")                                # the verifying solve
    out, launches = _launches(lambda: _solve(sde, 3, 3, "euler"))
    assert launches == 1 and torch.equal(out, _solve(sde, 3, 3, "euler", stepwise=True))

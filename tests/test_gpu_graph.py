"""HIP-graph replay of a whole solve (options={'hip_graph': True}) equals the eager path bit for bit, also
after changing the Brownian seed, the initial state and the SDE parameters between replays."""
import pytest
import torch

from workloads import problems

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _stepwise_route(monkeypatch):
    """These tests are about the launch graph of a STEPWISE solve. Their little SDEs (mu * y, 0.2 * y) are per-channel
    expressions, which the default route would run as one trajectory launch (tests/test_gpu_recognise.py): switched off."""
    from torchsde_amd import recognise
    monkeypatch.setattr(recognise, "ENABLED", False)
DEV = "cuda"


@pytest.mark.parametrize("prob,method,levy,shape", [
    ("gbm_ito", "euler", "none", (256, 16, 16)),
    ("gbm_strat", "midpoint", "none", (256, 16, 16)),
    ("gbm_ito", "srk", "space-time", (128, 8, 8)),
    ("general_ito", "euler", "none", (128, 4, 4)),
    ("gbm_ito", "milstein", "none", (128, 8, 8)),
    ("gbm_strat", "reversible_heun", "none", (128, 8, 8)),     # carries (f, g, z) between steps
    ("general_strat", "reversible_heun", "none", (64, 4, 4)),
])
def test_graph_replay_equals_eager(prob, method, levy, shape):
    import torchsde_amd
    B, d, m = shape
    steps, dt = 24, 2.0 ** -6
    ts = torch.tensor([0.0, 10 * dt, steps * dt], device=DEV)
    sde = problems.make(prob, d=d, m=m).to(DEV)
    grad_free = {"grad_free": True} if method == "milstein" else {}   # derivative form needs autograd in-loop

    def solve(entropy, y0, graph):
        bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(B, m), device=DEV, dtype=torch.float32,
                                           entropy=entropy, dt=dt, levy_area_approximation=levy)
        opts = dict(grad_free)
        if graph:
            opts["hip_graph"] = True
        with torch.no_grad():
            return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=dt, options=opts)

    y_a = torch.full((B, d), 0.1, device=DEV)
    y_b = torch.rand(B, d, device=DEV) * 0.2
    for entropy, y0 in [(1, y_a), (2, y_a), (3, y_b)]:        # first call captures, later calls replay
        assert torch.equal(solve(entropy, y0, True), solve(entropy, y0, False)), (entropy,)
    with torch.no_grad():                                      # parameters are read through pointers
        for p in sde.parameters():
            p.mul_(0.9)
    assert torch.equal(solve(4, y_b, True), solve(4, y_b, False))


def test_graph_replay_of_derivative_form_milstein():
    """The diffusion VJP of derivative-form Milstein runs through autograd INSIDE the captured region."""
    import torchsde_amd
    B, d, steps, dt = 128, 8, 16, 2.0 ** -6
    sde = problems.make("mlpdiag_ito", d=d).to(DEV)
    ts = torch.tensor([0.0, steps * dt], device=DEV)
    y0 = torch.full((B, d), 0.1, device=DEV)

    def solve(entropy, graph):
        bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(B, d), device=DEV, dtype=torch.float32,
                                           entropy=entropy, dt=dt)
        with torch.no_grad():
            return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="milstein", dt=dt,
                                       options={"hip_graph": True} if graph else None)
    for entropy in (1, 2):
        assert torch.equal(solve(entropy, True), solve(entropy, False))


@pytest.mark.parametrize("prob,method,adjoint_method,dt", [
    ("mlpdiag_ito", "euler", "euler", 2.0 ** -6),
    ("mlpdiag_ito", "milstein", None, 2.0 ** -6),            # default adjoint of Ito diagonal: Milstein
    ("mlpdiag_strat", "midpoint", None, 2.0 ** -6),
    ("general_strat", "midpoint", None, 2.0 ** -6),
    ("mlpdiag_strat", "midpoint", None, 0.013),               # reversed steps do not line up with the cells
    ("mlpdiag_strat", "reversible_heun", "adjoint_reversible_heun", 2.0 ** -6),   # stateful solver, exact adjoint
    ("general_strat", "reversible_heun", "adjoint_reversible_heun", 2.0 ** -6),
])
def test_adjoint_backward_graph_replay_equals_eager(prob, method, adjoint_method, dt):
    """`sdeint_adjoint(..., options={'hip_graph': True}, adjoint_options={'hip_graph': True})`: forward solve and
    backward sweep each replay one HIP graph; gradients equal the eager path bit for bit, across new Brownian seeds,
    new incoming gradients and in-place parameter updates (an optimiser step)."""
    import torchsde_amd
    B, d, m = 64, 4, 4
    T = 16 * 2.0 ** -6
    ts = torch.tensor([0.0, 0.4 * T, T], device=DEV)
    sde = problems.make(prob, d=d, m=m).to(DEV)

    def grads(entropy, weight, graph):
        y0 = torch.full((B, d), 0.1, device=DEV, requires_grad=True)
        bm = torchsde_amd.BrownianInterval(0.0, T, size=(B, m), device=DEV, dtype=torch.float32, entropy=entropy)
        opts = {"hip_graph": True} if graph else {}
        ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=bm, method=method, adjoint_method=adjoint_method, dt=dt,
                                         options=dict(opts), adjoint_options=dict(opts))
        sde.zero_grad()
        (ys * weight).sum().backward()
        return [ys.detach(), y0.grad] + [p.grad.clone() for p in sde.parameters()]

    w1 = torch.linspace(-1, 1, 3 * B * d, device=DEV).reshape(3, B, d)
    w2 = torch.rand(3, B, d, device=DEV)
    def check(entropy, w):
        got, ref = grads(entropy, w, True), grads(entropy, w, False)
        assert torch.equal(got[0], ref[0]), entropy      # ys: same bits
        # The gradients sum several autograd contributions per step (g.v, the Ito correction, its double backward);
        # the engine orders them by per-thread sequence numbers, and the recorded sweep was built on the caller's
        # thread while the eager one runs on the engine's worker thread: last-bit differences are expected.
        for a, b in zip(got[1:], ref[1:]):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-7)

    for entropy, w in [(5, w1), (6, w1), (7, w2)]:             # capture, replay, replay with new cotangents
        check(entropy, w)
    with torch.no_grad():
        for p in sde.parameters():
            p.mul_(0.95)
    check(8, w2)


@pytest.mark.parametrize("prob,method", [
    ("mlpdiag_ito", "euler"), ("mlpdiag_ito", "milstein"), ("mlpdiag_strat", "midpoint"),
    ("general_strat", "midpoint"), ("mlpdiag_strat", "reversible_heun"), ("mlpdiag_strat", "heun"),
])
def test_backprop_through_solver_graph_replay_equals_eager(prob, method):
    """`sdeint(..., options={'hip_graph': True})` with autograd ON: the forward solve (recorded with its autograd
    graph) and the back-propagation through it replay as two HIP graphs; same `ys` bits, same gradients up to the
    summation order of autograd, across new seeds, new initial states and in-place parameter updates."""
    import torchsde_amd
    B, d, m = 64, 4, 4
    dt = 2.0 ** -6
    T = 16 * dt
    ts = torch.tensor([0.0, 0.4 * T, T], device=DEV)
    sde = problems.make(prob, d=d, m=m).to(DEV)

    def grads(entropy, y_value, graph):
        y0 = torch.full((B, d), y_value, device=DEV, requires_grad=True)
        bm = torchsde_amd.BrownianInterval(0.0, T, size=(B, m), device=DEV, dtype=torch.float32, entropy=entropy)
        ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=dt, options={"hip_graph": True} if graph else {})
        sde.zero_grad()
        (ys ** 2).sum().backward()
        return [ys.detach(), y0.grad] + [p.grad.clone() for p in sde.parameters()]

    def check(entropy, y_value):
        got, ref = grads(entropy, y_value, True), grads(entropy, y_value, False)
        assert torch.equal(got[0], ref[0]), entropy
        for a, b in zip(got[1:], ref[1:]):
            torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6)

    for entropy, y_value in [(1, 0.1), (2, 0.1), (3, 0.2)]:
        check(entropy, y_value)
    with torch.no_grad():
        for p in sde.parameters():
            p.mul_(0.95)
    check(4, 0.2)


def test_backprop_graph_refuses_foreign_trainable_tensors():
    """A trainable tensor that is not a module parameter cannot receive a gradient from a replayed backward graph:
    such a solve is detected at capture time and runs eagerly (with a warning) instead of returning wrong grads."""
    import warnings
    import torchsde_amd
    B, d = 16, 4
    outside = torch.full((d,), 0.3, device=DEV, requires_grad=True)

    class Leaky(torch.nn.Module):
        noise_type, sde_type = "diagonal", "ito"

        def f(self, t, y):
            return -outside * y

        def g(self, t, y):
            return 0.2 * y

    y0 = torch.full((B, d), 0.5, device=DEV, requires_grad=True)
    ts = torch.tensor([0.0, 0.25], device=DEV)
    bm = torchsde_amd.BrownianInterval(0.0, 0.25, size=(B, d), device=DEV, dtype=torch.float32, entropy=1)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        ys = torchsde_amd.sdeint(Leaky(), y0, ts, bm=bm, method="euler", dt=2.0 ** -5, options={"hip_graph": True})
    assert any("running eagerly" in str(x.message) for x in w)
    ys.sum().backward()
    assert outside.grad is not None and outside.grad.abs().sum() > 0


def test_backprop_graph_detects_overwritten_activations():
    import torchsde_amd
    B, d = 32, 4
    sde = problems.make("mlpdiag_ito", d=d).to(DEV)
    ts = torch.tensor([0.0, 0.25], device=DEV)

    def solve(entropy):
        y0 = torch.full((B, d), 0.1, device=DEV, requires_grad=True)
        bm = torchsde_amd.BrownianInterval(0.0, 0.25, size=(B, d), device=DEV, dtype=torch.float32, entropy=entropy)
        return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="euler", dt=2.0 ** -5, options={"hip_graph": True})

    first, second = solve(1), solve(2)
    second.sum().backward()                       # the most recent solve can be differentiated
    with pytest.raises(RuntimeError, match="overwritten by a later solve"):
        first.sum().backward()


def test_latent_sde_training_step_with_graphs():
    """The latent-SDE training pattern of the reference's examples (`sdeint(..., logqp=True)`, loss = reconstruction
    + KL, back-propagation through the solver): with hip_graph the (ys, logqp) pair and every gradient match the
    eager run, over several optimiser steps."""
    import torchsde_amd

    class Latent(torch.nn.Module):
        noise_type, sde_type = "diagonal", "ito"

        def __init__(self):
            super().__init__()
            gen = torch.Generator().manual_seed(0)
            self.post = torch.nn.Linear(4, 4)
            self.prior = torch.nn.Linear(4, 4)
            with torch.no_grad():
                for p in self.parameters():
                    p.copy_(0.3 * torch.randn(p.shape, generator=gen))

        def f(self, t, y):       # posterior drift
            return torch.tanh(self.post(y))

        def h(self, t, y):       # prior drift
            return torch.tanh(self.prior(y))

        def g(self, t, y):
            return torch.full_like(y, 0.5)

    B = 64
    ts = torch.tensor([0.0, 0.25, 0.5], device=DEV)

    def train(graph, steps=3):
        torch.manual_seed(0)
        sde = Latent().to(DEV)
        opt = torch.optim.SGD(sde.parameters(), lr=0.1)
        losses = []
        for it in range(steps):
            y0 = torch.full((B, 4), 0.2, device=DEV)
            bm = torchsde_amd.BrownianInterval(0.0, 0.5, size=(B, 5), device=DEV, dtype=torch.float32, entropy=it)
            ys, logqp = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="euler", dt=2.0 ** -5, logqp=True,
                                            options={"hip_graph": True} if graph else None)
            loss = (ys[-1] ** 2).mean() + logqp.sum(0).mean()
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(loss.item())
        return losses, [p.detach().clone() for p in sde.parameters()]

    loss_g, params_g = train(True)
    loss_e, params_e = train(False)
    assert loss_g == pytest.approx(loss_e, rel=1e-5)
    for a, b in zip(params_g, params_e):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("method,adjoint_method", [("euler", None), ("reversible_heun", "adjoint_reversible_heun")])
def test_adjoint_graphs_with_logqp_names_and_extra(method, adjoint_method):
    """The wrappers of the reference's API (`logqp=True` augments the state with the KL column, `names=` renames the
    methods, `extra=True` returns the solver state) compose with the recorded forward and backward sweeps."""
    import torchsde_amd

    class Latent(torch.nn.Module):
        noise_type = "diagonal"

        def __init__(self, sde_type):
            super().__init__()
            self.sde_type = sde_type
            gen = torch.Generator().manual_seed(1)
            self.net = torch.nn.Linear(4, 4)
            self.prior_net = torch.nn.Linear(4, 4)
            with torch.no_grad():
                for p in self.parameters():
                    p.copy_(0.3 * torch.randn(p.shape, generator=gen))

        def posterior_drift(self, t, y):
            return torch.tanh(self.net(y))

        def prior_drift(self, t, y):
            return torch.tanh(self.prior_net(y))

        def diffusion(self, t, y):
            return 0.4 + 0.1 * torch.sigmoid(y)

    sde_type = "ito" if method == "euler" else "stratonovich"
    B = 32
    dt = 2.0 ** -5
    ts = torch.tensor([0.0, 8 * dt, 16 * dt], device=DEV)
    names = {"drift": "posterior_drift", "prior_drift": "prior_drift", "diffusion": "diffusion"}

    def run(graph, entropy):
        torch.manual_seed(0)
        sde = getattr(run, "sde", None)
        if sde is None:
            sde = run.sde = Latent(sde_type).to(DEV)
        y0 = torch.full((B, 4), 0.2, device=DEV, requires_grad=True)
        bm = torchsde_amd.BrownianInterval(0.0, 16 * dt, size=(B, 5), device=DEV, dtype=torch.float32, entropy=entropy)
        opts = {"hip_graph": True} if graph else {}
        out = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=bm, method=method, adjoint_method=adjoint_method, dt=dt,
                                          logqp=True, names=names, extra=True, options=dict(opts),
                                          adjoint_options=dict(opts))
        ys, logqp, extra = out
        sde.zero_grad()
        ((ys[-1] ** 2).mean() + logqp.sum(0).mean()).backward()
        return [ys.detach(), logqp.detach(), y0.grad] + [p.grad.clone() for p in sde.parameters()], extra

    for entropy in (1, 2, 3):
        (got, extra_g), (ref, extra_e) = run(True, entropy), run(False, entropy)
        assert len(extra_g) == len(extra_e)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), entropy
        for a, b in zip(got[2:], ref[2:]):
            torch.testing.assert_close(a, b, rtol=2e-5, atol=1e-6)

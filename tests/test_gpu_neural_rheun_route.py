"""`method="reversible_heun"` on UNCHANGED user modules whose drift and diffusion are networks of (t, y) takes the matrix-core
kernels (torchsde_amd/neural_rheun_route.py; ``-m gpu``): `sdeint` without and with autograd, and
`sdeint_adjoint(..., adjoint_method="adjoint_reversible_heun")` -- the reference's recommended training pair
(DOCUMENTATION.md:97,118; methods/reversible_heun.py:48-144) on the generator of examples/sde_gan.py:77-101 (restated in
tests/test_recognise.py) and on the reference's Neural* problems.

Pinned against the stepwise route (which replays the reference's `solver_rheun_*` / `adjoint_*_rheun` goldens,
tests/test_gpu_parity.py, test_gpu_adjoint.py), against the ORACLE's restatement of the pair on the same Brownian path
(oracle/solvers_ref.integrate_reversible_heun, oracle/adjoint_ref.reversible_heun_adjoint_gradients), and at the BASELINE
configs[2] size on sampled rows."""
import copy

import numpy as np
import pytest
import torch
from torch import nn

from tests import helpers
from tests.test_recognise import _GeneratorFunc
from workloads import configs, problems

pytestmark = pytest.mark.gpu
DEV = "cuda"
DT = 2.0 ** -5


def _bm(B, m, t1, entropy, dt=DT):
    import torchsde_amd
    return torchsde_amd.BrownianInterval(0.0, t1, size=(B, m), dtype=torch.float32, device=DEV, entropy=entropy, dt=dt)


def _book(sde):
    from torchsde_amd import solvers
    return getattr(sde, solvers.BaseSDESolver._RECOGNISED_ATTR, {"trusted": {}, "refused": {}})


def _launches(fn):
    from torchsde_amd import _native
    from torchsde_amd import kernels as K
    K.prof_begin(_native.KID_RHEUN_MLP, 256)
    out = fn()
    torch.cuda.synchronize()
    return out, K.prof_end()[1]


def _generator(num_layers, d=16, m=3, mlp=16, seed=0):
    torch.manual_seed(seed)
    return _GeneratorFunc(m, d, mlp, num_layers).to(DEV)


MODULES = {
    "sde_gan_1": (lambda: _generator(1), 16, 3),
    "sde_gan_2": (lambda: _generator(2), 16, 3),
    "sde_gan_3": (lambda: _generator(3, d=8, m=2, mlp=24), 8, 2),
    "neural_general": (lambda: problems.MLPGeneral(8, 5, "stratonovich", hidden=16).to(DEV), 8, 5),
    "neural_diagonal": (lambda: problems.make("netdiag_strat", d=12, hidden=16).to(DEV), 12, 12),
    "neural_scalar": (lambda: problems.make("netscalar_strat", d=6, hidden=8).to(DEV), 6, 1),
}


def _sdeint(sde, d, m, entropy, B=48, steps=16, stepwise=False, ts=None, grad=False, extra=False):
    import torchsde_amd
    y0 = torch.full((B, d), 0.2, device=DEV, requires_grad=grad)
    ts = torch.tensor([0.0, 5 * DT, steps * DT] if ts is None else ts, device=DEV)
    options = {"hip_graph": False}
    if stepwise:
        options["trajectory_kernel"] = False
    with torch.enable_grad() if grad else torch.no_grad():
        out = torchsde_amd.sdeint(sde, y0, ts, bm=_bm(B, m, float(ts[-1]), entropy), method="reversible_heun", dt=DT,
                                  options=options, extra=extra)
    return (out, y0) if grad else out


@pytest.mark.parametrize("name", sorted(MODULES))
def test_sampling_takes_one_launch(name):
    make, d, m = MODULES[name]
    sde = make()
    first = _sdeint(sde, d, m, 1)
    assert torch.equal(first, _sdeint(sde, d, m, 1, stepwise=True))                 # the verifying solve returns the stepwise one
    assert [v for k, v in _book(sde)["trusted"].items() if "kernels" in k] == [True], _book(sde)
    for entropy in (2, 3):
        fast, launches = _launches(lambda: _sdeint(sde, d, m, entropy))
        assert launches == 1
        slow = _sdeint(sde, d, m, entropy, stepwise=True)
        torch.testing.assert_close(fast, slow, rtol=5e-5, atol=5e-6)
        assert not torch.equal(fast[-1], fast[0])
    # an output time inside a step is interpolated in the kernel (base_solver.py:147)
    ts = [0.0, 3.4 * DT, 16 * DT]
    _sdeint(sde, d, m, 4, ts=ts)
    fast, launches = _launches(lambda: _sdeint(sde, d, m, 4, ts=ts))
    assert launches == 1
    torch.testing.assert_close(fast, _sdeint(sde, d, m, 4, ts=ts, stepwise=True), rtol=5e-5, atol=5e-6)
    # `extra=True`: the solver's final (f, g, z)
    ys, (f, g, z) = _sdeint(sde, d, m, 5, extra=True)
    ys_s, (f_s, g_s, z_s) = _sdeint(sde, d, m, 5, extra=True, stepwise=True)
    for a, b in ((ys, ys_s), (f, f_s), (g, g_s), (z, z_s)):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-5)


def _grads(sde, out, y0, weights):
    sde.zero_grad()
    (out * weights).sum().backward()
    return [y0.grad.clone()] + [torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in sde.parameters()]


def _same(got, want, rtol):
    for i, (a, b) in enumerate(zip(got, want)):
        scale = b.abs().max().item() + 1e-12
        err = (a - b).abs().max().item()
        assert err <= rtol * scale + 1e-7, (i, tuple(a.shape), err, scale)


@pytest.mark.parametrize("name", ["sde_gan_1", "sde_gan_2", "neural_general", "neural_diagonal", "neural_scalar"])
def test_autograd_through_sdeint_takes_the_exact_gradient_sweep(name):
    """`sdeint(method="reversible_heun")` with autograd recording: the kernels' backward pass against back-propagation through
    the stepwise loop (base_solver.py:143-149 under autograd)."""
    make, d, m = MODULES[name]
    sde = make()
    gen = torch.Generator(device=DEV).manual_seed(11)
    weights = torch.randn(3, 48, d, device=DEV, generator=gen)
    out, y0 = _sdeint(sde, d, m, 1, grad=True)                                      # the verifying solve (values and gradients)
    assert type(out.grad_fn).__name__ != "ReversibleHeunFnBackward"
    assert [v for k, v in _book(sde)["trusted"].items() if "kernels" in k] == [True], _book(sde)
    out, y0 = _sdeint(sde, d, m, 2, grad=True)
    assert type(out.grad_fn).__name__ == "ReversibleHeunFnBackward"
    got = _grads(sde, out, y0, weights)
    out_s, y0_s = _sdeint(sde, d, m, 2, grad=True, stepwise=True)
    torch.testing.assert_close(out.detach(), out_s.detach(), rtol=5e-5, atol=5e-6)
    _same(got, _grads(sde, out_s, y0_s, weights), 2e-3)


def _adjoint(sde, d, m, entropy, B=40, steps=16, stepwise=False, ts=None, adjoint_params=None):
    import torchsde_amd
    y0 = torch.full((B, d), 0.2, device=DEV, requires_grad=True)
    ts = torch.tensor([0.0, 4 * DT, steps * DT] if ts is None else ts, device=DEV)
    opts = {"trajectory_kernel": False} if stepwise else None
    ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=_bm(B, m, float(ts[-1]), entropy), method="reversible_heun",
                                     adjoint_method="adjoint_reversible_heun", dt=DT, options=opts, adjoint_options=opts,
                                     adjoint_params=adjoint_params)
    return ys, y0


@pytest.mark.parametrize("name", ["sde_gan_1", "sde_gan_2", "sde_gan_3", "neural_general", "neural_diagonal", "neural_scalar"])
def test_sdeint_adjoint_with_the_reversible_pair(name):
    """Forward one launch, backward the reconstruction sweep on the matrix cores; gradients on the user's own nn.Linear tensors,
    against the stepwise pair (tsde_rheun_* kernels + autograd VJPs of the user's code) on the same path."""
    make, d, m = MODULES[name]
    sde = make()
    gen = torch.Generator(device=DEV).manual_seed(13)
    weights = torch.randn(3, 40, d, device=DEV, generator=gen)
    ys, y0 = _adjoint(sde, d, m, 1)                                                 # the verifying solve
    assert type(ys.grad_fn).__name__ != "ReversibleHeunFnBackward"
    verdicts = [v for k, v in _book(sde)["trusted"].items() if "adjoint" in k]
    assert verdicts == [True], _book(sde)
    first = _grads(sde, ys, y0, weights)
    ys_s, y0_s = _adjoint(sde, d, m, 1, stepwise=True)
    _same(first, _grads(sde, ys_s, y0_s, weights), 1e-5)                            # (the stepwise result, with its gradients)
    (ys, y0), launches = _launches(lambda: _adjoint(sde, d, m, 2))
    assert type(ys.grad_fn).__name__ == "ReversibleHeunFnBackward" and launches == 1
    got = _grads(sde, ys, y0, weights)
    ys_s, y0_s = _adjoint(sde, d, m, 2, stepwise=True)
    torch.testing.assert_close(ys.detach(), ys_s.detach(), rtol=5e-5, atol=5e-6)
    _same(got, _grads(sde, ys_s, y0_s, weights), 2e-3)


def test_what_stays_on_the_stepwise_pair():
    make, d, m = MODULES["sde_gan_1"]
    sde = make()
    for entropy in (1, 2):                       # a narrower `adjoint_params`
        ys, _ = _adjoint(sde, d, m, entropy, adjoint_params=tuple(sde._drift.parameters()))
        assert type(ys.grad_fn).__name__ != "ReversibleHeunFnBackward"
    sde = make()
    for entropy in (1, 2):                       # an output time inside a step (the backward pass steps to every output)
        with pytest.warns(UserWarning):
            ys, _ = _adjoint(sde, d, m, entropy, ts=[0.0, 3.5 * DT, 16 * DT])
        assert type(ys.grad_fn).__name__ != "ReversibleHeunFnBackward"
    wide = _generator(1, d=16, m=3, mlp=72)      # hidden layers wider than 64, five Linear layers: no kernel
    deep = _generator(4, d=8, m=2, mlp=16)
    for sde, dd, mm in ((wide, 16, 3), (deep, 8, 2)):
        for entropy in (1, 2):
            out, launches = _launches(lambda: _sdeint(sde, dd, mm, entropy))
            assert launches == 0
        assert any("kernels" in str(k) for k in _book(sde)["refused"]), _book(sde)
    # a stop-gradient in front of the nets: followed for sampling, never with a gradient
    class Stopped(_GeneratorFunc):
        def f_and_g(self, t, x):
            return super().f_and_g(t, x.detach())
    torch.manual_seed(0)
    sde = Stopped(3, 16, 16, 1).to(DEV)
    for entropy in (1, 2):
        ys, _ = _adjoint(sde, 16, 3, entropy)
        assert type(ys.grad_fn).__name__ != "ReversibleHeunFnBackward"
    _sdeint(sde, 16, 3, 1)
    out, launches = _launches(lambda: _sdeint(sde, 16, 3, 2))
    assert launches == 1


@pytest.mark.parametrize("name", ["sde_gan_2", "neural_general"])
def test_the_pair_against_the_oracle(name):
    """Values, dL/dy0 and every parameter gradient against oracle/adjoint_ref.reversible_heun_adjoint_gradients (the
    restatement of adjoint.py:64-127 + reversible_heun.py:76-144 pinned by the reference's `adjoint_*_rheun` goldens), float64
    and float32 on the counter path of the same rows."""
    from oracle import adjoint_ref
    make, d, m = MODULES[name]
    sde = make()
    B, steps = 24, 16
    ts_list = [0.0, 4 * DT, steps * DT]
    gen = torch.Generator(device=DEV).manual_seed(17)
    weights = torch.randn(3, B, d, device=DEV, generator=gen)
    for entropy in (7, 7):                                                          # (the second solve: the kernels)
        ys, y0 = _adjoint(sde, d, m, entropy, B=B)
    assert type(ys.grad_fn).__name__ == "ReversibleHeunFnBackward"
    got = _grads(sde, ys, y0, weights)
    edges = np.arange(steps + 1) * DT
    refs = {}
    for dtype in (torch.float32, torch.float64):
        ref_sde = copy.deepcopy(sde).cpu().to(dtype)
        bm = helpers.counter_rows_bm(np.arange(B), m, 7, edges, dtype)
        y0_ref = torch.full((B, d), 0.2, dtype=dtype)
        ys_ref, gy, gp = adjoint_ref.reversible_heun_adjoint_gradients(
            ref_sde, y0_ref, torch.tensor(ts_list, dtype=dtype), bm, DT, weights.cpu().to(dtype))
        refs[dtype] = [ys_ref, gy] + list(gp)
    new = [ys.detach()] + got
    for i, (a, r32, r64) in enumerate(zip(new, refs[torch.float32], refs[torch.float64])):
        helpers.assert_within_reference_rounding(a, r32, r64, f"quantity {i}", factor=6.0, floor=2e-6)


def test_configs2_shape_rows_vs_oracle():
    """16384 x 32 x 16, hidden 64, 1000 steps of reversible Heun as ONE launch: sampled rows against the oracle's restatement of
    reversible_heun.py:61-73 on the same Brownian path."""
    import torchsde_amd
    from oracle import solvers_ref
    c = configs.WORKLOADS["c3_midpoint_general_default_route_b16384_d32_m16"]
    B, d, m, n, dt = c["B"], c["d"], c["m"], c["nsteps"], c["dt"]
    sde = configs.make_problem(c["problem"], d, m, DEV)
    y0 = torch.full((B, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, n * dt], device=DEV)

    def solve():
        bm = torchsde_amd.BrownianInterval(0.0, n * dt, size=(B, m), dtype=torch.float32, device=DEV, entropy=20240601, dt=dt)
        with torch.no_grad():
            return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="reversible_heun", dt=dt)
    solve()
    ys, launches = _launches(solve)
    assert launches == 1
    rows = helpers.sampled_rows(B, 24, seed=3, seams=(16, B - 16))
    edges = np.arange(n + 1) * dt
    out = {}
    for dtype in (torch.float32, torch.float64):
        ref_sde = copy.deepcopy(sde).cpu().to(dtype)
        bm = helpers.counter_rows_bm(rows, m, 20240601, edges, dtype)
        with torch.no_grad():
            out[dtype], _ = solvers_ref.integrate_reversible_heun(ref_sde, bm, torch.full((len(rows), d), 0.1, dtype=dtype),
                                                                  torch.tensor([0.0, n * dt], dtype=dtype), dt)
    new = ys[-1][torch.from_numpy(rows).to(DEV)]
    helpers.assert_within_reference_rounding(new, out[torch.float32][-1], out[torch.float64][-1], "configs[2] reversible Heun")


def _sdeint_method(sde, d, m, entropy, method, B=48, steps=16, stepwise=False, ts=None):
    import torchsde_amd
    y0 = torch.full((B, d), 0.2, device=DEV)
    ts = torch.tensor([0.0, 5.5 * DT, steps * DT] if ts is None else ts, device=DEV)
    options = {"hip_graph": False}
    if stepwise:
        options["trajectory_kernel"] = False
    with torch.no_grad():
        return torchsde_amd.sdeint(sde, y0, ts, bm=_bm(B, m, float(ts[-1]), entropy), method=method, dt=DT, options=options)


@pytest.mark.parametrize("method", ["midpoint", "heun", "euler_heun"])
@pytest.mark.parametrize("name", ["sde_gan_1", "sde_gan_3", "neural_general", "neural_diagonal", "neural_scalar"])
def test_the_stateless_stratonovich_schemes_on_deep_networks(name, method):
    """`tsde_deep_mlp_forward`: midpoint (the Stratonovich default of `sdeint`, sdeint.py:155), Heun, Euler-Heun for the same
    nets -- the sde_gan generator at depth 1 and 3, and the reference's Neural* problems under the two schemes the two-layer
    kernel does not have. One launch, against the stepwise route."""
    make, d, m = MODULES[name]
    sde = make()
    if method == "euler_heun" and name.startswith("sde_gan"):
        # Euler-Heun calls `g` alone (euler_heun.py:38); a module that only defines `f_and_g` raises in the reference as here
        with pytest.raises(RuntimeError, match="Method `g` has not been provided"):
            _sdeint_method(sde, d, m, 1, method)
        return
    first = _sdeint_method(sde, d, m, 1, method)
    assert torch.equal(first, _sdeint_method(sde, d, m, 1, method, stepwise=True))
    assert [v for v in _book(sde)["trusted"].values()] == [True], _book(sde)
    fast, launches = _launches(lambda: _sdeint_method(sde, d, m, 2, method))
    # (midpoint on a two-layer tanh / softplus net is the older kernel's: kernel family 8, not counted here)
    assert launches == (0 if (method == "midpoint" and name.startswith("neural")) else 1)
    torch.testing.assert_close(fast, _sdeint_method(sde, d, m, 2, method, stepwise=True), rtol=5e-5, atol=5e-6)


def test_euler_on_an_ito_deep_network():
    class ItoGenerator(type(MODULES["sde_gan_2"][0]())):
        sde_type = "ito"
    torch.manual_seed(0)
    sde = ItoGenerator(3, 16, 16, 2).to(DEV)
    _sdeint_method(sde, 16, 3, 1, "euler")
    fast, launches = _launches(lambda: _sdeint_method(sde, 16, 3, 2, "euler"))
    assert launches == 1
    torch.testing.assert_close(fast, _sdeint_method(sde, 16, 3, 2, "euler", stepwise=True), rtol=5e-5, atol=5e-6)
    # the default method of sdeint for general Ito noise IS Euler (sdeint.py:147-156): the drop-in call
    import torchsde_amd
    y0 = torch.full((48, 16), 0.2, device=DEV)
    ts = torch.tensor([0.0, 16 * DT], device=DEV)
    with torch.no_grad():
        out, launches = _launches(lambda: torchsde_amd.sdeint(sde, y0, ts, bm=_bm(48, 3, 16 * DT, 2), dt=DT))
    assert launches == 1

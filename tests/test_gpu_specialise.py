"""Expression programs compiled at run time (torchsde_amd/specialise.py; ``-m gpu``): the interpreter's instruction words as
straight-line code in the interpreter's own kernel. Same operations, same order, `-ffp-contract=off`: the SAME BITS as the
interpreter (which tests/test_gpu_programs.py pins against the stepwise route and the oracle)."""
import random

import pytest
import torch
from torch import nn

from workloads import problems

pytestmark = pytest.mark.gpu
DEV = "cuda"
B, D, STEPS, DT = 256, 8, 24, 2.0 ** -7


@pytest.fixture(autouse=True)
def _compile_in_the_calling_thread(monkeypatch, tmp_path_factory):
    from torchsde_amd import specialise
    monkeypatch.setattr(specialise, "MODE", "sync")
    monkeypatch.setenv("TSDE_SPECIALISE_CACHE", str(tmp_path_factory.getbasetemp() / "specialised"))
    yield


def _solve(sde, entropy, method, levy, m, dtype=torch.float32, stepwise=False, ts=None, d=D):
    import torchsde_amd
    y0 = torch.full((B, d), 0.1, device=DEV, dtype=dtype)
    ts = torch.tensor([0.0, 9.5 * DT, STEPS * DT] if ts is None else ts, device=DEV, dtype=dtype)
    bm = torchsde_amd.BrownianInterval(0.0, float(ts[-1]), size=(B, m), device=DEV, dtype=dtype, entropy=entropy,
                                       levy_area_approximation=levy)
    options = {"hip_graph": False}
    if stepwise:
        options["trajectory_kernel"] = False
    with torch.no_grad():
        return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=DT, options=options)


def _interpreted(fn):
    from torchsde_amd import specialise
    mode = specialise.MODE
    specialise.MODE = "0"
    try:
        return fn()
    finally:
        specialise.MODE = mode


@pytest.mark.parametrize("method,levy,sde_type", [("euler", "none", "ito"), ("milstein", "none", "ito"), ("srk", "space-time", "ito"),
                                                  ("midpoint", "none", "stratonovich"), ("heun", "none", "stratonovich")])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_exscalar_compiled_equals_interpreted_bit_for_bit(method, levy, sde_type, dtype):
    """The reference's ExScalar (tests/problems.py:75-103; scalar noise) under every scheme of the program kernel."""
    from torchsde_amd import specialise
    sde = problems.ScalarTrig(D, sde_type, dtype=dtype).to(DEV)
    _solve(sde, 1, method, levy, 1, dtype)               # earns trust for the route (stepwise result); compiles
    _solve(sde, 2, method, levy, 1, dtype)               # interpreter + compiled, compared: the program is verified
    done = [k for k, v in specialise.status().items() if isinstance(v, str) and v.endswith(".so")]
    assert done and any(specialise.verified(k) for k in done), specialise.status()
    fast = _solve(sde, 3, method, levy, 1, dtype)        # the compiled kernel
    slow = _interpreted(lambda: _solve(sde, 3, method, levy, 1, dtype))
    assert torch.equal(fast, slow)
    want = _solve(sde, 3, method, levy, 1, dtype, stepwise=True)
    torch.testing.assert_close(fast, want, **(dict(rtol=2e-5, atol=2e-6) if dtype == torch.float32 else dict(rtol=1e-10, atol=1e-12)))


@pytest.mark.parametrize("seed", [0, 15, 30, 45, 60, 75, 90, 105])
def test_random_expression_trees_compiled_equal_interpreted(seed):
    """The seeded random trees of tests/test_recognise.py (unary functions, powers, the four operations, constants, parameters,
    t; constants beyond the eight register rows) as user modules: compiled == interpreted, bit for bit, Euler and Milstein
    (the derivative program) and SRK."""
    import torch.nn.functional as F

    from tests.test_recognise import _M, _random_tree
    from torchsde_amd import solvers
    rng = random.Random(seed)
    src_f, src_g = _random_tree(rng, 5 if seed % 3 == 0 else 4), _random_tree(rng, 4 if seed % 3 == 0 else 3)
    if "y" not in src_f:
        src_f = f"({src_f}) * y"
    if "y" not in src_g:
        src_g = f"({src_g}) + torch.sin(y)"
    env = {"torch": torch, "F": F}
    # (bounded dynamics, as in tests/test_gpu_programs.py: 24 steps stay finite whatever the tree)
    f = eval(f"lambda s, t, y: torch.tanh({src_f}) - y", env)
    g = eval(f"lambda s, t, y: 0.3 * torch.tanh({src_g})", env)
    compared = 0
    for method, levy in (("euler", "none"), ("milstein", "none"), ("srk", "space-time")):
        sde = _M(f, g).to(DEV)
        sde.b = sde.b.to(DEV)
        d = sde.mu.numel()
        _solve(sde, 1, method, levy, d, d=d)
        book = getattr(sde, solvers.BaseSDESolver._RECOGNISED_ATTR)
        if book["refused"] or list(book["trusted"].values()) != [True]:
            continue                      # (a tree the program route does not take: tests/test_gpu_programs.py says why)
        _solve(sde, 2, method, levy, d, d=d)
        fast = _solve(sde, 3, method, levy, d, d=d)
        slow = _interpreted(lambda: _solve(sde, 3, method, levy, d, d=d))
        assert ((fast == slow) | (fast.isnan() & slow.isnan())).all(), (src_f, src_g, method)
        compared += 1
    assert compared > 0 or seed % 5 == 0


def test_without_a_compiler_or_switched_off_the_interpreter_runs(monkeypatch):
    from torchsde_amd import specialise
    sde = problems.ScalarTrig(D, "ito").to(DEV)
    monkeypatch.setattr(specialise, "compiler", lambda: None)
    before = dict(specialise.status())
    outs = [_solve(sde, e, "euler", "none", 1) for e in (1, 2, 2)]
    assert torch.equal(outs[1], outs[2]) and specialise.status() == before
    monkeypatch.undo()
    monkeypatch.setattr(specialise, "MODE", "0")
    assert torch.equal(_solve(sde, 2, "euler", "none", 1), outs[1])


def test_a_compiled_kernel_that_disagrees_is_never_used(monkeypatch):
    """The first launch of a compiled program runs beside the interpreter; a library whose result differs (here: one compiled
    from a deliberately altered unit) is marked and the interpreter keeps running."""
    from torchsde_amd import specialise
    true_source = specialise.source

    def wrong(*args, **kwargs):
        return true_source(*args, **kwargs).replace("return sin(v);", "return sin(v) * (T)1.0001;")
    monkeypatch.setattr(specialise, "source", wrong)
    sde = problems.ScalarTrig(D, "ito").to(DEV)
    for e in (1, 2, 3):
        got = _solve(sde, e, "euler", "none", 1)
    assert any(specialise.verified(k) is False for k in specialise.status())
    assert torch.equal(got, _interpreted(lambda: _solve(sde, 3, "euler", "none", 1)))


@pytest.mark.parametrize("method,levy", [("euler", "none"), ("milstein", "none"), ("srk", "space-time")])
def test_training_through_sdeint_compiled_sensitivities_equal_interpreted(method, levy):
    """`sdeint` with autograd recording on the reference's ExScalar: the programs on dual numbers
    (`tsde_trajectory_prog_diag_sens`) as generated code -- values, dL/dy0 and the parameter gradient bit for bit the
    interpreter's."""
    import torchsde_amd
    sde = problems.ScalarTrig(D, "ito").to(DEV)
    ts = torch.tensor([0.0, 7 * DT, STEPS * DT], device=DEV)
    gen = torch.Generator(device=DEV).manual_seed(5)
    weights = torch.randn(3, B, D, device=DEV, generator=gen)

    def train(entropy):
        y0 = torch.full((B, D), 0.1, device=DEV, requires_grad=True)
        sde.zero_grad()
        bm = torchsde_amd.BrownianInterval(0.0, STEPS * DT, size=(B, 1), device=DEV, entropy=entropy, levy_area_approximation=levy)
        ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=DT, options={"hip_graph": False})
        (ys * weights).sum().backward()
        return ys.detach(), y0.grad.clone(), sde.p.grad.clone(), type(ys.grad_fn).__name__
    train(1)                  # the verifying solve of the route (stepwise)
    train(2)                  # compiles; interpreter + compiled compared
    train(3)
    fast = train(4)
    assert fast[3].startswith("_ProgTrajectoryFn")
    slow = _interpreted(lambda: train(4))
    for a, b in zip(fast[:3], slow[:3]):
        assert torch.equal(a, b)
    from torchsde_amd import specialise
    assert any(specialise.verified(k) for k in specialise.status())


@pytest.mark.parametrize("method,levy,sde_type,m", [("euler", "none", "ito", 3), ("srk", "space-time", "ito", 8),
                                                    ("midpoint", "none", "stratonovich", 16), ("milstein", "none", "ito", 4)])
def test_additive_noise_drift_program_compiled_equals_interpreted(method, levy, sde_type, m):
    """The reference's ExAdditive (tests/problems.py:106-132: drift and diffusion use t) on `tsde_trajectory_prog_additive` with
    the drift program compiled: every channel-count class, Euler / midpoint / SRK (SRA1)."""
    sde = problems.make("additive_ito" if sde_type == "ito" else "additive_strat", d=D, m=m).to(DEV)
    _solve(sde, 1, method, levy, m)
    _solve(sde, 2, method, levy, m)
    fast = _solve(sde, 3, method, levy, m)
    slow = _interpreted(lambda: _solve(sde, 3, method, levy, m))
    assert torch.equal(fast, slow)
    torch.testing.assert_close(fast, _solve(sde, 3, method, levy, m, stepwise=True), rtol=2e-5, atol=2e-6)
    from torchsde_amd import specialise
    assert any(specialise.verified(k) for k, v in specialise.status().items())

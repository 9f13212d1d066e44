"""The oracle's restatement of the reference's solver steps reproduces the REAL reference's outputs
(golden fixtures produced by tests/golden/make_golden.py from /root/reference) under replayed increments."""
import pytest
import torch

from oracle import solvers_ref
from tests import helpers


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("name", helpers.solver_cases())
def test_oracle_matches_reference(name, tag):
    case = helpers.Case(name, tag)
    sde = case.sde()
    bm = solvers_ref.ReplayBrownian(case.table())
    with torch.no_grad():
        if case.method == "reversible_heun":
            ys, _ = solvers_ref.integrate_reversible_heun(sde, bm, case.y0(), case.ts, case.dt)
        else:
            ys = solvers_ref.integrate(sde, bm, case.y0(), case.ts, case.dt, case.method, case.options)
    assert ys.shape == case.ys.shape
    if case.problem.startswith(("general", "readme", "mlpdiag")) or "additive" in case.problem or "scalar" in case.problem:
        # bmm / Linear layers: same library, same machine -> still expected equal; allow 2 ulp-scale slack
        torch.testing.assert_close(ys, case.ys, rtol=2e-6 if tag == "f32" else 1e-13, atol=1e-7 if tag == "f32" else 1e-15)
    else:
        assert torch.equal(ys, case.ys), f"max diff {(ys - case.ys).abs().max().item():.3e}"


def _adaptive_cases():
    import os
    return sorted(f[len("adaptive_"):-4] for f in os.listdir(helpers.GOLDEN) if f.startswith("adaptive_"))


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("name", _adaptive_cases())
def test_oracle_adaptive_matches_reference(name, tag):
    """Step-doubling adaptive stepping: same accepted steps and same outputs as the real reference when fed the
    increments the reference's own BrownianInterval produced (recorded per queried interval)."""
    case = helpers.Case(name, tag, prefix="adaptive_")
    bm = solvers_ref.ReplayBrownian(case.table())
    with torch.no_grad():
        ys = solvers_ref.integrate(case.sde(), bm, case.y0(), case.ts, case.dt, case.method, adaptive=True,
                                   rtol=float(case.z["rtol"]), atol=float(case.z["atol"]))
    torch.testing.assert_close(ys, case.ys, rtol=0, atol=0)


def _closed_form_cases():
    import os
    return sorted(f[len("closed_form_mlp_"):-4] for f in os.listdir(helpers.GOLDEN) if f.startswith("closed_form_mlp_"))


@pytest.mark.parametrize("name", _closed_form_cases())
def test_oracle_matches_reference_on_the_counter_path(name):
    """The oracle's stepping loop + the C twin of the counter RNG reproduce what the REAL reference computed for the
    perceptron-drift module on that path (tests/golden/make_golden.py: gen_closed_form), gradients included -- the
    chain the GPU tests of the trajectory kernels hang from."""
    import numpy as np

    from oracle import counter
    z = helpers.load(f"closed_form_mlp_{name}.npz")
    B, d, hidden, steps = (int(v) for v in z["shape"])
    dt, with_grads = float(z["dt"]), bool(z["with_grads"])
    sde = helpers.mlp_module_from(z, torch.float64, "cpu")
    edges = np.arange(steps + 1) * dt

    have_h = str(z["levy"]) != "none"

    def bm(ta, tb, return_U=False):
        W, U, _ = counter.query(B * d, int(z["entropy"]), edges, float(ta), float(tb), dtype=np.float32, have_h=have_h)
        W = torch.from_numpy(W).reshape(B, d).double()
        return (W, torch.from_numpy(U).reshape(B, d).double()) if return_U else W

    y0 = torch.tensor(z["y0"], requires_grad=with_grads)
    with torch.set_grad_enabled(with_grads):
        ys = solvers_ref.integrate(sde, bm, y0, torch.tensor(z["ts"]), dt, str(z["method"]), None)
    torch.testing.assert_close(ys.detach(), torch.tensor(z["ys"]), rtol=1e-12, atol=1e-13)
    if with_grads:
        (ys * torch.tensor(z["weights"])).sum().backward()
        torch.testing.assert_close(y0.grad, torch.tensor(z["grad__y0"]), rtol=1e-10, atol=1e-12)
        for pname, p in sde.named_parameters():
            torch.testing.assert_close(p.grad, torch.tensor(z["grad__" + pname]), rtol=1e-10, atol=1e-12)


def _affine_cases():
    import os
    return sorted(f[len("closed_form_affine_"):-4] for f in os.listdir(helpers.GOLDEN) if f.startswith("closed_form_affine_"))


@pytest.mark.parametrize("name", _affine_cases())
def test_oracle_matches_reference_on_the_counter_path_affine(name):
    """Same chain for the affine diagonal module: every scheme of its trajectory kernel (SRK with the space-time Levy
    area of the counter path included)."""
    import numpy as np

    import torchsde_amd
    from oracle import counter
    z = helpers.load(f"closed_form_affine_{name}.npz")
    B, d, steps = (int(v) for v in z["shape"])
    dt, levy = float(z["dt"]), str(z["levy"])
    sde = torchsde_amd.AffineDiagonalSDE(*(torch.tensor(z["param__" + k]) for k in
                                           ("drift_rate", "drift_shift", "diff_rate", "diff_shift")),
                                         sde_type=str(z["sde_type"]), dtype=torch.float64)
    edges = np.arange(steps + 1) * dt

    def bm(ta, tb, return_U=False):
        W, U, _ = counter.query(B * d, int(z["entropy"]), edges, float(ta), float(tb), dtype=np.float64,
                                have_h=levy != "none")
        W = torch.from_numpy(W).reshape(B, d)
        return (W, torch.from_numpy(U).reshape(B, d)) if return_U else W

    y0 = torch.tensor(z["y0"], requires_grad=True)
    ys = solvers_ref.integrate(sde, bm, y0, torch.tensor(z["ts"]), dt, str(z["method"]), None)
    torch.testing.assert_close(ys.detach(), torch.tensor(z["ys"]), rtol=1e-12, atol=1e-13)
    (ys * torch.tensor(z["weights"])).sum().backward()
    torch.testing.assert_close(y0.grad, torch.tensor(z["grad__y0"]), rtol=1e-10, atol=1e-12)
    for pname, p in sde.named_parameters():
        torch.testing.assert_close(p.grad, torch.tensor(z["grad__" + pname]), rtol=1e-10, atol=1e-12)


def _expr_cases():
    import os
    return sorted(f[len("closed_form_expr_"):-4] for f in os.listdir(helpers.GOLDEN) if f.startswith("closed_form_expr_"))


@pytest.mark.parametrize("name", _expr_cases())
def test_oracle_matches_reference_on_the_counter_path_expressions(name):
    """Same chain for the elementwise-expression module (three cases: the SDE of the reference's own benchmark)."""
    import numpy as np

    import torchsde_amd
    from oracle import counter
    z = helpers.load(f"closed_form_expr_{name}.npz")
    B, d, steps = (int(v) for v in z["shape"])
    dt, levy = float(z["dt"]), str(z["levy"])
    coefs = [torch.tensor(z["param__" + n]) for n in torchsde_amd.ElementwiseDiagonalSDE._NAMES]
    sde = torchsde_amd.ElementwiseDiagonalSDE(str(z["drift"]), str(z["diffusion"]), coefs[:4], coefs[4:],
                                              sde_type=str(z["sde_type"]), dtype=torch.float64)
    edges = np.arange(steps + 1) * dt

    def bm(ta, tb, return_U=False):
        W, U, _ = counter.query(B * d, int(z["entropy"]), edges, float(ta), float(tb), dtype=np.float64,
                                have_h=levy != "none")
        W = torch.from_numpy(W).reshape(B, d)
        return (W, torch.from_numpy(U).reshape(B, d)) if return_U else W

    with torch.no_grad():
        ys = solvers_ref.integrate(sde, bm, torch.tensor(z["y0"]), torch.tensor(z["ts"]), dt, str(z["method"]), None)
    torch.testing.assert_close(ys, torch.tensor(z["ys"]), rtol=1e-12, atol=1e-13)


def _poly3_cases():
    import os
    return sorted(f[len("recognised_poly3_"):-4] for f in os.listdir(helpers.GOLDEN) if f.startswith("recognised_poly3_"))


def poly3_module(z, dtype=torch.float64):
    """The plain user module of a recognised_poly3_* fixture (workloads.problems.DoubleWell / Logistic) with the fixture's
    parameter values."""
    from workloads import problems
    d = int(z["shape"][1])
    sde = problems.DoubleWell(d) if str(z["problem"]) == "DoubleWell" else problems.Logistic(d, str(z["sde_type"]))
    sde = sde.to(dtype)
    sde.sde_type = str(z["sde_type"])
    with torch.no_grad():
        for name, p in sde.named_parameters():
            p.copy_(torch.tensor(z["param__" + name]).to(dtype))
    return sde


@pytest.mark.parametrize("name", _poly3_cases())
def test_oracle_matches_reference_on_polynomial_user_modules(name):
    """The oracle's loop on the double-well / logistic modules, counter path, against the REAL reference's output."""
    import numpy as np

    from oracle import counter
    z = helpers.load(f"recognised_poly3_{name}.npz")
    B, d, steps = (int(v) for v in z["shape"])
    dt, levy = float(z["dt"]), str(z["levy"])
    edges = np.arange(steps + 1) * dt

    def bm(ta, tb, return_U=False):
        W, U, _ = counter.query(B * d, int(z["entropy"]), edges, float(ta), float(tb), dtype=np.float64,
                                have_h=levy != "none")
        W = torch.from_numpy(W).reshape(B, d)
        return (W, torch.from_numpy(U).reshape(B, d)) if return_U else W

    with torch.no_grad():
        ys = solvers_ref.integrate(poly3_module(z), bm, torch.tensor(z["y0"]), torch.tensor(z["ts"]), dt,
                                   str(z["method"]), None)
    torch.testing.assert_close(ys, torch.tensor(z["ys"]), rtol=1e-12, atol=1e-13)


def _additive_cases():
    import os
    return sorted(f[len("recognised_additive_"):-4] for f in os.listdir(helpers.GOLDEN) if f.startswith("recognised_additive_"))


def additive_module(z, dtype=torch.float64):
    """The plain user module of a recognised_additive_* fixture (workloads.problems.AdditiveDecay / AdditiveShared /
    MLPNetAdditive: the shapes of the reference's ExAdditive and NeuralAdditive) with the fixture's parameter values."""
    from workloads import problems
    _, d, _, m = (int(v) for v in z["shape"])
    sde_type = str(z["sde_type"])
    sde = {"AdditiveDecay": lambda: problems.AdditiveDecay(d, m, sde_type),
           "AdditiveShared": lambda: problems.AdditiveShared(d, m, sde_type),
           "MLPNetAdditive": lambda: problems.MLPNetAdditive(d, m, sde_type, hidden=8)}[str(z["problem"])]().to(dtype)
    sde.sde_type = sde_type
    with torch.no_grad():
        for name, p in sde.named_parameters():
            p.copy_(torch.tensor(z["param__" + name]).to(dtype))
    return sde


@pytest.mark.parametrize("name", _additive_cases())
def test_oracle_matches_reference_on_additive_user_modules(name):
    """The oracle's loop (Euler, Milstein, midpoint, SRK's additive step srk.py:90-111) on the additive-noise modules, counter
    path, against the REAL reference's output (make_golden.py gen_additive)."""
    import numpy as np

    from oracle import counter
    z = helpers.load(f"recognised_additive_{name}.npz")
    B, d, steps, m = (int(v) for v in z["shape"])
    dt, levy = float(z["dt"]), str(z["levy"])
    edges = np.arange(steps + 1) * dt

    def bm(ta, tb, return_U=False):
        W, U, _ = counter.query(B * m, int(z["entropy"]), edges, float(ta), float(tb), dtype=np.float64,
                                have_h=levy != "none")
        W = torch.from_numpy(W).reshape(B, m)
        return (W, torch.from_numpy(U).reshape(B, m)) if return_U else W

    with torch.no_grad():
        ys = solvers_ref.integrate(additive_module(z), bm, torch.tensor(z["y0"]), torch.tensor(z["ts"]), dt,
                                   str(z["method"]), None)
    torch.testing.assert_close(ys, torch.tensor(z["ys"]), rtol=1e-12, atol=1e-13)

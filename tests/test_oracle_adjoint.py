"""The oracle's restatement of the reference's stochastic adjoint reproduces the REAL reference's gradients
(golden fixtures from /root/reference under replayed increments)."""
import pytest
import torch

from oracle import adjoint_ref, solvers_ref
from tests import helpers

CASES = [n for n in helpers.adjoint_cases() if not n.endswith("_rheun")]


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("name", CASES)
def test_oracle_adjoint_matches_reference(name, tag):
    case = helpers.Case(name, tag, prefix="adjoint_")
    z = case.z
    adjoint_method = str(z["adjoint_method"]) or None
    sde = case.sde()
    table = case.table()
    bm = solvers_ref.ReplayBrownian(table)
    wt = torch.tensor(z[f"{tag}__loss_weights"], dtype=case.dtype)
    ys, grad_y0, grad_params = adjoint_ref.adjoint_gradients(sde, case.y0(), case.ts, bm, case.dt, case.method,
                                                             adjoint_method, wt)
    rtol, atol = (2e-5, 2e-6) if tag == "f32" else (1e-11, 1e-13)
    torch.testing.assert_close(ys, case.ys, rtol=rtol, atol=atol)
    torch.testing.assert_close(grad_y0, torch.tensor(z[f"{tag}__grad_y0"], dtype=case.dtype), rtol=rtol, atol=atol)
    for j, g in enumerate(grad_params):
        torch.testing.assert_close(g, torch.tensor(z[f"{tag}__grad_p{j}"], dtype=case.dtype), rtol=rtol,
                                   atol=atol * 10)


def _closed_form_adjoint_cases():
    import os
    return sorted(f[len("closed_form_adjoint_"):-4] for f in os.listdir(helpers.GOLDEN)
                  if f.startswith("closed_form_adjoint_"))


@pytest.mark.parametrize("name", _closed_form_adjoint_cases())
def test_oracle_adjoint_matches_reference_on_the_counter_path(name):
    """The oracle's adjoint + the C twin of the counter RNG reproduce what the REAL reference's
    `sdeint_adjoint(adjoint_method="euler")` computed for the perceptron-drift module on that path
    (tests/golden/make_golden.py: gen_closed_form_adjoint) -- the chain tsde_adjoint_mlp_diag hangs from."""
    import numpy as np

    from oracle import counter
    z = helpers.load(f"closed_form_adjoint_{name}.npz")
    B, d, hidden, steps = (int(v) for v in z["shape"])
    dt = float(z["dt"])
    sde = helpers.mlp_module_from(z, torch.float64, "cpu")
    edges = np.arange(steps + 1) * dt

    def bm(ta, tb, return_U=False):
        W, _, _ = counter.query(B * d, int(z["entropy"]), edges, float(ta), float(tb), dtype=np.float32, have_h=False)
        return torch.from_numpy(W).reshape(B, d).double()

    ys, grad_y0, grad_params = adjoint_ref.adjoint_gradients(sde, torch.tensor(z["y0"]), torch.tensor(z["ts"]), bm, dt,
                                                             str(z["method"]), str(z["adjoint_method"]),
                                                             torch.tensor(z["weights"]))
    torch.testing.assert_close(ys, torch.tensor(z["ys"]), rtol=1e-12, atol=1e-13)
    torch.testing.assert_close(grad_y0, torch.tensor(z["grad__y0"]), rtol=1e-10, atol=1e-12)
    for (pname, _), g in zip(sde.named_parameters(), grad_params):
        torch.testing.assert_close(g, torch.tensor(z["grad__" + pname]), rtol=1e-10, atol=1e-11)


def _adaptive_cases():
    import os
    return sorted(f[len("adjoint_adaptive_"):-4] for f in os.listdir(helpers.GOLDEN)
                  if f.startswith("adjoint_adaptive_") and not f.endswith("_rheun.npz"))


@pytest.mark.parametrize("name", _adaptive_cases())
def test_oracle_adaptive_adjoint_matches_reference(name):
    """`adjoint_adaptive=True`: the replay table holds only the intervals the reference asked for, so the oracle has to
    take the reference's accept / reject decisions to get through at all; then the gradients to rounding."""
    case = helpers.Case(name, "f64", prefix="adjoint_adaptive_")
    z = case.z
    sde = case.sde()
    bm = solvers_ref.ReplayBrownian(case.table())
    wt = torch.tensor(z["f64__loss_weights"])
    ys, grad_y0, grad_params = adjoint_ref.adjoint_gradients(
        sde, case.y0(), case.ts, bm, case.dt, case.method, str(z["adjoint_method"]) or None, wt, adjoint_adaptive=True,
        adjoint_rtol=float(z["adjoint_rtol"]), adjoint_atol=float(z["adjoint_atol"]), dt_min=float(z["dt_min"]))
    torch.testing.assert_close(ys, case.ys, rtol=1e-11, atol=1e-13)
    torch.testing.assert_close(grad_y0, torch.tensor(z["f64__grad_y0"]), rtol=1e-9, atol=1e-11)
    for j, g in enumerate(grad_params):
        torch.testing.assert_close(g, torch.tensor(z[f"f64__grad_p{j}"]), rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("name", [n for n in helpers.adjoint_cases() if n.endswith("_rheun")])
def test_oracle_reversible_heun_adjoint_matches_reference(name):
    case = helpers.Case(name, "f64", prefix="adjoint_")
    z = case.z
    sde = case.sde()
    bm = solvers_ref.ReplayBrownian(case.table())
    ys, grad_y0, grad_params = adjoint_ref.reversible_heun_adjoint_gradients(
        sde, case.y0(), case.ts, bm, case.dt, torch.tensor(z["f64__loss_weights"]))
    torch.testing.assert_close(ys, case.ys, rtol=1e-11, atol=1e-13)
    torch.testing.assert_close(grad_y0, torch.tensor(z["f64__grad_y0"]), rtol=1e-10, atol=1e-12)
    for j, g in enumerate(grad_params):
        torch.testing.assert_close(g, torch.tensor(z[f"f64__grad_p{j}"]), rtol=1e-10, atol=1e-11)


@pytest.mark.parametrize("name", ["mlpdiag_strat_rheun", "general_strat_rheun"])
def test_oracle_adaptive_reversible_heun_adjoint_matches_reference(name):
    case = helpers.Case(name, "f64", prefix="adjoint_adaptive_")
    z = case.z
    sde = case.sde()
    bm = solvers_ref.ReplayBrownian(case.table())
    ys, grad_y0, grad_params = adjoint_ref.reversible_heun_adjoint_gradients(
        sde, case.y0(), case.ts, bm, case.dt, torch.tensor(z["f64__loss_weights"]), adjoint_adaptive=True,
        adjoint_rtol=float(z["adjoint_rtol"]), adjoint_atol=float(z["adjoint_atol"]), dt_min=float(z["dt_min"]))
    torch.testing.assert_close(ys, case.ys, rtol=1e-11, atol=1e-13)
    torch.testing.assert_close(grad_y0, torch.tensor(z["f64__grad_y0"]), rtol=1e-9, atol=1e-11)
    for j, g in enumerate(grad_params):
        torch.testing.assert_close(g, torch.tensor(z[f"f64__grad_p{j}"]), rtol=1e-9, atol=1e-10)

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

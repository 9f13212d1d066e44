"""The oracle's restatement of the `logqp=True` path (base_sde.py:240-306, sdeint.py:142-144, 284-295) reproduces the
REAL reference's `ys` and log-ratio increments (tests/golden/logqp_*.npz) under replayed increments."""
import os

import pytest
import torch

from oracle import solvers_ref
from tests import helpers


def logqp_cases():
    return sorted(f[len("logqp_"):-4] for f in os.listdir(helpers.GOLDEN) if f.startswith("logqp_"))


def names_of(z):
    s = str(z["names"])
    return dict(kv.split("=") for kv in s.split(",")) if s else None


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("name", logqp_cases())
def test_oracle_logqp_matches_reference(name, tag):
    case = helpers.Case(name, tag, prefix="logqp_")
    bm = solvers_ref.ReplayBrownian(case.table())
    with torch.no_grad():
        ys, log_ratio = solvers_ref.integrate_logqp(case.sde(), bm, case.y0(), case.ts, case.dt, case.method,
                                                    names=names_of(case.z))
    want = torch.tensor(case.z[f"{tag}__log_ratio"], dtype=case.dtype)
    rtol, atol = (2e-5, 1e-6) if tag == "f32" else (1e-12, 1e-14)
    torch.testing.assert_close(ys, case.ys, rtol=rtol, atol=atol)
    torch.testing.assert_close(log_ratio, want, rtol=rtol, atol=atol)

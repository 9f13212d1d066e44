"""`logqp=True` on the HIP path against the REAL reference (``-m gpu``): `ys`, the log-ratio increments and -- for the
`sdeint_adjoint` / back-propagation cases -- every gradient of ``sum(ys * w) + sum(log_ratio * v)``, under replayed
increments (tests/golden/logqp_*.npz, written by tests/golden/make_golden.py from /root/reference).
Reference: base_sde.py:240-306 (`SDELogqp`), sdeint.py:142-144, 284-295, base_sde.py:212-237 (`names=`)."""
import pytest
import torch

from tests import helpers
from tests.test_oracle_logqp import logqp_cases, names_of

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("name", logqp_cases())
def test_logqp_matches_reference_golden(name, tag):
    import torchsde_amd
    case = helpers.Case(name, tag, prefix="logqp_")
    z = case.z
    mode = str(z["adjoint"])
    sde = case.sde(DEV)
    y0 = case.y0(DEV).requires_grad_(bool(mode))
    bm = helpers.make_replay_bm(case.table(DEV), (case.B, case.m), case.dtype, DEV, case.levy)
    kw = dict(bm=bm, method=case.method, dt=case.dt, logqp=True, names=names_of(z))
    if mode == "default":
        ys, log_ratio = torchsde_amd.sdeint_adjoint(sde, y0, case.ts.to(DEV), **kw)
    elif mode == "backprop":
        ys, log_ratio = torchsde_amd.sdeint(sde, y0, case.ts.to(DEV), **kw)
    else:
        with torch.no_grad():
            ys, log_ratio = torchsde_amd.sdeint(sde, y0, case.ts.to(DEV), **kw)
    rtol, atol = (3e-4, 3e-5) if tag == "f32" else (1e-9, 1e-11)
    assert ys.shape == case.ys.shape and log_ratio.shape == (len(case.ts) - 1, case.B)
    torch.testing.assert_close(ys.detach().cpu(), case.ys, rtol=rtol, atol=atol)
    torch.testing.assert_close(log_ratio.detach().cpu(), torch.tensor(z[f"{tag}__log_ratio"], dtype=case.dtype),
                               rtol=rtol, atol=atol)
    if not mode:
        return
    wy = torch.tensor(z[f"{tag}__wy"], dtype=case.dtype, device=DEV)
    wl = torch.tensor(z[f"{tag}__wl"], dtype=case.dtype, device=DEV)
    ((ys * wy).sum() + (log_ratio * wl).sum()).backward()
    ref = torch.tensor(z[f"{tag}__grad_y0"], dtype=case.dtype)
    scale = ref.abs().max().item()
    torch.testing.assert_close(y0.grad.cpu(), ref, rtol=rtol, atol=atol * max(1.0, scale))
    for j, p in enumerate(sde.parameters()):
        ref = torch.tensor(z[f"{tag}__grad_p{j}"], dtype=case.dtype)
        got = torch.zeros_like(ref) if p.grad is None else p.grad.cpu()
        torch.testing.assert_close(got, ref, rtol=rtol, atol=atol * 10 * max(1.0, ref.abs().max().item()))

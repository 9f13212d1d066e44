"""The trajectory kernels of the closed-form SDEs against the REAL reference (run with ``-m gpu``): ``tests/golden/
closed_form_mlp_*.npz`` and ``closed_form_affine_*.npz`` hold ``torchsde.sdeint`` outputs -- and, through autograd,
the gradients -- of the same module in float64, computed by the reference itself (tests/golden/make_golden.py:
gen_closed_form, gen_closed_form_affine) on the increments of the counter-RNG path that the kernels regenerate on the
GPU from (entropy, cell)."""
import os

import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu
DEV = "cuda"
CASES = sorted(f[len("closed_form_mlp_"):-4] for f in os.listdir(helpers.GOLDEN) if f.startswith("closed_form_mlp_"))


def test_fixtures_present():
    assert len(CASES) == 7


@pytest.mark.parametrize("name", CASES)
def test_trajectory_kernels_match_the_reference(name):
    import torchsde_amd
    z = helpers.load(f"closed_form_mlp_{name}.npz")
    B, d, hidden, steps = (int(v) for v in z["shape"])
    dt, with_grads = float(z["dt"]), bool(z["with_grads"])
    sde = helpers.mlp_module_from(z, torch.float32, DEV)
    y0 = torch.tensor(z["y0"], dtype=torch.float32, device=DEV, requires_grad=with_grads)
    ts = torch.tensor(z["ts"], dtype=torch.float32, device=DEV)
    bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(B, d), dtype=torch.float32, device=DEV,
                                       entropy=int(z["entropy"]), dt=dt, levy_area_approximation=str(z["levy"]))
    with torch.set_grad_enabled(with_grads):
        ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=str(z["method"]), dt=dt)
        if not with_grads:                   # the one-launch kernel and the stepwise path of this package agree too
            again = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=str(z["method"]), dt=dt,
                                        options={"trajectory_kernel": False})
            torch.testing.assert_close(ys, again, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(ys.detach().cpu().double(), torch.tensor(z["ys"]), rtol=1e-4, atol=1e-5)
    if not with_grads:
        return
    assert "MlpTrajectoryFn" in type(ys.grad_fn).__name__            # the kernels, not the stepwise path
    (ys * torch.tensor(z["weights"], dtype=torch.float32, device=DEV)).sum().backward()
    got = {name_: p.grad for name_, p in sde.named_parameters()}
    got["y0"] = y0.grad
    for key, grad in got.items():
        ref = torch.tensor(z["grad__" + key])
        err = (grad.detach().cpu().double() - ref).abs().max().item()
        assert err <= 1e-3 * ref.abs().max().item() + 1e-6, f"{key}: {err:.3e} vs scale {ref.abs().max().item():.3e}"


AFFINE_CASES = sorted(f[len("closed_form_affine_"):-4] for f in os.listdir(helpers.GOLDEN)
                      if f.startswith("closed_form_affine_"))


@pytest.mark.parametrize("dtype,rtol", [(torch.float64, 1e-9), (torch.float32, 2e-4)])
@pytest.mark.parametrize("name", AFFINE_CASES)
def test_affine_trajectory_kernel_matches_the_reference(name, dtype, rtol):
    """tsde_trajectory_affine_diag(_sens), all five schemes: outputs and the gradients of its forward-mode sensitivities
    against the reference's sdeint + autograd on the same path (float64 fixture; the float64 launch agrees to 1e-9)."""
    import torchsde_amd
    assert len(AFFINE_CASES) == 5
    z = helpers.load(f"closed_form_affine_{name}.npz")
    B, d, steps = (int(v) for v in z["shape"])
    dt = float(z["dt"])
    sde = torchsde_amd.AffineDiagonalSDE(*(torch.tensor(z["param__" + k]) for k in
                                           ("drift_rate", "drift_shift", "diff_rate", "diff_shift")),
                                         sde_type=str(z["sde_type"]), dtype=dtype).to(DEV)
    y0 = torch.tensor(z["y0"], dtype=dtype, device=DEV, requires_grad=True)
    ts = torch.tensor(z["ts"], dtype=dtype, device=DEV)
    bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(B, d), dtype=dtype, device=DEV, entropy=int(z["entropy"]),
                                       dt=dt, levy_area_approximation=str(z["levy"]))
    ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=str(z["method"]), dt=dt)
    assert "TrajectoryFn" in type(ys.grad_fn).__name__               # the kernel, not the stepwise path
    torch.testing.assert_close(ys.detach().cpu().double(), torch.tensor(z["ys"]), rtol=rtol, atol=rtol * 1e-1)
    (ys * torch.tensor(z["weights"], dtype=dtype, device=DEV)).sum().backward()
    got = {k: p.grad for k, p in sde.named_parameters()}
    got["y0"] = y0.grad
    for key, grad in got.items():
        ref = torch.tensor(z["grad__" + key])
        err = (grad.detach().cpu().double() - ref).abs().max().item()
        assert err <= 10 * rtol * ref.abs().max().item() + 1e-12, f"{key}: {err:.3e} vs {ref.abs().max().item():.3e}"

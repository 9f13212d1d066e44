"""Heun and Euler-Heun (the reference's Stratonovich predictor-corrector schemes, methods/heun.py:35-48 and
methods/euler_heun.py:29-42) on the elementwise trajectory kernels (``-m gpu``): an UNCHANGED diagonal- or scalar-noise user
module -- affine, single-function expression, cubic, expression program, coefficients that depend on t -- solved with
`method="heun"` / `"euler_heun"` is one launch; pinned against the stepwise route (tsde_step_diag + tsde_heun_final around
the user's torch code), against the ORACLE's restatement of the reference's loop on the same Brownian path at 65536 x 64 x
1000, and -- with autograd recording -- against back-propagation through the stepwise solver."""
import pytest
import torch

from tests import helpers
from tests.test_gpu_programs import _book, _launches, _solve
from workloads import problems

pytestmark = pytest.mark.gpu
DEV = "cuda"
D = 8


def _strat(sde):
    sde.sde_type = "stratonovich"
    return sde


MODULES = {
    "gbm (affine)": lambda: problems.make("gbm_strat", d=D),
    "f = y, g = exp(-y) (one function)": lambda: _strat(problems.ExpDiffusion()),
    "double well (cubic)": lambda: _strat(problems.DoubleWell(D)),
    "scheduled (coefficients of t)": lambda: _strat(problems.ScheduledDiag(D)),
    "ExScalar (program, scalar noise)": lambda: problems.ScalarTrig(D, "stratonovich"),
}


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("method", ["heun", "euler_heun"])
@pytest.mark.parametrize("name", sorted(MODULES))
def test_predictor_corrector_schemes_are_one_launch(name, method, dtype):
    sde = MODULES[name]().to(DEV).to(dtype)
    _solve(sde, 1, method, dtype=dtype, d=D)
    assert list(_book(sde)["trusted"].values()) == [True], _book(sde)
    fast, launches = _launches(lambda: _solve(sde, 2, method, dtype=dtype, d=D))
    assert launches == 1
    tol = dict(rtol=2e-5, atol=2e-6) if dtype == torch.float32 else dict(rtol=1e-11, atol=1e-12)
    # (f = y, g = exp(-y) leaves the reals on a few paths in 32 steps: the same elements on both routes)
    torch.testing.assert_close(fast, _solve(sde, 2, method, stepwise=True, dtype=dtype, d=D), equal_nan=True, **tol)


@pytest.mark.parametrize("method", ["heun", "euler_heun"])
def test_time_as_an_operand_of_the_programs(method):
    """The second evaluation happens at t1 = t0 + dt: the programs read that time."""
    from tests.test_gpu_programs import _TimeInTheArithmetic
    sde = _TimeInTheArithmetic("stratonovich").to(DEV)
    _solve(sde, 1, method)
    assert list(_book(sde)["trusted"].values()) == [True], _book(sde)
    fast, launches = _launches(lambda: _solve(sde, 2, method))
    assert launches == 1
    torch.testing.assert_close(fast, _solve(sde, 2, method, stepwise=True), rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("method", ["heun", "euler_heun"])
def test_full_size_rows_vs_oracle(method):
    """65536 x 64, 1000 steps of Stratonovich GBM: sampled rows against the oracle (heun.py:35-48, euler_heun.py:29-42 on the
    same Brownian path), with the bound of tests/test_gpu_full_size_oracle.py."""
    import torchsde_amd
    from tests.test_gpu_full_size_oracle import _bm, _oracle_forward
    Bf, d, n, dt = 65536, 64, 1000, 2.0 ** -10
    sde = problems.make("gbm_strat", d=d).to(DEV)
    y0 = torch.full((Bf, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, n * dt], device=DEV)
    with torch.no_grad():
        torchsde_amd.sdeint(sde, y0, ts, bm=_bm(Bf, d, n, dt, 5), method=method, dt=dt)
        ys, launches = _launches(lambda: torchsde_amd.sdeint(sde, y0, ts, bm=_bm(Bf, d, n, dt, 20240601), method=method, dt=dt))
    assert launches == 1 and list(_book(sde)["trusted"].values()) == [True], _book(sde)
    rows = helpers.sampled_rows(Bf, 48, seed=4, seams=(32, Bf - 32))
    ref32, ref64 = _oracle_forward(sde, rows, d, d, 20240601, n, dt, method, 0.1)
    helpers.assert_within_reference_rounding(ys[-1][torch.from_numpy(rows).to(DEV)], ref32[-1], ref64[-1],
                                             f"Stratonovich GBM, {method}, trajectory kernel")


@pytest.mark.parametrize("method", ["heun", "euler_heun"])
@pytest.mark.parametrize("name", ["gbm (affine)", "ExScalar (program, scalar noise)"])
def test_gradients_through_sdeint(name, method):
    """Autograd on: the sensitivity kernels (affine: tsde_trajectory_affine_diag_sens; programs on dual numbers) carry the new
    schemes too; gradients on y0 and on the module's parameters equal back-propagation through the stepwise solver."""
    from tests.test_gpu_programs import _train
    sde = MODULES[name]().to(DEV)
    _train(sde, 1, method, "none", False, torch.float32, d=D)               # the verifying solve
    book = _book(sde)
    assert list(book["trusted"].values()) == [True] and [k[-1] for k in book["trusted"]] == ["autograd"], book
    ys, gy, gp = _train(sde, 2, method, "none", False, torch.float32, d=D)
    assert type(ys.grad_fn).__name__ in ("_TrajectoryFnBackward", "_ProgTrajectoryFnBackward"), ys.grad_fn
    ys_s, gy_s, gp_s = _train(sde, 2, method, "none", True, torch.float32, d=D)
    torch.testing.assert_close(ys, ys_s, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(gy, gy_s, rtol=2e-4, atol=2e-5)
    for key in gp_s:
        torch.testing.assert_close(gp[key], gp_s[key], rtol=5e-4, atol=5e-5)

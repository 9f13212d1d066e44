"""GPU parity of the perceptron-drift sampling kernel (``tsde_trajectory_mlp_diag``, both layers on the f32 matrix
cores; run with ``-m gpu``) against the stepwise path of the same module (torch ``Linear`` layers between the per-step
kernels) on the same Brownian path. The two differ only in the summation order of the matrix products."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _sde(d, hidden, activation, sde_type="ito", seed=0, diffusion="affine"):
    import torchsde_amd
    torch.manual_seed(seed)
    sigmoid = diffusion == "sigmoid"
    sde = torchsde_amd.MLPDriftDiagonalSDE(d, hidden, activation=activation, sde_type=sde_type, diffusion=diffusion,
                                           diff_scale=0.4 if sigmoid else 1.0,
                                           diff_rate=(2.0 if sigmoid else 0.2) * torch.rand(d) - 0.1,
                                           diff_shift=0.1 + 0.2 * torch.rand(d))
    with torch.no_grad():       # asymmetric, well-scaled weights (a transposed operand cannot pass)
        sde.lin1.weight.copy_(torch.randn(hidden, d) / d ** 0.5)
        sde.lin2.weight.copy_(torch.randn(d, hidden) / hidden ** 0.5)
        sde.lin1.bias.copy_(0.3 * torch.randn(hidden))
        sde.lin2.bias.copy_(0.3 * torch.randn(d))
    return sde.to(DEV)


def _solve(sde, y0, ts, method, dt, entropy, trajectory, row_offset=0):
    import torchsde_amd
    bm = torchsde_amd.BrownianInterval(float(ts[0]), float(ts[-1]), size=tuple(y0.shape), dtype=y0.dtype, device=DEV,
                                       entropy=entropy, row_offset=row_offset)
    with torch.no_grad():
        return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=dt, options={"trajectory_kernel": trajectory})


@pytest.mark.parametrize("activation", ["tanh", "softplus"])
@pytest.mark.parametrize("d,hidden", [(32, 32), (64, 64), (128, 128), (32, 128), (128, 64), (64, 32),
                                      (4, 16), (8, 100), (20, 50), (100, 7), (36, 33),     # padded to the MFMA tiles
                                      (32, 256), (64, 256), (16, 200)])                     # wide hidden layers
@pytest.mark.parametrize("method,sde_type", [("euler", "ito"), ("milstein", "ito"), ("midpoint", "stratonovich")])
def test_matches_stepwise_path(method, sde_type, d, hidden, activation):
    B = 300                                   # not a multiple of the 32-row wave tile or the 128-row block
    sde = _sde(d, hidden, activation, sde_type=sde_type)
    y0 = (0.5 * torch.randn(B, d, generator=torch.Generator().manual_seed(1))).to(DEV)
    dt = 2.0 ** -5
    ts = torch.tensor([0.0, 4 * dt, 5 * dt, 16 * dt], device=DEV)
    fast = _solve(sde, y0, ts, method, dt, 3, trajectory=True)
    ref = _solve(sde, y0, ts, method, dt, 3, trajectory=False)
    assert fast.shape == (4, B, d) and torch.isfinite(fast).all() and torch.equal(fast[0], y0)
    torch.testing.assert_close(fast, ref, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("d,hidden", [(64, 64), (128, 128), (20, 50)])
@pytest.mark.parametrize("method,sde_type", [("euler", "ito"), ("milstein", "ito"), ("milstein", "stratonovich"),
                                             ("midpoint", "stratonovich")])
def test_sigmoid_diffusion_matches_stepwise_path(method, sde_type, d, hidden):
    """g = diff_scale * sigmoid(diff_rate * y + diff_shift): the per-channel diffusion of latent-SDE models."""
    B = 300
    sde = _sde(d, hidden, "softplus", sde_type=sde_type, diffusion="sigmoid")
    y0 = (0.5 * torch.randn(B, d, generator=torch.Generator().manual_seed(1))).to(DEV)
    dt = 2.0 ** -5
    ts = torch.tensor([0.0, 4 * dt, 16 * dt], device=DEV)
    fast = _solve(sde, y0, ts, method, dt, 3, trajectory=True)
    ref = _solve(sde, y0, ts, method, dt, 3, trajectory=False)
    torch.testing.assert_close(fast, ref, rtol=2e-4, atol=2e-5)


def test_stratonovich_milstein_and_long_solve():
    d, hidden, B = 64, 128, 4096
    sde = _sde(d, hidden, "tanh", sde_type="stratonovich")
    y0 = torch.full((B, d), 0.1, device=DEV)
    dt = 2.0 ** -7
    ts = torch.tensor([0.0, 128 * dt], device=DEV)
    fast = _solve(sde, y0, ts, "milstein", dt, 8, trajectory=True)
    ref = _solve(sde, y0, ts, "milstein", dt, 8, trajectory=False)
    torch.testing.assert_close(fast, ref, rtol=1e-3, atol=1e-4)


def test_sharding_invariance_and_fallbacks():
    """Rows solved with `row_offset` equal the same rows of the full solve bit for bit; with autograd on, gradients come
    out either way (tests/test_gpu_mlp_backward.py)."""
    import torchsde_amd
    d, hidden, B = 32, 64, 512
    sde = _sde(d, hidden, "softplus")
    y0 = (0.3 * torch.randn(B, d, generator=torch.Generator().manual_seed(2))).to(DEV)
    dt = 2.0 ** -5
    ts = torch.tensor([0.0, 8 * dt], device=DEV)
    full = _solve(sde, y0, ts, "euler", dt, 5, trajectory=True)
    part = _solve(sde, y0[200:328], ts, "euler", dt, 5, trajectory=True, row_offset=200)
    assert torch.equal(full[:, 200:328], part)
    y_grad = y0.clone().requires_grad_(True)
    bm = torchsde_amd.BrownianInterval(0.0, 8 * dt, size=(B, d), device=DEV, dtype=torch.float32, entropy=5)
    ys = torchsde_amd.sdeint(sde, y_grad, ts, bm=bm, method="euler", dt=dt)
    ys[-1].sum().backward()
    assert y_grad.grad is not None and sde.lin1.weight.grad is not None
    torch.testing.assert_close(ys.detach(), full, rtol=2e-4, atol=2e-5)


def test_c_abi_rejects_unsupported_shapes():
    from torchsde_amd import _native
    lib = _native.load()
    x = torch.zeros(64, 48, device=DEV)
    traj = _native.Traj()
    for d, hidden in ((6, 32), (132, 32), (128, 129), (32, 257)):
        args = (x.data_ptr(),) * 2 + (64, d, hidden) + (x.data_ptr(),) * 6 + (0, 1.0, 0, 0, traj, 1, 0, None, 0, None)
        assert lib.tsde_trajectory_mlp_diag(*args) != 0 and b"multiple of 4" in lib.tsde_last_error()


def test_training_paths_still_work_on_the_module():
    """`sdeint_adjoint` (its forward pass runs the sampling kernel, its backward sweep the stepwise adjoint) and
    back-propagation through the stepwise solver agree on the gradients of the same module."""
    import torchsde_amd
    d, hidden, B = 32, 64, 256
    sde = _sde(d, hidden, "tanh", sde_type="stratonovich")
    dt = 2.0 ** -7
    ts = torch.tensor([0.0, 32 * dt], device=DEV)

    def grads(fn, **kw):
        y0 = torch.full((B, d), 0.1, device=DEV, requires_grad=True)
        bm = torchsde_amd.BrownianInterval(0.0, 32 * dt, size=(B, d), device=DEV, dtype=torch.float32, entropy=4)
        sde.zero_grad()
        ys = fn(sde, y0, ts, bm=bm, method="midpoint", dt=dt, **kw)
        ys[-1].sum().backward()
        return ys.detach(), y0.grad, sde.lin1.weight.grad.clone()

    ys_a, gy_a, gw_a = grads(torchsde_amd.sdeint_adjoint)
    ys_b, gy_b, gw_b = grads(torchsde_amd.sdeint)
    torch.testing.assert_close(ys_a, ys_b, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(gy_a, gy_b, rtol=5e-2, atol=5e-3)
    assert ((gw_a - gw_b).abs().max() / gw_b.abs().max()).item() < 5e-2


@pytest.mark.parametrize("d,hidden", [(32, 64), (128, 128), (20, 50)])
@pytest.mark.parametrize("method,sde_type,levy", [("euler", "ito", "none"), ("milstein", "ito", "none"),
                                                  ("midpoint", "stratonovich", "none"), ("srk", "ito", "space-time")])
def test_output_times_inside_steps_are_interpolated_in_the_kernel(method, sde_type, levy, d, hidden):
    """Arbitrary `ts` (the common case: `linspace` against a `dt` that does not divide it): the reference interpolates
    linearly between the step boundaries either side (base_solver.py:147, interp.py:15-18). The kernel writes w0 y_k at
    the start of the step and adds w1 y_{k+1} at its end; several outputs may fall into one step."""
    import torchsde_amd
    B, dt = 150, 2.0 ** -5
    sde = _sde(d, hidden, "softplus", sde_type=sde_type, diffusion="sigmoid")
    y0 = (0.5 * torch.randn(B, d, generator=torch.Generator().manual_seed(4))).to(DEV)
    ts = torch.tensor([0.0, 0.3 * dt, 2.5 * dt, 2.75 * dt, 6 * dt, 9.01 * dt, 12.9 * dt], device=DEV)

    def solve(trajectory):
        bm = torchsde_amd.BrownianInterval(0.0, float(ts[-1]), size=(B, d), dtype=torch.float32, device=DEV, entropy=21,
                                           levy_area_approximation=levy)
        with torch.no_grad():
            return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=dt, options={"trajectory_kernel": trajectory})

    fast, ref = solve(True), solve(False)
    assert torch.isfinite(fast).all() and torch.equal(fast[0], y0)
    torch.testing.assert_close(fast, ref, rtol=2e-4, atol=2e-5)


def test_every_default_of_sdeint_takes_the_kernel():
    """`sdeint(sde, y0, torch.linspace(0, 1, 7))` as a user would write it: default method (SRK for diagonal Ito noise),
    default dt = 1e-3 (1001 float32 steps, the last one tiny), outputs inside steps."""
    import torchsde_amd
    from torchsde_amd import kernels as K
    B, d, hidden = 96, 32, 64
    sde = _sde(d, hidden, "tanh", diffusion="sigmoid")
    y0 = (0.5 * torch.randn(B, d, generator=torch.Generator().manual_seed(6))).to(DEV)
    ts = torch.linspace(0, 1, 7, device=DEV)
    launches = []
    original = K.trajectory_mlp_diag

    def spy(*args, **kwargs):
        launches.append(1)
        return original(*args, **kwargs)

    def solve(**options):
        bm = torchsde_amd.BrownianInterval(0.0, 1.0, size=(B, d), dtype=torch.float32, device=DEV, entropy=33,
                                           levy_area_approximation="space-time")
        with torch.no_grad():
            return torchsde_amd.sdeint(sde, y0, ts, bm=bm, options=options or None)

    K.trajectory_mlp_diag = spy
    try:
        fast = solve()
    finally:
        K.trajectory_mlp_diag = original
    assert launches == [1], "the default call did not take the one-launch kernel"
    ref = solve(trajectory_kernel=False)
    torch.testing.assert_close(fast, ref, rtol=5e-4, atol=5e-5)

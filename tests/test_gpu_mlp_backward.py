"""GPU parity of the gradient of the perceptron-drift trajectory kernel (``tsde_trajectory_mlp_diag_backward`` +
``tsde_gram_partials``; run with ``-m gpu``): ``sdeint`` with autograd on, Euler or Milstein, on an
``MLPDriftDiagonalSDE`` must
return the gradients autograd gives when it records the stepwise solve of the same module on the same Brownian path
(the reference's way: torchsde/_core/base_solver.py:114-134 + methods/euler.py:31-36 under ``loss.backward()``). The
two differ in the summation order of the matrix products (and of the batch reductions of the parameter gradients)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _sde(d, hidden, activation, seed=0, scalar_diffusion=False, sde_type="ito", diffusion="affine"):
    import torchsde_amd
    torch.manual_seed(seed)
    sigmoid = diffusion == "sigmoid"
    rate = 0.05 if scalar_diffusion else (2.0 if sigmoid else 0.2) * torch.rand(d) - 0.1
    shift = 0.2 if scalar_diffusion else 0.1 + 0.2 * torch.rand(d)
    sde = torchsde_amd.MLPDriftDiagonalSDE(d, hidden, activation=activation, diff_rate=rate, diff_shift=shift,
                                           sde_type=sde_type, diffusion=diffusion, diff_scale=0.4 if sigmoid else 1.0)
    with torch.no_grad():       # asymmetric, well-scaled weights (a transposed operand cannot pass)
        sde.lin1.weight.copy_(torch.randn(hidden, d) / d ** 0.5)
        sde.lin2.weight.copy_(torch.randn(d, hidden) / hidden ** 0.5)
        sde.lin1.bias.copy_(0.3 * torch.randn(hidden))
        sde.lin2.bias.copy_(0.3 * torch.randn(d))
    return sde.to(DEV)


def _gradients(sde, y0, ts, dt, entropy, trajectory, weights, method="euler"):
    """Loss = sum_j <weights[j], ys[j]> over ALL outputs (so every cotangent entry point is exercised)."""
    import torchsde_amd
    y = y0.clone().requires_grad_(True)
    bm = torchsde_amd.BrownianInterval(float(ts[0]), float(ts[-1]), size=tuple(y0.shape), dtype=y0.dtype, device=DEV,
                                       entropy=entropy)
    sde.zero_grad()
    ys = torchsde_amd.sdeint(sde, y, ts, bm=bm, method=method, dt=dt, options={"trajectory_kernel": trajectory})
    assert ("MlpTrajectoryFn" in type(ys.grad_fn).__name__) == trajectory
    (ys * weights).sum().backward()
    named = {name: p.grad.clone() for name, p in sde.named_parameters()}
    named["y0"] = y.grad.clone()
    return ys.detach(), named


def _assert_gradients_close(fast, ref, rtol=2e-3):
    for name, g_ref in ref.items():
        scale = g_ref.abs().max().item()
        err = (fast[name] - g_ref).abs().max().item()
        assert err <= rtol * scale + 1e-6, f"{name}: max error {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("activation", ["tanh", "softplus"])
@pytest.mark.parametrize("d,hidden", [(32, 32), (64, 64), (128, 128), (32, 128), (128, 64), (64, 32),
                                      (4, 16), (8, 100), (20, 52), (100, 8), (36, 44), (124, 120),    # padded tiles
                                      (32, 256), (64, 256), (16, 200)])       # wide hidden layers (d <= 64)
def test_gradients_match_autograd_of_the_stepwise_solve(d, hidden, activation):
    B = 300                                   # not a multiple of the 16-row wave tile or the 128-row block
    sde = _sde(d, hidden, activation)
    gen = torch.Generator().manual_seed(1)
    y0 = (0.5 * torch.randn(B, d, generator=gen)).to(DEV)
    dt = 2.0 ** -5
    ts = torch.tensor([0.0, 4 * dt, 5 * dt, 16 * dt], device=DEV)
    weights = torch.randn(4, B, d, generator=gen).to(DEV)
    ys_fast, fast = _gradients(sde, y0, ts, dt, 3, True, weights)
    ys_ref, ref = _gradients(sde, y0, ts, dt, 3, False, weights)
    torch.testing.assert_close(ys_fast, ys_ref, rtol=2e-4, atol=2e-5)
    assert set(fast) == set(ref) == {"lin1.weight", "lin1.bias", "lin2.weight", "lin2.bias", "diff_rate", "diff_shift",
                                     "y0"}
    _assert_gradients_close(fast, ref)


@pytest.mark.parametrize("d,hidden", [(32, 32), (64, 64), (128, 128), (20, 52), (124, 120)])
def test_sigmoid_diffusion_gradients_match_autograd_of_the_stepwise_solve(d, hidden):
    """The latent-SDE-style module of BASELINE configs[4]: perceptron drift, g = scale * sigmoid(w*y + b)."""
    B = 300
    sde = _sde(d, hidden, "softplus", diffusion="sigmoid")
    gen = torch.Generator().manual_seed(3)
    y0 = (0.5 * torch.randn(B, d, generator=gen)).to(DEV)
    dt = 2.0 ** -5
    ts = torch.tensor([0.0, 4 * dt, 16 * dt], device=DEV)
    weights = torch.randn(3, B, d, generator=gen).to(DEV)
    ys_fast, fast = _gradients(sde, y0, ts, dt, 4, True, weights)
    ys_ref, ref = _gradients(sde, y0, ts, dt, 4, False, weights)
    torch.testing.assert_close(ys_fast, ys_ref, rtol=2e-4, atol=2e-5)
    _assert_gradients_close(fast, ref)
    # Milstein with this diffusion has no reverse sweep: stepwise path, gradients all the same
    import torchsde_amd
    y = y0.clone().requires_grad_(True)
    bm = torchsde_amd.BrownianInterval(0.0, 16 * dt, size=(B, d), dtype=torch.float32, device=DEV, entropy=4)
    ys = torchsde_amd.sdeint(sde, y, ts, bm=bm, method="milstein", dt=dt)
    assert "MlpTrajectoryFn" not in type(ys.grad_fn).__name__


def test_the_two_statements_of_the_latent_sde_of_the_benchmark_agree():
    """bench.py's adjoint workload (a user module: nn.Sequential drift, 0.1 * sigmoid(w*y + b) diffusion) and its
    training workload (the closed-form module with the same parameter values) are the same SDE: bit-identical on the
    stepwise path, and the trajectory kernels reproduce the user module's solve and its autograd gradients."""
    from workloads import configs
    import torchsde_amd
    d, B, steps, dt = 128, 512, 24, 2.0 ** -9
    user = configs.make_problem("latent_diag", d, d, DEV)
    closed = configs.make_problem("latent_diag_closed_form", d, d, DEV)
    ts = torch.tensor([0.0, steps * dt], device=DEV)

    def solve(sde, options):
        y = torch.full((B, d), 0.1, device=DEV, requires_grad=True)
        bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(B, d), dtype=torch.float32, device=DEV, entropy=21,
                                           dt=dt)
        sde.zero_grad()
        ys = torchsde_amd.sdeint(sde, y, ts, bm=bm, method="euler", dt=dt, options=options)
        ys[-1].sum().backward()
        return ys.detach(), y.grad

    ys_user, gy_user = solve(user, None)
    ys_step, gy_step = solve(closed, {"trajectory_kernel": False})
    assert torch.equal(ys_user, ys_step) and torch.equal(gy_user, gy_step)
    ys_fast, gy_fast = solve(closed, None)
    torch.testing.assert_close(ys_fast, ys_user, rtol=2e-4, atol=2e-5)
    named = {"y0": (gy_fast, gy_user), "lin1.weight": (closed.lin1.weight.grad, user.net[0].weight.grad),
             "lin2.bias": (closed.lin2.bias.grad, user.net[2].bias.grad), "diff_rate": (closed.diff_rate.grad, user.w.grad),
             "diff_shift": (closed.diff_shift.grad, user.b.grad)}
    _assert_gradients_close({k: v[0] for k, v in named.items()}, {k: v[1] for k, v in named.items()})


@pytest.mark.parametrize("activation", ["tanh", "softplus"])
@pytest.mark.parametrize("d,hidden", [(64, 64), (128, 128), (20, 52), (124, 120)])
@pytest.mark.parametrize("sde_type", ["ito", "stratonovich"])
def test_milstein_gradients_match_autograd_of_the_stepwise_solve(sde_type, d, hidden, activation):
    B = 200
    sde = _sde(d, hidden, activation, sde_type=sde_type)
    gen = torch.Generator().manual_seed(2)
    y0 = (0.5 * torch.randn(B, d, generator=gen)).to(DEV)
    dt = 2.0 ** -5
    ts = torch.tensor([0.0, 3 * dt, 12 * dt], device=DEV)
    weights = torch.randn(3, B, d, generator=gen).to(DEV)
    ys_fast, fast = _gradients(sde, y0, ts, dt, 6, True, weights, method="milstein")
    ys_ref, ref = _gradients(sde, y0, ts, dt, 6, False, weights, method="milstein")
    torch.testing.assert_close(ys_fast, ys_ref, rtol=2e-4, atol=2e-5)
    _assert_gradients_close(fast, ref)


def test_chunked_sweep_and_scalar_diffusion_parameters(monkeypatch):
    """A stash budget of a few steps forces many chunks (state, cotangent pointer and accumulators carried between
    launches); scalar diffusion parameters receive the sum over channels."""
    from torchsde_amd import kernels as K
    d, hidden, B = 64, 32, 1000
    sde = _sde(d, hidden, "softplus", scalar_diffusion=True)
    gen = torch.Generator().manual_seed(4)
    y0 = (0.5 * torch.randn(B, d, generator=gen)).to(DEV)
    dt = 2.0 ** -6
    ts = torch.tensor([0.0, 3 * dt, 7 * dt, 8 * dt, 21 * dt, 40 * dt], device=DEV)
    weights = torch.randn(6, B, d, generator=gen).to(DEV)
    _, whole = _gradients(sde, y0, ts, dt, 9, True, weights)
    monkeypatch.setattr(K._MlpTrajectoryFn, "STASH_BYTES", 7 * B * (d + 2 * hidden) * 4)     # 7 steps per chunk
    _, chunked = _gradients(sde, y0, ts, dt, 9, True, weights)
    _, ref = _gradients(sde, y0, ts, dt, 9, False, weights)
    assert whole["diff_rate"].shape == ref["diff_rate"].shape == ()
    _assert_gradients_close(chunked, ref)
    _assert_gradients_close(whole, ref)
    # the sweep itself is chunk-invariant (same kernels, same order per row); only the weight sums regroup
    torch.testing.assert_close(chunked["y0"], whole["y0"], rtol=0, atol=0)


@pytest.mark.parametrize("method", ["euler", "milstein"])
def test_states_recomputed_per_chunk_give_the_same_bits(monkeypatch, method):
    """Without room for every step's state the forward launch keeps the chunk boundaries only and the backward pass
    re-runs the sampling kernel per chunk: same kernel, same increments -> every gradient bit-identical."""
    from torchsde_amd import kernels as K
    d, hidden, B = 32, 64, 500
    sde = _sde(d, hidden, "tanh")
    gen = torch.Generator().manual_seed(8)
    y0 = (0.5 * torch.randn(B, d, generator=gen)).to(DEV)
    dt = 2.0 ** -6
    ts = torch.tensor([0.0, 2 * dt, 9 * dt, 14 * dt, 37 * dt], device=DEV)       # outputs inside and on chunk edges
    weights = torch.randn(5, B, d, generator=gen).to(DEV)
    monkeypatch.setattr(K._MlpTrajectoryFn, "STASH_BYTES", 7 * B * (d + 2 * hidden) * 4)     # 7 steps per chunk
    ys_all, keep_all = _gradients(sde, y0, ts, dt, 13, True, weights, method=method)
    monkeypatch.setattr(K._MlpTrajectoryFn, "STATE_BYTES", 1)                                # nothing fits
    ys_ck, recomputed = _gradients(sde, y0, ts, dt, 13, True, weights, method=method)
    assert torch.equal(ys_all, ys_ck)
    for name in keep_all:
        assert torch.equal(keep_all[name], recomputed[name]), name
    _, ref = _gradients(sde, y0, ts, dt, 13, False, weights, method=method)
    _assert_gradients_close(recomputed, ref)


def test_long_solve_training_shape():
    """Many steps at a latent-SDE-like shape: error growth stays at the level of the summation-order difference."""
    d, hidden, B = 128, 128, 2048
    sde = _sde(d, hidden, "softplus")
    y0 = torch.full((B, d), 0.1, device=DEV)
    dt = 2.0 ** -9
    ts = torch.tensor([0.0, 200 * dt], device=DEV)
    weights = torch.ones(2, B, d, device=DEV)
    weights[0] = 0.0
    _, fast = _gradients(sde, y0, ts, dt, 11, True, weights)
    _, ref = _gradients(sde, y0, ts, dt, 11, False, weights)
    _assert_gradients_close(fast, ref, rtol=5e-3)


@pytest.mark.parametrize("activation", ["tanh", "softplus"])
def test_gradients_match_the_cpu_oracle_in_float64(activation):
    """The chain to the reference: its stepping loop restated on the CPU (oracle/solvers_ref.py, euler.py:31-36) in
    float64 under autograd, fed the SAME increments by the C twin of the generator (oracle/counter.py)."""
    import copy

    import numpy as np

    import torchsde_amd
    from oracle import counter, solvers_ref
    d, hidden, B, steps, dt, entropy = 32, 64, 64, 16, 2.0 ** -5, 777
    sde = _sde(d, hidden, activation)
    gen = torch.Generator().manual_seed(5)
    y0 = (0.5 * torch.randn(B, d, generator=gen)).to(DEV)
    weights = torch.randn(3, B, d, generator=gen).to(DEV)
    ts = torch.tensor([0.0, 5 * dt, steps * dt], device=DEV)

    y = y0.clone().requires_grad_(True)
    bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(B, d), dtype=torch.float32, device=DEV, entropy=entropy,
                                       dt=dt)
    sde.zero_grad()
    ys = torchsde_amd.sdeint(sde, y, ts, bm=bm, method="euler", dt=dt)
    assert "MlpTrajectoryFn" in type(ys.grad_fn).__name__
    (ys * weights).sum().backward()
    fast = {name: p.grad.detach().cpu().double() for name, p in sde.named_parameters()}
    fast["y0"] = y.grad.detach().cpu().double()

    ref_sde = copy.deepcopy(sde).cpu().double()
    ref_sde.zero_grad()
    edges = np.arange(steps + 1) * dt

    def bm_cpu(ta, tb, return_U=False):
        W, _, _ = counter.query(B * d, entropy, edges, float(ta), float(tb), dtype=np.float32, have_h=False)
        return torch.from_numpy(W).reshape(B, d).double()

    y_ref = y0.detach().cpu().double().requires_grad_(True)
    with torch.enable_grad():
        ys_ref = solvers_ref.integrate(ref_sde, bm_cpu, y_ref, ts.cpu().double(), dt, "euler", None)
        (ys_ref * weights.cpu().double()).sum().backward()
    torch.testing.assert_close(ys.detach().cpu().double(), ys_ref.detach(), rtol=1e-4, atol=1e-5)
    ref = {name: p.grad for name, p in ref_sde.named_parameters()}
    ref["y0"] = y_ref.grad
    _assert_gradients_close(fast, ref, rtol=1e-3)


@pytest.mark.parametrize("k,m,n", [(1, 4, 4), (17, 128, 128), (1000, 64, 32), (4099, 100, 7), (70000, 128, 64),
                                   (33000, 36, 128), (5000, 256, 64), (3000, 40, 200), (2000, 130, 129)])
def test_gram_matches_float64(k, m, n):
    from torchsde_amd import kernels as K
    gen = torch.Generator().manual_seed(k)
    a = torch.randn(k, m, generator=gen).to(DEV)
    b = torch.randn(k, n, generator=gen).to(DEV)
    ref = (a.double().t() @ b.double())
    out, sums = K.gram(a, b, column_sums=True)
    assert out.shape == (m, n) and out.dtype == torch.float32 and sums.shape == (m,)
    assert (out.double() - ref).abs().max().item() <= 1e-5 * (k ** 0.5) * 8 + 1e-5
    assert (sums.double() - a.double().sum(0)).abs().max().item() <= 1e-5 * (k ** 0.5) * 8 + 1e-5
    assert torch.equal(out, K.gram(a, b))          # fixed summation order: bit-reproducible
    # unaligned operands (a row offset of one float) take the scalar loader
    if k > 4 and m % 4 == 0:
        a1, b1 = a.reshape(-1)[1:1 + (k - 1) * m].reshape(k - 1, m), b[1:].contiguous()
        assert a1.data_ptr() % 16 != 0 and a1.is_contiguous()
        shifted = K.gram(a1, b1)
        assert (shifted.double() - a1.double().t() @ b1.double()).abs().max().item() <= 1e-5 * (k ** 0.5) * 8 + 1e-5


def test_falls_back_when_the_sweep_does_not_apply():
    """Midpoint, a hidden width that is not a multiple of 4, or an extra parameter on a subclass: gradients
    still come out (stepwise path), and agree with the Euler sweep where both apply."""
    import torchsde_amd
    d, B = 32, 128
    dt = 2.0 ** -5
    ts = torch.tensor([0.0, 8 * dt], device=DEV)

    def run(sde, method):
        y = torch.full((B, d), 0.2, device=DEV, requires_grad=True)
        bm = torchsde_amd.BrownianInterval(0.0, 8 * dt, size=(B, d), device=DEV, dtype=torch.float32, entropy=2)
        sde.zero_grad()
        ys = torchsde_amd.sdeint(sde, y, ts, bm=bm, method=method, dt=dt)
        ys[-1].sum().backward()
        return ys.grad_fn, y.grad

    fn, g = run(_sde(d, 64, "tanh"), "euler")
    assert "MlpTrajectoryFn" in type(fn).__name__ and torch.isfinite(g).all()
    fn, g = run(_sde(d, 64, "tanh"), "milstein")
    assert "MlpTrajectoryFn" in type(fn).__name__ and torch.isfinite(g).all()
    fn, g = run(_sde(d, 64, "tanh", sde_type="stratonovich"), "midpoint")
    assert "MlpTrajectoryFn" not in type(fn).__name__ and torch.isfinite(g).all()
    fn, g = run(_sde(d, 30, "tanh"), "euler")
    assert "MlpTrajectoryFn" not in type(fn).__name__ and torch.isfinite(g).all()
    fn, g = run(_sde(d, 260, "tanh"), "euler")                  # wider than the LDS takes
    assert "MlpTrajectoryFn" not in type(fn).__name__ and torch.isfinite(g).all()

    class Extra(torchsde_amd.MLPDriftDiagonalSDE):
        def __init__(self):
            super().__init__(d, 64)
            self.gain = torch.nn.Parameter(torch.tensor(1.0))

        def f(self, t, y):
            return self.gain * super().f(t, y)

        def closed_form(self, d, dtype, device):
            return super().closed_form(d, dtype, device) if float(self.gain.detach()) == 1.0 else None

    extra = Extra().to(DEV)
    fn, g = run(extra, "euler")
    assert "MlpTrajectoryFn" not in type(fn).__name__ and extra.gain.grad is not None


def test_c_abi_rejects_unsupported_arguments():
    from torchsde_amd import _native
    lib = _native.load()
    x = torch.zeros(64, 64, device=DEV)
    traj = _native.Traj()
    ptr = x.data_ptr()
    for d, hidden, fragment in ((6, 32, b"multiples of 4"), (32, 30, b"multiples of 4"), (132, 32, b"multiples of 4")):
        args = (ptr,) * 7 + (0, ptr, ptr, -1, 64, d, hidden) + (ptr,) * 5 + (0, 1.0, 0, 0, traj, 0, 0, 1, 0, None, 0, None)
        assert lib.tsde_trajectory_mlp_diag_backward(*args) != 0 and fragment in lib.tsde_last_error()
    args = (ptr,) * 7 + (0, ptr, ptr, -1, 64, 32, 32) + (ptr,) * 5 + (0, 1.0, 0, 0, traj, 0, 5, 1, 0, None, 0, None)
    assert lib.tsde_trajectory_mlp_diag_backward(*args) != 0 and b"k_hi" in lib.tsde_last_error()
    args = (ptr,) * 7 + (0, ptr, ptr, -1, 64, 32, 32) + (ptr,) * 5 + (0, 1.0, 0, 3, traj, 0, 0, 1, 0, None, 0, None)   # midpoint
    assert lib.tsde_trajectory_mlp_diag_backward(*args) != 0 and b"Euler or Milstein" in lib.tsde_last_error()
    args = (ptr,) * 7 + (0, ptr, ptr, -1, 64, 32, 32) + (ptr,) * 5 + (1, 1.0, 0, 1, traj, 0, 0, 1, 0, None, 0, None)
    assert lib.tsde_trajectory_mlp_diag_backward(*args) != 0 and b"sigmoid with Euler" in lib.tsde_last_error()
    assert lib.tsde_gram_partials(ptr, None, ptr, 129, ptr, 4, 10, 129, 4, 1, 0, None) != 0
    assert b"[1, 128]" in lib.tsde_last_error()

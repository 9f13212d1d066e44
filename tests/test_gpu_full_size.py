"""BASELINE.json's configurations at their FULL single-GPU sizes (run with ``-m gpu``), checked through
size-independent properties instead of an oracle run (the CPU oracle needs minutes at these sizes):

* linearity -- for a linear SDE, scaling y0 by 2 scales every float of the solution by exactly 2;
* the closed-form solution of geometric Brownian motion ON THE SAME PATH (W_T re-queried from the generator);
* additivity of the generator: the steps' increments sum to the increment of the whole interval;
* sharding invariance, HIP-graph replay == eager, trajectory kernel == stepwise path (all bit-exact);
* for the adjoint: d(sum y_T)/dy0 of a linear SDE equals y_T / y0.
"""
import pytest
import torch
from torch import nn

from workloads import problems

pytestmark = pytest.mark.gpu
DEV = "cuda"
DT = 2.0 ** -10


def _bm(B, m, n, entropy=31, levy="none", row_offset=0, dtype=torch.float32):
    import torchsde_amd
    return torchsde_amd.BrownianInterval(0.0, n * DT, size=(B, m), dtype=dtype, device=DEV, entropy=entropy, dt=DT,
                                         levy_area_approximation=levy, row_offset=row_offset)


def _sdeint(sde, y0, n, method, bm, options=None):
    import torchsde_amd
    ts = torch.tensor([0.0, n * DT], device=DEV, dtype=y0.dtype)
    with torch.no_grad():
        return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=DT, options=options)


def test_c2_euler_b65536_d64_s1000():
    """configs[1]: diagonal Ito Euler, batch 65536 x state 64 x 1000 steps."""
    B, d, n = 65536, 64, 1000
    sde = problems.make("gbm_ito", d=d).to(DEV)
    y0 = torch.full((B, d), 0.1, device=DEV)
    ys = _sdeint(sde, y0, n, "euler", _bm(B, d, n))
    assert ys.shape == (2, B, d) and torch.isfinite(ys).all() and torch.equal(ys[0], y0)
    # linearity (exact: every operation of the step commutes with a power-of-two scaling)
    assert torch.equal(_sdeint(sde, 2 * y0, n, "euler", _bm(B, d, n))[-1], 2 * ys[-1])
    # HIP-graph replay and row sharding reproduce the same bits
    assert torch.equal(_sdeint(sde, y0, n, "euler", _bm(B, d, n), {"hip_graph": True}), ys)
    half = _sdeint(sde, y0[B // 2:], n, "euler", _bm(B // 2, d, n, row_offset=B // 2))
    assert torch.equal(half[-1], ys[-1, B // 2:])
    # closed form on the same path: strong error of Euler at dt = 2^-10 (order 1/2) is ~1e-2 relative
    W_T = _bm(B, d, n)(0.0, n * DT)
    exact = sde.exact(y0, n * DT, W_T)
    rel = ((ys[-1] - exact).abs() / exact.abs()).mean().item()
    assert rel < 3e-2, rel
    # Milstein (order 1) on the same path is an order of magnitude closer
    ys_m = _sdeint(sde, y0, n, "milstein", _bm(B, d, n), {"hip_graph": True})
    rel_m = ((ys_m[-1] - exact).abs() / exact.abs()).mean().item()
    assert rel_m < 0.3 * rel, (rel, rel_m)


def test_c2_generator_additivity_at_full_size():
    """Sum of the 1000 cell increments == the increment of [0, T] (merge of all cells), 4M elements."""
    B, d, n = 65536, 64, 1000
    bm = _bm(B, d, n)
    total = torch.zeros(B, d, device=DEV, dtype=torch.float64)
    for k in range(0, n, 50):
        total += bm(k * DT, (k + 50) * DT).double()
    whole = bm(0.0, n * DT).double()
    assert (total - whole).abs().max().item() < 2e-4           # ~1000 fp32 merges of O(1) values
    assert abs(whole.var().item() - n * DT) < 0.01 * n * DT


def test_c2_closed_form_trajectory_kernel_full_size():
    import torchsde_amd
    B, d, n = 65536, 64, 1000
    gbm = problems.make("gbm_ito", d=d).to(DEV)
    closed = torchsde_amd.AffineDiagonalSDE(gbm.mu.detach(), 0.0, gbm.sigma.detach(), 0.0, dtype=torch.float32).to(DEV)
    y0 = torch.full((B, d), 0.1, device=DEV)
    stepwise = _sdeint(gbm, y0, n, "euler", _bm(B, d, n), {"hip_graph": True})
    assert torch.equal(_sdeint(closed, y0, n, "euler", _bm(B, d, n)), stepwise)
    srk = _sdeint(closed, y0, n, "srk", _bm(B, d, n, levy="space-time"))
    srk_stepwise = _sdeint(gbm, y0, n, "srk", _bm(B, d, n, levy="space-time"), {"hip_graph": True})
    assert torch.equal(srk, srk_stepwise)


class _LinearGeneral(nn.Module):
    """dy_i = -y_i/2 dt + y_i sum_j S_ij dW_j: general noise, linear in y."""
    noise_type, sde_type = "general", "ito"

    def __init__(self, d, m):
        super().__init__()
        gen = torch.Generator().manual_seed(3)
        self.S = nn.Parameter(0.3 * torch.rand(d, m, generator=gen) - 0.15)

    def f(self, t, y):
        return -0.5 * y

    def g(self, t, y):
        return y.unsqueeze(-1) * self.S


def test_c3_general_noise_b16384_d32_m16():
    """configs[2] shape (general noise, batch 16384 x state 32 x 16 Brownian channels), Euler, 1000 steps."""
    B, d, m, n = 16384, 32, 16, 1000
    sde = _LinearGeneral(d, m).to(DEV)
    y0 = torch.full((B, d), 0.5, device=DEV)
    ys = _sdeint(sde, y0, n, "euler", _bm(B, m, n), {"hip_graph": True})
    assert ys.shape == (2, B, d) and torch.isfinite(ys).all()
    assert torch.equal(_sdeint(sde, 2 * y0, n, "euler", _bm(B, m, n), {"hip_graph": True})[-1], 2 * ys[-1])
    half = _sdeint(sde, y0[B // 2:], n, "euler", _bm(B // 2, m, n, row_offset=B // 2))
    assert torch.equal(half[-1], ys[-1, B // 2:])
    # closed form on the same path: y_i(T) = y_i(0) exp((-1/2 - |S_i|^2/2) T + S_i . W_T)
    W_T = _bm(B, m, n)(0.0, n * DT)
    S = sde.S.detach()
    exact = y0 * torch.exp((-0.5 - 0.5 * (S ** 2).sum(1)) * (n * DT) + W_T @ S.t())
    rel = ((ys[-1] - exact).abs() / exact.abs()).mean().item()
    assert rel < 3e-2, rel


def test_c4_midpoint_per_gpu_shard_b32768_d64():
    """configs[3]: the per-GPU shard (262144 / 8 rows) of the Stratonovich midpoint run, as rank 3 would see it."""
    B, d, n, rank = 32768, 64, 1000, 3
    sde = problems.make("gbm_strat", d=d).to(DEV)
    y0 = torch.full((B, d), 0.1, device=DEV)
    lo = rank * B - 1024
    # on the stepwise route (graph replay / eager), then on the default route: the first default solve of the module also
    # runs stepwise and compares (solvers._integrate_recognised), the later ones are one trajectory launch each
    for route in ("stepwise", "default"):
        stepwise = {"trajectory_kernel": False} if route == "stepwise" else {}
        if route == "default":
            _sdeint(sde, y0[:64], 8, "midpoint", _bm(64, d, 8))
        ys = _sdeint(sde, y0, n, "midpoint", _bm(B, d, n, row_offset=rank * B),
                     dict(stepwise, hip_graph=True) if stepwise else None)
        assert torch.isfinite(ys).all()
        # rows of the shard are the rows a run over the global batch produces (sample 2048 global rows around the seam)
        seam = _sdeint(sde, torch.full((2048, d), 0.1, device=DEV), n, "midpoint", _bm(2048, d, n, row_offset=lo),
                       stepwise or None)
        assert torch.equal(seam[-1, 1024:], ys[-1, :1024]), route
    # Stratonovich GBM closed form (problems.GBMDiag subtracts the correction in f): order-1 strong error
    W_T = _bm(B, d, n, row_offset=rank * B)(0.0, n * DT)
    exact = sde.exact(y0, n * DT, W_T)
    rel = ((ys[-1] - exact).abs() / exact.abs()).mean().item()
    assert rel < 5e-3, rel


def test_c5_adjoint_b32768_d128_s500():
    """configs[4] shape: sdeint_adjoint, diagonal noise, batch 32768 x state 128, 500 steps forward + backward.
    For the linear SDE y_T = y0 * M(path) the gradient of sum(y_T) w.r.t. y0 is M = y_T / y0 elementwise."""
    import torchsde_amd
    B, d, n, dt = 32768, 128, 500, 2.0 ** -9
    sde = problems.make("gbm_strat", d=d).to(DEV)
    y0 = torch.full((B, d), 0.1, device=DEV, requires_grad=True)
    ts = torch.tensor([0.0, n * dt], device=DEV)
    bm = torchsde_amd.BrownianInterval(0.0, n * dt, size=(B, d), dtype=torch.float32, device=DEV, entropy=17, dt=dt)
    ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=bm, method="midpoint", adjoint_method="midpoint", dt=dt)
    ys[-1].sum().backward()
    multiplier = (ys[-1] / y0).detach()
    rel = ((y0.grad - multiplier).abs() / multiplier.abs()).mean().item()
    assert torch.isfinite(y0.grad).all() and rel < 1e-2, rel
    for p in sde.parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all()


def test_c5_latent_sde_through_the_trajectory_kernels_full_size():
    """The latent SDE of configs[4] at B 32768 x d 128 x 500 steps, stated as the closed-form module: the matrix-core
    kernels against the stepwise path of the same module on the same Brownian path -- the final state, the loss
    gradients w.r.t. y0 and all six parameters (autograd over the 500 recorded steps is the comparison), sharding
    invariance of a block of rows (bit-exact), and a second backward pass reproducing the first bit for bit."""
    from workloads import configs
    import torchsde_amd
    B, d, n, dt = 32768, 128, 500, 2.0 ** -9
    sde = configs.make_problem("latent_diag_closed_form", d, d, DEV)
    ts = torch.tensor([0.0, n * dt], device=DEV)

    def solve(options, rows=slice(None), row_offset=0):
        y = torch.full((B, d), 0.1, device=DEV)[rows].clone().requires_grad_(True)
        bm = torchsde_amd.BrownianInterval(0.0, n * dt, size=tuple(y.shape), dtype=torch.float32, device=DEV, entropy=77,
                                           dt=dt, row_offset=row_offset)
        sde.zero_grad()
        ys = torchsde_amd.sdeint(sde, y, ts, bm=bm, method="euler", dt=dt, options=options)
        (ys[-1] ** 2).mean().backward()
        grads = {name: p.grad.clone() for name, p in sde.named_parameters()}
        grads["y0"] = y.grad.clone()
        return ys[-1].detach(), grads

    fast, g_fast = solve(None)
    again, g_again = solve(None)
    assert torch.equal(fast, again) and all(torch.equal(g_fast[k], g_again[k]) for k in g_fast)
    part, _ = solve(None, rows=slice(4096, 4096 + 2048), row_offset=4096)
    assert torch.equal(part, fast[4096:4096 + 2048])
    ref, g_ref = solve({"trajectory_kernel": False})
    assert torch.isfinite(fast).all()
    torch.testing.assert_close(fast, ref, rtol=1e-3, atol=1e-4)
    for key, want in g_ref.items():
        err = (g_fast[key] - want).abs().max().item()
        assert err <= 5e-3 * want.abs().max().item() + 1e-9, f"{key}: {err:.3e} vs scale {want.abs().max().item():.3e}"

"""BASELINE.json's configurations at their FULL single-GPU sizes against the ORACLE on sampled rows (``-m gpu``).

Batch rows are independent and the counter generator is addressable by global row (`row_offset` on the device,
``elem0 = row * m`` in the oracle's C twin), so the oracle can integrate exactly the rows it is asked about: each test
runs the full-size solve on the GPU, picks ~64 global rows on both sides of every tiling / shard seam plus random
ones, integrates those rows on the CPU with the oracle's restatement of the reference (float32 and float64) on the
SAME Brownian path, and requires

    max |hip32 - ref64|  <=  4 * max |ref32 - ref64| + 1e-6 * scale.

A kernel that misbehaves only on large grids (two-tile blocks, the streaming variant, 64-bit indexing of outputs beyond
2^31 elements) fails here even though every small-shape test passes.
"""
import numpy as np
import pytest
import torch

from tests import helpers
from workloads import configs, problems

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True)
def _few_cpu_threads():
    """The oracle integrates 64-256 rows: torch's CPU ops on such sizes are several times SLOWER with the hundreds of
    threads of the GPU box's host than with 8."""
    before = torch.get_num_threads()
    torch.set_num_threads(min(8, before))
    yield
    torch.set_num_threads(before)


def _edges(n, dt):
    return np.arange(n + 1) * dt


def _oracle_forward(sde32, rows, d, m, entropy, n, dt, method, y0_value, levy=False, ts=None):
    """Final states (or all `ts`) of `rows` from the oracle in float32 and float64 (same parameter VALUES: the float64
    run uses the float32 parameters widened, so that the only difference is the arithmetic)."""
    import copy
    from oracle import solvers_ref
    out = {}
    for dtype in (torch.float32, torch.float64):
        sde = copy.deepcopy(sde32).cpu().to(dtype)      # same parameter values, widened
        bm = helpers.counter_rows_bm(rows, m, entropy, _edges(n, dt), dtype, levy=levy)
        y0 = torch.full((len(rows), d), y0_value, dtype=dtype)
        tt = torch.tensor([0.0, n * dt] if ts is None else ts, dtype=dtype)
        with torch.no_grad():
            out[dtype] = solvers_ref.integrate(sde, bm, y0, tt, dt, method)
    return out[torch.float32], out[torch.float64]


def _bm(B, m, n, dt, entropy, levy="none", row_offset=0):
    import torchsde_amd
    return torchsde_amd.BrownianInterval(0.0, n * dt, size=(B, m), dtype=torch.float32, device=DEV, entropy=entropy,
                                         dt=dt, levy_area_approximation=levy, row_offset=row_offset)


@pytest.mark.parametrize("launch", ["graph", "eager"])
def test_c2_euler_b65536_d64_s1000_rows_vs_oracle(launch):
    """configs[1]: diagonal Ito Euler, 65536 x 64 x 1000 steps (the headline workload, as bench.py runs it)."""
    import torchsde_amd
    c = configs.WORKLOADS["c2_euler_diag_b65536_d64_s1000"]
    B, d, n, dt = c["B"], c["d"], c["nsteps"], c["dt"]
    sde = configs.make_problem(c["problem"], d, d, DEV)
    y0 = torch.full((B, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, n * dt], device=DEV)
    with torch.no_grad():
        ys = torchsde_amd.sdeint(sde, y0, ts, bm=_bm(B, d, n, dt, 20240601), method="euler", dt=dt,
                                 options={"hip_graph": True} if launch == "graph" else None)
    # 256-thread blocks own 2 x 256 consecutive 16-byte groups = 2048 elements = 32 rows: rows either side of those
    rows = helpers.sampled_rows(B, 64, seed=2, seams=(32, 2048 * 32 // d, B - 32))
    ref32, ref64 = _oracle_forward(sde, rows, d, d, 20240601, n, dt, "euler", 0.1)
    new = ys[-1][torch.from_numpy(rows).to(DEV)]
    helpers.assert_within_reference_rounding(new, ref32[-1], ref64[-1], "C2 Euler final state")
    # for this elementwise SDE the step arithmetic is the reference's bit for bit: what differs is the increment
    # (hardware log/sin/cos vs libm), so the float32 oracle on the same path is close in absolute terms too
    assert (new.cpu() - ref32[-1]).abs().max().item() < 2e-6


def test_c2_milstein_srk_rows_vs_oracle():
    """The configs[1] shape through Milstein and SRK (north_star names all three steps), rows against the oracle."""
    import torchsde_amd
    B, d, n, dt = 65536, 64, 1000, 2.0 ** -10
    sde = configs.make_problem("gbm_ito", d, d, DEV)
    y0 = torch.full((B, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, n * dt], device=DEV)
    rows = helpers.sampled_rows(B, 48, seed=5, seams=(32,))
    idx = torch.from_numpy(rows).to(DEV)
    for method, levy in (("milstein", "none"), ("srk", "space-time")):
        with torch.no_grad():
            ys = torchsde_amd.sdeint(sde, y0, ts, bm=_bm(B, d, n, dt, 99, levy=levy), method=method, dt=dt,
                                     options={"hip_graph": True})
        ref32, ref64 = _oracle_forward(sde, rows, d, d, 99, n, dt, method, 0.1, levy=levy != "none")
        helpers.assert_within_reference_rounding(ys[-1][idx], ref32[-1], ref64[-1], f"C2 {method} final state")


def test_c3_euler_general_b16384_d32_m16_rows_vs_oracle():
    """configs[2] on the method the reference has for general noise (Euler): NeuralGeneral-style SDE, hidden 64."""
    import torchsde_amd
    c = configs.WORKLOADS["c3_euler_general_b16384_d32_m16"]
    B, d, m, n, dt = c["B"], c["d"], c["m"], c["nsteps"], c["dt"]
    sde = configs.make_problem(c["problem"], d, m, DEV)
    y0 = torch.full((B, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, n * dt], device=DEV)
    with torch.no_grad():
        ys = torchsde_amd.sdeint(sde, y0, ts, bm=_bm(B, m, n, dt, 20240601), method="euler", dt=dt,
                                 options={"hip_graph": True})
    rows = helpers.sampled_rows(B, 64, seed=3, seams=(4, 8, B - 4))     # one wave per row / per 256-group span
    ref32, ref64 = _oracle_forward(sde, rows, d, m, 20240601, n, dt, "euler", 0.1)
    helpers.assert_within_reference_rounding(ys[-1][torch.from_numpy(rows).to(DEV)], ref32[-1], ref64[-1],
                                             "C3 Euler-general final state")


def test_c4_midpoint_rank3_shard_rows_vs_oracle():
    """configs[3]: rank 3's shard (rows 98304 .. 131071 of 262144) of the Stratonovich midpoint run; the oracle
    integrates GLOBAL rows, so a wrong row offset anywhere in the shard fails."""
    import torchsde_amd
    c = configs.WORKLOADS["c4_midpoint_diag_b32768_d64"]
    B, d, n, dt, rank = c["B"], c["d"], c["nsteps"], c["dt"], 3
    sde = configs.make_problem(c["problem"], d, d, DEV)
    y0 = torch.full((B, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, n * dt], device=DEV)
    with torch.no_grad():
        ys = torchsde_amd.sdeint(sde, y0, ts, bm=_bm(B, d, n, dt, 20240601, row_offset=rank * B), method="midpoint",
                                 dt=dt, options={"hip_graph": True})
    local = helpers.sampled_rows(B, 64, seed=4, seams=(32,))
    ref32, ref64 = _oracle_forward(sde, local + rank * B, d, d, 20240601, n, dt, "midpoint", 0.1)
    helpers.assert_within_reference_rounding(ys[-1][torch.from_numpy(local).to(DEV)], ref32[-1], ref64[-1],
                                             "C4 midpoint shard final state")


def test_c4_full_262144_rows_streaming_variant_vs_oracle():
    """The whole configs[3] batch on one device (262144 x 64: 64 MiB per stream) and a 1M-row batch (256 MiB per stream,
    beyond the Infinity Cache: the nontemporal, uncapped-grid variant of the step kernels), 64 steps each."""
    import torchsde_amd
    d, n, dt = 64, 64, 2.0 ** -10
    sde = configs.make_problem("gbm_strat", d, d, DEV)
    for B in (262144, 1048576):
        y0 = torch.full((B, d), 0.1, device=DEV)
        ts = torch.tensor([0.0, n * dt], device=DEV)
        with torch.no_grad():
            ys = torchsde_amd.sdeint(sde, y0, ts, bm=_bm(B, d, n, dt, 7), method="midpoint", dt=dt)
        rows = helpers.sampled_rows(B, 48, seed=B, seams=(32, 65536, 8 * 2048 * 32 // d))
        ref32, ref64 = _oracle_forward(sde, rows, d, d, 7, n, dt, "midpoint", 0.1)
        helpers.assert_within_reference_rounding(ys[-1][torch.from_numpy(rows).to(DEV)], ref32[-1], ref64[-1],
                                                 f"midpoint B={B}")
        del ys, y0


def test_trajectory_kernel_outputs_beyond_2_31_elements_vs_oracle():
    """One launch of the closed-form trajectory kernel writing 600 output times of a 65536 x 64 state: 2.5e9 elements
    (10 GB), so output offsets need 64 bits; rows of early, middle and late outputs against the oracle."""
    import torchsde_amd
    B, d, n, dt = 65536, 64, 600, 2.0 ** -10
    gbm = problems.make("gbm_ito", d=d)
    closed = torchsde_amd.AffineDiagonalSDE(gbm.mu.detach(), 0.0, gbm.sigma.detach(), 0.0, dtype=torch.float32).to(DEV)
    y0 = torch.full((B, d), 0.1, device=DEV)
    ts = torch.arange(n + 1, device=DEV, dtype=torch.float32) * dt
    with torch.no_grad():
        ys = torchsde_amd.sdeint(closed, y0, ts, bm=_bm(B, d, n, dt, 5), method="euler", dt=dt)
    assert ys.numel() > 2 ** 31
    rows = helpers.sampled_rows(B, 32, seed=9)
    picks = [0, 1, 299, 511, 512, 513, 599, 600]            # 512 * 65536 * 64 = 2^31 elements
    ref32, ref64 = _oracle_forward(gbm, rows, d, d, 5, n, dt, "euler", 0.1, ts=[k * dt for k in range(n + 1)])
    idx = torch.from_numpy(rows).to(DEV)
    for k in picks:
        helpers.assert_within_reference_rounding(ys[k][idx], ref32[k], ref64[k], f"output {k}")
    del ys


def _loss_weights(B, d):
    """Deterministic, row-addressable loss weights w[r, c] = cos(0.37 r + 0.11 c) (float64 on the host)."""
    r = torch.arange(B, dtype=torch.float64).unsqueeze(1)
    c = torch.arange(d, dtype=torch.float64).unsqueeze(0)
    return torch.cos(0.37 * r + 0.11 * c)


def _oracle_adjoint(sde32, rows, d, entropy, n, dt, method, adjoint_method, wt_rows, levy=False):
    import copy
    from oracle import adjoint_ref
    out = {}
    for dtype in (torch.float32, torch.float64):
        sde = copy.deepcopy(sde32).cpu().to(dtype)
        bm = helpers.counter_rows_bm(rows, d, entropy, _edges(n, dt), dtype, levy=levy)
        y0 = torch.full((len(rows), d), 0.1, dtype=dtype)
        w = torch.stack([torch.zeros_like(wt_rows), wt_rows]).to(dtype)
        out[dtype] = adjoint_ref.adjoint_gradients(sde, y0, torch.tensor([0.0, n * dt], dtype=dtype), bm, dt, method,
                                                   adjoint_method, w)
    return out[torch.float32], out[torch.float64]


@pytest.mark.parametrize("launch", ["graph", "eager"])
def test_c5_sdeint_adjoint_latent_b32768_d128_s500_rows_vs_oracle(launch):
    """configs[4] AS BENCHMARKED: the Ito latent SDE (MLP drift, sigmoid diffusion), ``sdeint_adjoint`` with
    ``method="euler", adjoint_method="euler"``, 32768 x 128, 500 steps forward + backward. Final states and dL/dy0 of
    sampled rows against the oracle's restatement of the reference's adjoint (adjoint_sde.py:177-216, 296-323)."""
    import torchsde_amd
    c = configs.WORKLOADS["c5_adjoint_latent_b32768_d128_s500"]
    B, d, n, dt = c["B"], c["d"], c["nsteps"], c["dt"]
    sde = configs.make_problem(c["problem"], d, d, DEV)
    wt = _loss_weights(B, d)
    y0 = torch.full((B, d), 0.1, device=DEV, requires_grad=True)
    ts = torch.tensor([0.0, n * dt], device=DEV)
    gopt = {"hip_graph": True} if launch == "graph" else {}
    ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=_bm(B, d, n, dt, 20240601), method="euler", adjoint_method="euler",
                                     dt=dt, options=dict(gopt), adjoint_options=dict(gopt))
    (ys[-1] * wt.to(DEV, torch.float32)).sum().backward()
    rows = helpers.sampled_rows(B, 64, seed=6, seams=(16, 32))
    idx = torch.from_numpy(rows).to(DEV)
    (ys32, gy32, _), (ys64, gy64, _) = _oracle_adjoint(sde, rows, d, 20240601, n, dt, "euler", "euler",
                                                        wt[torch.from_numpy(rows)])
    helpers.assert_within_reference_rounding(ys[-1][idx], ys32[-1], ys64[-1], "C5 final state")
    helpers.assert_within_reference_rounding(y0.grad[idx], gy32, gy64, "C5 dL/dy0")
    full_param_grads = [p.grad.clone() for p in sde.parameters()]
    assert all(torch.isfinite(g).all() for g in full_param_grads)

    if launch == "graph":
        return
    # parameter gradients are sums over rows: the full-size ones equal the sum over 8 row shards (each a separate
    # sdeint_adjoint call with its global row offset) ...
    total = [torch.zeros_like(g, dtype=torch.float64) for g in full_param_grads]
    S = B // 8
    for r in range(8):
        sde.zero_grad()
        ys_r = torchsde_amd.sdeint_adjoint(sde, y0[r * S:(r + 1) * S].detach().requires_grad_(True), ts,
                                           bm=_bm(S, d, n, dt, 20240601, row_offset=r * S), method="euler",
                                           adjoint_method="euler", dt=dt)
        (ys_r[-1] * wt[r * S:(r + 1) * S].to(DEV, torch.float32)).sum().backward()
        for acc, p in zip(total, sde.parameters()):
            acc += p.grad.double()
    for (name, _), full, acc in zip(sde.named_parameters(), full_param_grads, total):
        err = (full.double() - acc).abs().max().item()
        assert err <= 2e-4 * acc.abs().max().item() + 1e-6, f"{name}: full-size gradient vs sum of shards: {err:.3e}"


def test_c5_parameter_gradients_b256_vs_oracle():
    """... and at 256 rows (global rows 4096 .. 4351 of the same path) they are the oracle's."""
    import torchsde_amd
    B, d, n, dt, r0 = 256, 128, 500, 2.0 ** -9, 4096
    sde = configs.make_problem("latent_diag", d, d, DEV)
    wt = _loss_weights(r0 + B, d)[r0:]
    y0 = torch.full((B, d), 0.1, device=DEV, requires_grad=True)
    ts = torch.tensor([0.0, n * dt], device=DEV)
    ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=_bm(B, d, n, dt, 20240601, row_offset=r0), method="euler",
                                     adjoint_method="euler", dt=dt)
    (ys[-1] * wt.to(DEV, torch.float32)).sum().backward()
    rows = np.arange(r0, r0 + B)
    (ys32, gy32, gp32), (ys64, gy64, gp64) = _oracle_adjoint(sde, rows, d, 20240601, n, dt, "euler", "euler", wt)
    helpers.assert_within_reference_rounding(ys[-1], ys32[-1], ys64[-1], "final state")
    helpers.assert_within_reference_rounding(y0.grad, gy32, gy64, "dL/dy0")
    for (name, p), g32, g64 in zip(sde.named_parameters(), gp32, gp64):
        helpers.assert_within_reference_rounding(p.grad, g32, g64, f"dL/d{name}")


@pytest.mark.parametrize("method,adjoint_method", [("euler", "euler"), ("euler", "milstein"), (None, None)])
def test_c5_sdeint_adjoint_on_the_closed_form_module_rows_vs_oracle(method, adjoint_method):
    """configs[4] through `sdeint_adjoint` -- (euler, euler), (euler, milstein) and with EVERY default (forward SRK,
    backward Milstein) -- with the latent SDE stated as the
    closed-form module: the forward sampling kernel and the stochastic adjoint on the matrix cores
    (tsde_adjoint_mlp_diag) at 32768 x 128 x 500 steps. Final states and dL/dy0 of sampled rows against the oracle's
    restatement of the reference's adjoint on the user-module statement of the same SDE; the six parameter gradients
    against the stepwise stochastic adjoint of the full batch (pinned to the oracle by the tests above)."""
    import torchsde_amd
    c = configs.WORKLOADS["c5_adjoint_mlp_b32768_d128_s500"]
    B, d, n, dt = c["B"], c["d"], c["nsteps"], c["dt"]
    closed = configs.make_problem(c["problem"], d, d, DEV)
    user = configs.make_problem("latent_diag", d, d, DEV)
    wt = _loss_weights(B, d)
    ts = torch.tensor([0.0, n * dt], device=DEV)

    def run(sde):
        y0 = torch.full((B, d), 0.1, device=DEV, requires_grad=True)
        sde.zero_grad()
        levy = "space-time" if method is None else "none"
        ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=_bm(B, d, n, dt, 20240601, levy=levy), method=method,
                                         adjoint_method=adjoint_method, dt=dt)
        (ys[-1] * wt.to(DEV, torch.float32)).sum().backward()
        return ys, y0.grad, [p.grad.clone() for p in sde.parameters()]

    ys, gy, gp = run(closed)
    assert type(ys.grad_fn).__name__.startswith("_MlpAdjointFn")
    rows = helpers.sampled_rows(B, 64, seed=8, seams=(16, 64, 128))
    idx = torch.from_numpy(rows).to(DEV)
    (ys32, gy32, _), (ys64, gy64, _) = _oracle_adjoint(user, rows, d, 20240601, n, dt, method or "srk",
                                                        adjoint_method or "milstein", wt[torch.from_numpy(rows)],
                                                        levy=method is None)
    # (the matrix products accumulate in another order than the oracle's float32 GEMMs: a wider factor than for the
    #  elementwise kernels, still relative to the oracle's own float32 rounding)
    helpers.assert_within_reference_rounding(ys[-1][idx], ys32[-1], ys64[-1], "final state", factor=8.0, floor=1e-5)
    helpers.assert_within_reference_rounding(gy[idx], gy32, gy64, "dL/dy0", factor=8.0, floor=1e-5)
    _, _, gp_user = run(user)
    names = [name for name, _ in closed.named_parameters()]
    # parameter order: closed-form (diff_rate, diff_shift, lin1.weight, lin1.bias, lin2.weight, lin2.bias) =
    # user module (w, b, net.0.weight, net.0.bias, net.2.weight, net.2.bias)
    for name, got, want in zip(names, gp, gp_user):
        err = (got - want).abs().max().item()
        assert err <= 2e-3 * want.abs().max().item() + 1e-6, f"{name}: {err:.3e} vs scale {want.abs().max().item():.3e}"


@pytest.mark.parametrize("name", ["c2_euler_expdiff_b65536_d64_s1000", "c2_euler_expdiff_closed_form_b65536_d64_s1000"])
def test_c2_reference_benchmark_sde_rows_vs_oracle(name):
    """SURVEY 8d's nonlinear second workload -- the SDE of the reference's own benchmark, f = y, g = exp(-y)
    (benchmarks/brownian.py:131-139) -- at the headline's shape, stepwise (user torch ops + tsde_step_diag) and as an
    elementwise-expression module (one launch of tsde_trajectory_expr_diag), rows against the oracle."""
    import torchsde_amd
    c = configs.WORKLOADS[name]
    B, d, n, dt = c["B"], c["d"], c["nsteps"], c["dt"]
    sde = configs.make_problem(c["problem"], d, d, DEV)
    y0 = torch.full((B, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, n * dt], device=DEV)
    with torch.no_grad():
        ys = torchsde_amd.sdeint(sde, y0, ts, bm=_bm(B, d, n, dt, 20240601), method="euler", dt=dt)
    assert bool(torch.isfinite(ys[-1]).all())
    rows = helpers.sampled_rows(B, 64, seed=8, seams=(32, 2048 * 32 // d, B - 32))
    plain = configs.make_problem("exp_diffusion", d, d, "cpu")          # the oracle integrates the plain torch module
    ref32, ref64 = _oracle_forward(plain, rows, d, d, 20240601, n, dt, "euler", 0.1)
    helpers.assert_within_reference_rounding(ys[-1][torch.from_numpy(rows).to(DEV)], ref32[-1], ref64[-1], name)


@pytest.mark.parametrize("method,levy", [("euler", "none"), ("srk", "space-time")])
def test_c5_sampling_kernel_b32768_d128_s500_rows_vs_oracle(method, levy):
    """The perceptron-drift sampling kernel (both layers on the f32 matrix cores, one launch per solve) at the configs[4]
    shape, Euler and SRK (three drift evaluations per step), rows against the oracle integrating the same module's
    torch statement of f and g."""
    import torchsde_amd
    c = configs.WORKLOADS["c5_sampling_mlp_b32768_d128_s500"]
    B, d, n, dt = c["B"], c["d"], c["nsteps"], c["dt"]
    sde = configs.make_problem(c["problem"], d, d, DEV)
    y0 = torch.full((B, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, 200 * dt, n * dt], device=DEV)
    with torch.no_grad():
        ys = torchsde_amd.sdeint(sde, y0, ts, bm=_bm(B, d, n, dt, 4242, levy=levy), method=method, dt=dt)
    # workgroups own 64-row (and, on small grids, 32-row) tiles: rows either side of the first and the last seam
    rows = helpers.sampled_rows(B, 64, seed=9, seams=(32, 64, 128, B - 64))
    ref32, ref64 = _oracle_forward(sde, rows, d, d, 4242, n, dt, method, 0.1, levy=levy != "none",
                                   ts=[0.0, 200 * dt, n * dt])
    idx = torch.from_numpy(rows).to(DEV)
    for k in (1, 2):
        helpers.assert_within_reference_rounding(ys[k][idx], ref32[k], ref64[k], f"C5 sampling kernel {method}, output {k}")


def test_c5_training_kernels_b32768_d128_s500_rows_vs_oracle():
    """The configs[4] SDE trained by back-propagation THROUGH the solver (`sdeint` + `loss.backward()`, the reference's
    discretise-then-optimise route): sampling kernel writing every step, reverse sweep and weight-gradient products on
    the matrix cores, 32768 x 128 x 500 steps. dL/dy0 of sampled rows against autograd through the oracle's Euler loop
    on the same path (rows are independent, so a row's input gradient needs only that row). (Parameter gradients sum
    over all rows; they are pinned at sizes the oracle can take whole, tests/test_gpu_mlp_backward.py.)"""
    import copy

    import torchsde_amd
    from oracle import solvers_ref
    c = configs.WORKLOADS["c5_training_mlp_b32768_d128_s500"]
    B, d, n, dt = c["B"], c["d"], c["nsteps"], c["dt"]
    closed = configs.make_problem(c["problem"], d, d, DEV)
    wt = _loss_weights(B, d)
    ts = torch.tensor([0.0, n * dt], device=DEV)
    y0 = torch.full((B, d), 0.1, device=DEV, requires_grad=True)
    ys = torchsde_amd.sdeint(closed, y0, ts, bm=_bm(B, d, n, dt, 777), method="euler", dt=dt)
    assert "Mlp" in type(ys.grad_fn).__name__, type(ys.grad_fn).__name__
    (ys[-1] * wt.to(DEV, torch.float32)).sum().backward()
    rows = helpers.sampled_rows(B, 48, seed=10, seams=(16, 64, 128))
    idx = torch.from_numpy(rows).to(DEV)
    got = {}
    for dtype in (torch.float32, torch.float64):
        sde = copy.deepcopy(closed).cpu().to(dtype)
        bm = helpers.counter_rows_bm(rows, d, 777, _edges(n, dt), dtype)
        y_rows = torch.full((len(rows), d), 0.1, dtype=dtype, requires_grad=True)
        out = solvers_ref.integrate(sde, bm, y_rows, torch.tensor([0.0, n * dt], dtype=dtype), dt, "euler")
        (out[-1] * wt[torch.from_numpy(rows)].to(dtype)).sum().backward()
        got[dtype] = (out[-1].detach(), y_rows.grad)
    helpers.assert_within_reference_rounding(ys[-1][idx].detach(), got[torch.float32][0], got[torch.float64][0],
                                             "final state", factor=8.0, floor=1e-5)
    helpers.assert_within_reference_rounding(y0.grad[idx], got[torch.float32][1], got[torch.float64][1], "dL/dy0",
                                             factor=8.0, floor=1e-5)

"""Additive noise on `tsde_trajectory_prog_additive` (``-m gpu``): an UNCHANGED user module whose drift is elementwise code
and whose diffusion depends on t only -- the reference's ExAdditive (tests/problems.py:106-132), a constant matrix handed back
as `sigma.expand(B, d, m)`, a network of t (the g of NeuralAdditive, tests/problems.py:195-224) -- takes one launch per solve:
Euler (euler.py:29-37), Milstein (the same step: base_sde.py:157-158), midpoint (midpoint.py:29-45) and the default method,
SRK = SRA1 (srk.py:90-111). Pinned against the stepwise route and, at 16384 x 32 x 8 x 500, against the ORACLE's restatement
of the reference's loop on the same Brownian path."""
import pytest
import torch
from torch import nn

from tests import helpers
from tests.test_gpu_programs import _book, _launches
from workloads import problems

pytestmark = pytest.mark.gpu
DEV = "cuda"
B, STEPS, DT = 256, 32, 2.0 ** -7

SCHEMES = [("euler", "ito", "none"), ("milstein", "ito", "none"), ("srk", "ito", "space-time"),
           ("midpoint", "stratonovich", "none"), ("milstein", "stratonovich", "none")]


def _solve(sde, entropy, method, levy, d, m, stepwise=False, dtype=torch.float32, rows=B, row_offset=0, ts=None):
    import torchsde_amd
    y0 = torch.full((rows, d), 0.3, device=DEV, dtype=dtype)
    ts = torch.tensor([0.0, 11.5 * DT, STEPS * DT] if ts is None else ts, device=DEV, dtype=dtype)
    bm = torchsde_amd.BrownianInterval(0.0, STEPS * DT, size=(rows, m), device=DEV, dtype=dtype, entropy=entropy, dt=DT,
                                       levy_area_approximation=levy, row_offset=row_offset)
    options = {"hip_graph": False}
    if stepwise:
        options["trajectory_kernel"] = False
    with torch.no_grad():
        return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=DT, options=options)


def _check(sde, method, levy, d, m, dtype=torch.float32):
    _solve(sde, 1, method, levy, d, m, dtype=dtype)                    # both routes, compared: earns the trust
    assert list(_book(sde)["trusted"].values()) == [True], _book(sde)
    fast, launches = _launches(lambda: _solve(sde, 2, method, levy, d, m, dtype=dtype))
    assert launches == 1
    tol = dict(rtol=2e-5, atol=2e-6) if dtype == torch.float32 else dict(rtol=1e-11, atol=1e-12)
    torch.testing.assert_close(fast, _solve(sde, 2, method, levy, d, m, stepwise=True, dtype=dtype), **tol)
    return fast


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("d,m", [(8, 3), (10, 3), (16, 4), (32, 8), (12, 16), (64, 5)])
@pytest.mark.parametrize("method,sde_type,levy", SCHEMES)
def test_the_references_additive_problem_is_one_launch(method, sde_type, levy, d, m, dtype):
    """ExAdditive: f = b / sqrt(1 + t) - y / (2 + 2 t), g = a b / sqrt(1 + t) repeated over the m columns. Every channel count
    path of the kernel: whole Philox quads (m = 4, 8, 16), single draws (m = 3, 5), four channels per lane and one (d = 10)."""
    sde = problems.AdditiveDecay(d, m, sde_type).to(DEV).to(dtype)
    _check(sde, method, levy, d, m, dtype)


@pytest.mark.parametrize("method,sde_type,levy", SCHEMES)
def test_a_constant_diffusion_matrix(method, sde_type, levy):
    """`sigma.expand(B, d, m)`: the table is the one matrix; the drift `-rate * y`."""
    sde = problems.AdditiveShared(16, 8, sde_type).to(DEV)
    _check(sde, method, levy, 16, 8)


class _NetOfTime(nn.Module):
    """The diffusion of the reference's NeuralAdditive (tests/problems.py:208-220): a network of t alone, reshaped to
    (B, d, m); an elementwise drift that uses t."""
    noise_type = "additive"

    def __init__(self, d, m, sde_type):
        super().__init__()
        self.d, self.m, self.sde_type = d, m, sde_type
        torch.manual_seed(3)
        self.g_net = nn.Sequential(nn.Linear(1, 8), nn.Softplus(), nn.Linear(8, d * m), nn.Sigmoid())
        self.rate = nn.Parameter(torch.rand(d) + 0.5)

    def f(self, t, y):
        return -self.rate * torch.tanh(y) * torch.cos(t)

    def g(self, t, y):
        return self.g_net(t.expand(y.size(0), 1)).view(y.size(0), self.d, self.m)


@pytest.mark.parametrize("method,sde_type,levy", SCHEMES)
def test_a_diffusion_that_is_a_network_of_t(method, sde_type, levy):
    sde = _NetOfTime(8, 4, sde_type).to(DEV)
    _check(sde, method, levy, 8, 4)


@pytest.mark.parametrize("d,m,hidden", [(8, 3, 8), (16, 4, 32), (12, 5, 8), (32, 8, 64), (64, 16, 40), (20, 2, 8), (64, 8, 128),
                                        (24, 4, 96), (3, 2, 8), (10, 3, 16), (37, 5, 24)])
@pytest.mark.parametrize("method,sde_type,levy", SCHEMES)
def test_the_references_neural_additive_problem_is_one_launch(method, sde_type, levy, d, m, hidden):
    """NeuralAdditive (tests/problems.py:195-224): f_net of cat([t, y]) on the matrix cores (`tsde_trajectory_mlp_additive`),
    g_net of t alone tabulated over the stage times and contracted with the increments by one more MFMA per four channels."""
    sde = problems.MLPNetAdditive(d, m, sde_type, hidden=hidden).to(DEV)
    _solve(sde, 1, method, levy, d, m)
    assert list(_book(sde)["trusted"].values()) == [True], _book(sde)
    fast, launches = _launches(lambda: _solve(sde, 2, method, levy, d, m))
    assert launches == 1
    torch.testing.assert_close(fast, _solve(sde, 2, method, levy, d, m, stepwise=True), rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("method,levy", [("euler", "none"), ("srk", "space-time")])
def test_neural_additive_rows_vs_oracle(method, levy):
    """16384 x 32 x 8, hidden 64, 500 steps: sampled rows against the oracle's restatement of the reference's loop on the same
    Brownian path (the bound of tests/test_gpu_full_size_oracle.py)."""
    import torchsde_amd
    from tests.test_gpu_full_size_oracle import _oracle_forward
    Bf, d, m, n, dt = 16384, 32, 8, 500, 2.0 ** -9
    sde = problems.MLPNetAdditive(d, m, "ito", hidden=64).to(DEV)
    y0 = torch.full((Bf, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, n * dt], device=DEV)

    def bm(entropy):
        return torchsde_amd.BrownianInterval(0.0, n * dt, size=(Bf, m), dtype=torch.float32, device=DEV, entropy=entropy, dt=dt,
                                             levy_area_approximation=levy)
    before = torch.get_num_threads()
    torch.set_num_threads(min(8, before))
    try:
        with torch.no_grad():
            torchsde_amd.sdeint(sde, y0, ts, bm=bm(5), method=method, dt=dt)
            ys, launches = _launches(lambda: torchsde_amd.sdeint(sde, y0, ts, bm=bm(20240601), method=method, dt=dt))
        assert launches == 1 and list(_book(sde)["trusted"].values()) == [True], _book(sde)
        rows = helpers.sampled_rows(Bf, 48, seed=8, seams=(2, 16, Bf - 2))
        ref32, ref64 = _oracle_forward(sde, rows, d, m, 20240601, n, dt, method, 0.1, levy=levy != "none")
        helpers.assert_within_reference_rounding(ys[-1][torch.from_numpy(rows).to(DEV)], ref32[-1], ref64[-1],
                                                 f"neural additive, {method}, matrix-core kernel")
    finally:
        torch.set_num_threads(before)


def test_parameters_are_read_at_every_solve_and_row_offsets_shard():
    """The table and the constants are this solve's live values (an optimiser step between solves is seen); two shards with
    row offsets reproduce the unsharded solve bit for bit (the increments are addressed by global row)."""
    sde = problems.AdditiveDecay(8, 4, "ito").to(DEV)
    a = _check(sde, "srk", "space-time", 8, 4)
    with torch.no_grad():
        sde.a.mul_(1.5)
        sde.b.add_(0.25)
    b, launches = _launches(lambda: _solve(sde, 2, "srk", "space-time", 8, 4))
    assert launches == 1 and not torch.equal(a, b)
    torch.testing.assert_close(b, _solve(sde, 2, "srk", "space-time", 8, 4, stepwise=True), rtol=2e-5, atol=2e-6)
    half = B // 2
    _solve(sde, 1, "srk", "space-time", 8, 4, rows=half)               # (trust is per batch size)
    lo = _solve(sde, 2, "srk", "space-time", 8, 4, rows=half)
    hi, launches = _launches(lambda: _solve(sde, 2, "srk", "space-time", 8, 4, rows=half, row_offset=half))
    assert launches == 1
    assert torch.equal(torch.cat([lo, hi], 1), b)


def test_what_stays_stepwise():
    """A diffusion computed from the state, one that reads t on the host, more than 16 channels: refused with the reason; the
    solve is the stepwise one."""
    import torchsde_amd

    class FromState(problems.AdditiveShared):
        def g(self, t, y):
            return (self.sigma * y.mean()).expand(y.size(0), -1, -1)

    class HostTime(problems.AdditiveShared):
        def g(self, t, y):
            return self.sigma.expand(y.size(0), -1, -1) * (1.0 if t > 0.5 else 2.0)

    for cls, m, reason in ((FromState, 4, "mean"), (HostTime, 4, "reads t on the host"), (problems.AdditiveShared, 32, "16")):
        sde = cls(8, m, "ito").to(DEV)
        out, launches = _launches(lambda: (_solve(sde, 1, "euler", "none", 8, m), _solve(sde, 2, "euler", "none", 8, m))[1])
        assert launches == 0                     # (no trajectory-kernel launch: the per-step kernels ran)
        assert any(reason in r for r in _book(sde)["refused"].values()), _book(sde)
        assert torch.equal(out, _solve(sde, 2, "euler", "none", 8, m, stepwise=True))
    assert any("stays stepwise" in line for line in torchsde_amd.recognise.describe(sde))


@pytest.mark.parametrize("method,levy", [("euler", "none"), ("srk", "space-time")])
def test_additive_rows_vs_oracle(method, levy):
    """16384 x 32 x 8, 500 steps of the reference's additive problem: sampled rows against the oracle's restatement of the
    reference's loop (euler.py:29-37; srk.py:90-111 with tableaus/sra1.py) on the same Brownian path, with the bound of
    tests/test_gpu_full_size_oracle.py."""
    import torchsde_amd
    from tests.test_gpu_full_size_oracle import _oracle_forward
    Bf, d, m, n, dt = 16384, 32, 8, 500, 2.0 ** -9
    sde = problems.AdditiveDecay(d, m, "ito").to(DEV)
    y0 = torch.full((Bf, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, n * dt], device=DEV)

    def bm(entropy):
        return torchsde_amd.BrownianInterval(0.0, n * dt, size=(Bf, m), dtype=torch.float32, device=DEV, entropy=entropy, dt=dt,
                                             levy_area_approximation=levy)
    before = torch.get_num_threads()
    torch.set_num_threads(min(8, before))
    try:
        with torch.no_grad():
            torchsde_amd.sdeint(sde, y0, ts, bm=bm(5), method=method, dt=dt)
            ys, launches = _launches(lambda: torchsde_amd.sdeint(sde, y0, ts, bm=bm(20240601), method=method, dt=dt))
        assert launches == 1 and list(_book(sde)["trusted"].values()) == [True], _book(sde)
        rows = helpers.sampled_rows(Bf, 48, seed=8, seams=(2, 8, Bf - 2))
        ref32, ref64 = _oracle_forward(sde, rows, d, m, 20240601, n, dt, method, 0.1, levy=levy != "none")
        helpers.assert_within_reference_rounding(ys[-1][torch.from_numpy(rows).to(DEV)], ref32[-1], ref64[-1],
                                                 f"additive noise, {method}, program + table kernel")
    finally:
        torch.set_num_threads(before)


def _golden_cases():
    import os
    return sorted(f[len("recognised_additive_"):-4] for f in os.listdir(helpers.GOLDEN) if f.startswith("recognised_additive_"))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("name", _golden_cases())
def test_additive_user_modules_on_the_trajectory_kernels_match_the_reference(name, dtype):
    """tests/golden/recognised_additive_*.npz: the REAL reference's `sdeint` of plain additive-noise modules (the shapes of
    its ExAdditive, a constant matrix, its NeuralAdditive) in float64 on the counter path (make_golden.py gen_additive). The
    verifying first solve (the stepwise result) and the kernel launches after it must both reproduce it. (The network kernel is
    float32 only: in float64 a network drift stays stepwise, which is checked too.)"""
    import torchsde_amd
    from tests.test_oracle_solvers import additive_module
    z = helpers.load(f"recognised_additive_{name}.npz")
    Bz, d, steps, m = (int(v) for v in z["shape"])
    dt, levy = float(z["dt"]), str(z["levy"])
    sde = additive_module(z, dtype).to(DEV)
    y0 = torch.tensor(z["y0"], dtype=dtype, device=DEV)
    ts = torch.tensor(z["ts"], dtype=dtype, device=DEV)
    want = torch.tensor(z["ys"])
    network = str(z["problem"]) == "MLPNetAdditive"
    tol = dict(rtol=1e-9, atol=1e-11) if dtype == torch.float64 else dict(rtol=2e-4, atol=2e-5)

    def solve():
        bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(Bz, m), dtype=dtype, device=DEV, entropy=int(z["entropy"]),
                                           dt=dt, levy_area_approximation=levy)
        with torch.no_grad():
            return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=str(z["method"]), dt=dt, options={"hip_graph": False})
    torch.testing.assert_close(solve().double().cpu(), want, **tol)                 # both routes, the stepwise one returned
    fast, launches = _launches(solve)
    torch.testing.assert_close(fast.double().cpu(), want, **tol)
    if network and dtype == torch.float64:
        assert launches == 0 and not _book(sde)["trusted"], _book(sde)
    else:
        assert launches == 1 and list(_book(sde)["trusted"].values()) == [True], _book(sde)
        assert [key[0][0][:2] for key in _book(sde)["trusted"]] == [("perceptron" if network else "program", "additive")]

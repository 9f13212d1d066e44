"""Neural SDEs -- drift AND diffusion two-layer perceptrons of (t, y), the reference's Neural* problems
(tests/problems.py:135-252) and BASELINE configs[2] -- on `tsde_trajectory_mlp_general` (``-m gpu``): an UNCHANGED user
module, no options; the first solve of a form runs both routes and returns the stepwise result, later ones are one launch.

Pinned three ways: against the stepwise route (which replays the reference's goldens), against the ORACLE's restatement of
the reference's loop on the same Brownian path at the full configs[2] size (sampled rows, the bound of
tests/test_gpu_full_size_oracle.py), and through the C ABI directly against a torch evaluation of the same networks."""
import numpy as np
import pytest
import torch
from torch import nn

from tests import helpers
from workloads import configs, problems

pytestmark = pytest.mark.gpu
DEV = "cuda"
DT = 2.0 ** -7


def _bm(B, m, t1, entropy, row_offset=0, levy="none"):
    import torchsde_amd
    return torchsde_amd.BrownianInterval(0.0, t1, size=(B, m), dtype=torch.float32, device=DEV, entropy=entropy, dt=DT,
                                         row_offset=row_offset, levy_area_approximation=levy)


def _solve(sde, m, entropy, method, B=96, d=None, steps=24, stepwise=False, ts=None, row_offset=0):
    import torchsde_amd
    d = sde.d if d is None else d
    y0 = torch.full((B, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, 7.5 * DT, steps * DT] if ts is None else ts, device=DEV)
    options = {"hip_graph": False}
    if stepwise:
        options["trajectory_kernel"] = False
    with torch.no_grad():
        levy = "space-time" if method in ("srk", None) else "none"
        return torchsde_amd.sdeint(sde, y0, ts, bm=_bm(B, m, float(ts[-1]), entropy, row_offset, levy), method=method, dt=DT,
                                   options=options)


def _book(sde):
    from torchsde_amd import solvers
    return getattr(sde, solvers.BaseSDESolver._RECOGNISED_ATTR, {"trusted": {}, "refused": {}})


def _launches(fn):
    from torchsde_amd import kernels as K
    K.prof_begin(8, 64)
    out = fn()
    torch.cuda.synchronize()
    return out, K.prof_end()[1]


GENERAL = [(8, 4, 8), (12, 4, 8), (8, 8, 8), (20, 8, 16), (32, 16, 64), (16, 16, 8), (8, 32, 24), (36, 8, 40),
           # any d <= 64 and any m <= 32: rows that are not 16-byte groups, channel counts between the tile widths
           (3, 5, 8), (10, 3, 16), (6, 2, 8), (17, 12, 24), (5, 20, 8), (1, 1, 8), (63, 7, 32)]


@pytest.mark.parametrize("d,m,hidden", GENERAL)
@pytest.mark.parametrize("method,sde_type", [("euler", "ito"), ("midpoint", "stratonovich")])
def test_general_noise_networks_take_the_matrix_core_kernel(d, m, hidden, method, sde_type):
    """NeuralGeneral (tests/problems.py:226-252): f_net, g_net of cat([t, y]), g reshaped to (B, d, m); an output time
    inside a step (interpolated in the kernel); every tile shape of the contraction (m = 4, 8, 16, 32; d not a multiple of
    16: padded tiles)."""
    sde = problems.MLPGeneral(d, m, sde_type, hidden=hidden).to(DEV)
    first = _solve(sde, m, 1, method)
    assert torch.equal(first, _solve(sde, m, 1, method, stepwise=True))             # the verifying solve returns the stepwise one
    assert list(_book(sde)["trusted"].values()) == [True], _book(sde)
    for entropy in (2, 3):
        fast, launches = _launches(lambda: _solve(sde, m, entropy, method))
        assert launches == 1
        slow = _solve(sde, m, entropy, method, stepwise=True)
        torch.testing.assert_close(fast, slow, rtol=2e-5, atol=2e-6)
        assert not torch.equal(fast[-1], fast[0])


@pytest.mark.parametrize("name,m_of", [("netdiag", lambda d: d), ("netscalar", lambda d: 1)])
@pytest.mark.parametrize("method,sde_type", [("euler", "ito"), ("midpoint", "stratonovich"), ("srk", "ito"), (None, "ito")])
def test_diagonal_and_scalar_noise_networks(name, m_of, method, sde_type):
    """NeuralDiagonal / NeuralScalar (tests/problems.py:135-192): 0.1 * sigmoid-closed g_net, (B, d) or (B, d, 1); Euler,
    midpoint, SRK (SRID2: three drift and four diffusion evaluations per step, srk.py:57-88) and the call with EVERY default
    (method None: `sdeint` picks SRK for diagonal and scalar Ito noise, sdeint.py:246-253)."""
    # (hidden above 64: the 128-unit instantiations; d = 3, 10, 37: rows that are not 16-byte groups)
    for d, hidden in ((8, 8), (20, 24), (64, 64), (16, 128), (64, 100), (3, 8), (10, 16), (37, 24)):
        sde = problems.make(f"{name}_{'ito' if sde_type == 'ito' else 'strat'}", d=d, hidden=hidden).to(DEV)
        m = m_of(d)
        _solve(sde, m, 1, method, d=d)
        assert list(_book(sde)["trusted"].values()) == [True], _book(sde)
        fast, launches = _launches(lambda: _solve(sde, m, 2, method, d=d))
        assert launches == 1
        torch.testing.assert_close(fast, _solve(sde, m, 2, method, d=d, stepwise=True), rtol=2e-5, atol=2e-6)


def test_rows_are_global_and_a_partial_last_group_is_handled():
    """The increments are those of GLOBAL rows (sharding-invariant), and a batch that is not a multiple of 16 works."""
    sde = problems.MLPGeneral(8, 8, "ito", hidden=8).to(DEV)
    whole = [_solve(sde, 8, 5, "euler", B=200) for _ in range(2)][1]                # (the second solve: the kernel)
    for lo, hi in ((0, 104), (104, 200)):
        part = [_solve(sde, 8, 5, "euler", B=hi - lo, row_offset=lo) for _ in range(2)][1]
        assert torch.equal(part, whole[:, lo:hi])
    odd = [_solve(sde, 8, 5, "euler", B=77) for _ in range(2)][1]
    assert torch.equal(odd, whole[:, :77])


@pytest.mark.parametrize("B,d", [(9, 3), (11, 10), (13, 5), (9, 63)])
def test_rows_times_width_not_a_multiple_of_four_and_offset_views_of_the_start(B, d):
    """`ys[1:]` starts rows * d * 4 bytes into its allocation, so a batch with rows * d % 4 != 0 handed the
    C entry points an output pointer that is not 16-byte aligned and a plain `sdeint` of a Neural* module raised. Such rows go
    element by element (the entry points ask for 16-byte alignment only when d % 4 == 0); a y0 that is an offset view of a
    larger tensor is copied to an aligned buffer."""
    import torchsde_amd
    for name, m in (("netdiag_ito", d), ("netscalar_ito", 1)):
        sde = problems.make(name, d=d, hidden=8).to(DEV)
        _solve(sde, m, 1, "euler", B=B, d=d)
        assert list(_book(sde)["trusted"].values()) == [True], _book(sde)
        fast, launches = _launches(lambda: _solve(sde, m, 2, "euler", B=B, d=d))
        assert launches == 1
        torch.testing.assert_close(fast, _solve(sde, m, 2, "euler", B=B, d=d, stepwise=True), rtol=2e-5, atol=2e-6)
    gen = problems.MLPGeneral(d, 3, "ito", hidden=8).to(DEV)
    add = problems.make("netadditive_ito", d=d, m=3, hidden=8).to(DEV) if d <= 16 else None
    for sde in (gen, add):
        if sde is None:
            continue
        _solve(sde, 3, 1, "euler", B=B, d=d)
        fast, launches = _launches(lambda: _solve(sde, 3, 2, "euler", B=B, d=d))
        assert launches == 1, _book(sde)
        torch.testing.assert_close(fast, _solve(sde, 3, 2, "euler", B=B, d=d, stepwise=True), rtol=2e-5, atol=2e-6)
    # an offset view as the start (d a multiple of 4, the view 4 bytes off a 16-byte boundary)
    sde = problems.make("netdiag_ito", d=8, hidden=8).to(DEV)
    big = torch.full((1 + 16 * 8,), 0.1, device=DEV)
    y0 = big[1:].view(16, 8)
    assert y0.data_ptr() % 16 != 0
    ts = torch.tensor([0.0, 24 * DT], device=DEV)
    outs = []
    for entropy in (1, 2, 2):
        with torch.no_grad():
            outs.append(torchsde_amd.sdeint(sde, y0, ts, bm=_bm(16, 8, 24 * DT, entropy), method="euler", dt=DT,
                                            options={"hip_graph": False, "trajectory_kernel": len(outs) < 2}))
    torch.testing.assert_close(outs[1], outs[2], rtol=2e-5, atol=2e-6)


def test_live_parameters_and_what_stays_stepwise():
    sde = problems.MLPGeneral(8, 4, "ito", hidden=8).to(DEV)
    _solve(sde, 4, 1, "euler")
    a = _solve(sde, 4, 2, "euler")
    with torch.no_grad():
        sde.g_net[2].bias.add_(0.5)                                               # an optimiser step
    b, launches = _launches(lambda: _solve(sde, 4, 2, "euler"))
    assert launches == 1 and not torch.equal(a, b)
    torch.testing.assert_close(b, _solve(sde, 4, 2, "euler", stepwise=True), rtol=2e-5, atol=2e-6)
    # schemes the kernel does not have, and states wider than 64 channels, keep the stepwise route
    wide = problems.MLPGeneral(68, 4, "ito", hidden=8).to(DEV)
    for _ in range(2):
        got, launches = _launches(lambda: _solve(wide, 4, 3, "euler", d=68))
        assert launches == 0
    # (Heun and Euler-Heun: not this kernel's -- since round 6 the deep-network kernel's, tests/test_gpu_neural_rheun_route.py)
    strat = problems.MLPGeneral(8, 4, "stratonovich", hidden=8).to(DEV)
    for _ in range(2):
        got, launches = _launches(lambda: _solve(strat, 4, 3, "heun"))
        assert launches == 0
    torch.testing.assert_close(got, _solve(strat, 4, 3, "heun", stepwise=True), rtol=2e-5, atol=2e-6)


def test_c_abi_against_a_torch_evaluation_of_the_same_networks():
    """tsde_trajectory_mlp_general called directly (no module, no interpretation): tanh nets without a time input, an
    identity-closed diffusion with a scale, Euler, materialised increments of the same generator."""
    import torchsde_amd
    from torchsde_amd import _native, kernels as K
    torch.manual_seed(0)
    B, d, m, hf, hg, steps = 80, 12, 8, 16, 24, 9
    mk = lambda *shape: (0.4 * torch.randn(*shape, device=DEV)).contiguous()      # noqa: E731
    fnet = K.NeuralNet(mk(d, hf), None, mk(hf), mk(hf, d), mk(d), _native.ACT_TANH)
    gnet = K.NeuralNet(mk(d, hg), mk(hg), mk(hg), mk(hg, d * m), mk(d * m), _native.ACT_TANH, _native.FINAL_NONE, 0.3)
    bm = _bm(B, m, steps * DT, 11)
    grid = np.arange(steps + 1) * DT
    bm.adopt_grid(grid)
    cells = np.asarray(bm.match_grid(grid), dtype=np.int64)
    rows = np.zeros((steps, 8))
    rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3] = DT, DT / 2, 1 / DT, np.sqrt(DT)
    rows[:, 4], rows[:, 5], rows[:, 6], rows[:, 7] = np.sqrt(DT), np.sqrt(DT / 12), DT, grid[:-1]
    schedule = K.TrajectorySchedule(rows, cells, [steps], [(0.0, 1.0)], torch.device(DEV), torch.float32)
    y0 = torch.full((B, d), 0.1, device=DEV)
    ys = torch.empty(1, B, d, device=DEV)
    K.trajectory_mlp_general(ys, y0, fnet, gnet, _native.NOISE_GENERAL, m, _native.TRAJ_EULER, schedule, bm)
    y = y0.double()
    W = lambda t: t.double()                                                       # noqa: E731
    for k in range(steps):
        t = grid[k]
        dW = bm(float(grid[k]), float(grid[k + 1])).double()
        f = torch.tanh(y @ W(fnet.tensors[0]) + W(fnet.tensors[2])) @ W(fnet.tensors[3]) + W(fnet.tensors[4])
        hid = torch.tanh(y @ W(gnet.tensors[0]) + W(gnet.tensors[2]) + W(gnet.tensors[1]) * t)
        g = (0.3 * (hid @ W(gnet.tensors[3]) + W(gnet.tensors[4]))).reshape(B, d, m)
        y = y + f * DT + torch.bmm(g, dW.unsqueeze(-1)).squeeze(-1)
    torch.testing.assert_close(ys[0].double(), y, rtol=2e-5, atol=2e-6)


def test_c3_full_size_default_route_rows_vs_oracle():
    """BASELINE configs[2] (16384 x 32 x 16, 1000 steps, hidden 64) as a drop-in call: the untouched NeuralGeneral-style
    module through `sdeint` with no options is ONE launch, and its sampled rows agree with the oracle's restatement of the
    reference's loop (euler.py:29-37 with misc.batch_mvp) on the same Brownian path."""
    import torchsde_amd
    from tests.test_gpu_full_size_oracle import _oracle_forward
    c = configs.WORKLOADS["c3_euler_general_b16384_d32_m16"]
    B, d, m, n, dt = c["B"], c["d"], c["m"], c["nsteps"], c["dt"]
    sde = configs.make_problem(c["problem"], d, m, DEV)
    y0 = torch.full((B, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, n * dt], device=DEV)

    def bm(entropy):
        return torchsde_amd.BrownianInterval(0.0, n * dt, size=(B, m), dtype=torch.float32, device=DEV, entropy=entropy, dt=dt)
    before = torch.get_num_threads()
    torch.set_num_threads(min(8, before))
    try:
        with torch.no_grad():
            first = torchsde_amd.sdeint(sde, y0, ts, bm=bm(20240601), method="euler", dt=dt)     # both routes, compared
            ys, launches = _launches(lambda: torchsde_amd.sdeint(sde, y0, ts, bm=bm(20240601), method="euler", dt=dt))
        assert launches == 1 and list(_book(sde)["trusted"].values()) == [True], _book(sde)
        torch.testing.assert_close(ys, first, rtol=1e-4, atol=1e-5)
        rows = helpers.sampled_rows(B, 64, seed=3, seams=(16, 64, B - 16))
        ref32, ref64 = _oracle_forward(sde, rows, d, m, 20240601, n, dt, "euler", 0.1)
        helpers.assert_within_reference_rounding(ys[-1][torch.from_numpy(rows).to(DEV)], ref32[-1], ref64[-1],
                                                 "C3 Euler-general final state, neural-SDE kernel")
    finally:
        torch.set_num_threads(before)


@pytest.mark.parametrize("name,noise,method,d,m", [("netdiag_ito", "diagonal", "euler", 32, 32),
                                                   ("netscalar_ito", "scalar", "euler", 32, 1),
                                                   ("netdiag_ito", "diagonal", "srk", 32, 32),
                                                   ("netscalar_ito", "scalar", "srk", 16, 1),
                                                   ("general_strat", "general", "midpoint", 16, 8)])
def test_neural_kernel_rows_vs_oracle(name, noise, method, d, m):
    """The other modes of the neural-SDE kernel -- diagonal and scalar noise (the reference's NeuralDiagonal, NeuralScalar),
    and the midpoint scheme on general noise -- against the ORACLE's restatement of the reference's loops (euler.py:29-37,
    midpoint.py:29-45) on the same Brownian path: 4096 rows, 256 steps, sampled rows, the bound of
    tests/test_gpu_full_size_oracle.py."""
    import torchsde_amd
    from tests.test_gpu_full_size_oracle import _oracle_forward
    Bf, n, dt = 4096, 256, 2.0 ** -8
    sde = (problems.make(name, d=d, m=m) if noise == "general" else problems.make(name, d=d, hidden=16)).to(DEV)
    y0 = torch.full((Bf, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, n * dt], device=DEV)

    levy = "space-time" if method == "srk" else "none"

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
        rows = helpers.sampled_rows(Bf, 48, seed=11, seams=(16, Bf - 16))
        ref32, ref64 = _oracle_forward(sde, rows, d, m, 20240601, n, dt, method, 0.1, levy=levy != "none")
        helpers.assert_within_reference_rounding(ys[-1][torch.from_numpy(rows).to(DEV)], ref32[-1], ref64[-1],
                                                 f"{name}, {method}, neural-SDE kernel")
    finally:
        torch.set_num_threads(before)


def test_opt_in_split_bf16_products_error_against_float64():
    """`options={"matrix_precision": "bf16x3"}` (an opt-in experiment, never the default): the diffusion
    net's second layer on v_mfma_f32_16x16x32_bf16 with heads and tails of both operands. Its error against a float64
    evaluation of the same recursion on the same increments is reported next to the exact-f32 mode's and bounded (measured
    on MI355X: 6.9e-7 exact, 1.7e-6 split, relative to max |y|; the dropped lo*lo term is ~2^-16 of a product)."""
    import torchsde_amd
    Bf, d, m, hidden, n, dt = 2048, 32, 16, 64, 256, 2.0 ** -8
    sde = problems.MLPGeneral(d, m, "ito", hidden=hidden).to(DEV)
    y0 = torch.full((Bf, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, n * dt], device=DEV)

    def bm():
        return torchsde_amd.BrownianInterval(0.0, n * dt, size=(Bf, m), dtype=torch.float32, device=DEV, entropy=77, dt=dt)

    def solve(options):
        with torch.no_grad():
            for _ in range(2):                       # (the first solve of a mode earns its trust and returns the stepwise result)
                out, launches = _launches(lambda: torchsde_amd.sdeint(sde, y0, ts, bm=bm(), method="euler", dt=dt, options=options))
        assert launches == 1
        return out[-1]
    exact, split = solve(None), solve({"matrix_precision": "bf16x3"})
    assert sorted(str(k[-1]) for k in _book(sde)["trusted"]) == [str(Bf), "bf16x3"] and all(
        v is True for v in _book(sde)["trusted"].values()), _book(sde)
    # float64 reference: the same Euler recursion, the same increments (materialised from the same generator)
    sde64 = problems.MLPGeneral(d, m, "ito", hidden=hidden).to(DEV).double()
    sde64.load_state_dict({k: v.double() for k, v in sde.state_dict().items()})
    path, y = bm(), y0.double()
    with torch.no_grad():
        for k in range(n):
            t = torch.tensor(k * dt, device=DEV, dtype=torch.float64)
            dW = path(k * dt, (k + 1) * dt).double()
            y = y + sde64.f(t, y) * dt + torch.bmm(sde64.g(t, y), dW.unsqueeze(-1)).squeeze(-1)
    scale = y.abs().max().item()
    err_exact = (exact.double() - y).abs().max().item() / scale
    err_split = (split.double() - y).abs().max().item() / scale
    print(f"max |y - y64| / max |y64|: exact f32 {err_exact:.2e}, split bf16 x3 {err_split:.2e}")
    assert err_exact < 5e-6 and err_split < 2e-5 and not torch.equal(exact, split)      # measured: 6.9e-7 and 1.7e-6
    with pytest.raises(ValueError, match="matrix_precision"):
        torchsde_amd.sdeint(sde, y0, ts, bm=bm(), method="euler", dt=dt, options={"matrix_precision": "fp8"})

"""Adaptive stepping with accept / reject decided on the device (``-m gpu``; torchsde_amd/adaptive.py,
csrc/adaptive.hip) against the host-driven form of the same loop (`options={"device_adaptive": False}`: one sync per
attempt), which tests/test_gpu_parity.py pins to the oracle's restatement of the reference's adaptive branch
(base_solver.py:117-142, adaptive_stepping.py:21-76; golden: tests/golden/adaptive_*.npz).

Both forms run the same kernels on the same Brownian path; the only arithmetic that differs is the controller's two
`pow` calls (device libm vs CPython), i.e. the proposed step sizes agree to an ulp of a double and the solutions to
rounding."""
import pytest
import torch

from workloads import problems

pytestmark = pytest.mark.gpu
DEV = "cuda"

CASES = [
    # problem, method, levy, (B, d, m), rtol, atol
    ("gbm_ito", "milstein", "none", (64, 8, 8), 1e-3, 1e-3),
    ("gbm_ito", "srk", "space-time", (64, 8, 8), 1e-4, 1e-4),
    ("gbm_ito", "euler", "none", (64, 8, 8), 1e-2, 1e-2),
    ("gbm_strat", "midpoint", "none", (64, 8, 8), 1e-3, 1e-3),
    ("gbm_strat", "heun", "none", (48, 4, 4), 1e-3, 1e-3),
    ("gbm_strat", "euler_heun", "none", (48, 4, 4), 1e-3, 1e-3),
    ("gbm_strat", "milstein", "none", (48, 4, 4), 1e-3, 1e-3),
    ("additive_ito", "euler", "none", (48, 4, 3), 1e-3, 1e-3),
    ("additive_ito", "srk", "space-time", (48, 4, 3), 1e-4, 1e-4),
    ("scalar_ito", "milstein", "none", (48, 4, 1), 1e-3, 1e-3),
    ("scalar_ito", "srk", "space-time", (48, 4, 1), 1e-3, 1e-3),
    ("general_ito", "euler", "none", (48, 4, 4), 1e-2, 1e-2),
    ("general_strat", "midpoint", "none", (48, 4, 4), 1e-3, 1e-3),
    ("mlpdiag_ito", "milstein", "none", (48, 4, 4), 1e-3, 1e-3),
]


def _solve(prob, method, levy, shape, rtol, atol, dtype, device_control, ts_list=(0.0, 0.3, 0.35, 1.0), options=None):
    import torchsde_amd
    from torchsde_amd import adaptive
    B, d, m = shape
    sde = problems.make(prob, dtype=dtype, d=d, m=m).to(DEV)
    y0 = torch.full((B, d), 0.1, dtype=dtype, device=DEV)
    ts = torch.tensor(ts_list, dtype=dtype, device=DEV)
    bm = torchsde_amd.BrownianInterval(ts_list[0], ts_list[-1], size=(B, m), dtype=dtype, device=DEV, entropy=99,
                                       levy_area_approximation=levy)
    adaptive.last_stats = None
    opts = dict(options or {}, device_adaptive=device_control)
    with torch.no_grad():
        ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=0.1, adaptive=True, rtol=rtol, atol=atol,
                                 options=opts)
    return ys, adaptive.last_stats


def _fine_fixed_step(prob, method, levy, shape, dtype, ts_list=(0.0, 0.3, 0.35, 1.0)):
    import torchsde_amd
    B, d, m = shape
    sde = problems.make(prob, dtype=dtype, d=d, m=m).to(DEV)
    y0 = torch.full((B, d), 0.1, dtype=dtype, device=DEV)
    ts = torch.tensor(ts_list, dtype=dtype, device=DEV)
    bm = torchsde_amd.BrownianInterval(ts_list[0], ts_list[-1], size=(B, m), dtype=dtype, device=DEV, entropy=99,
                                       levy_area_approximation=levy)
    with torch.no_grad():
        return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=2.0 ** -11)


@pytest.mark.filterwarnings("ignore:Numerical solution is not guaranteed")
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("prob,method,levy,shape,rtol,atol", CASES)
def test_device_control_equals_host_control(prob, method, levy, shape, rtol, atol, dtype):
    on_device, stats = _solve(prob, method, levy, shape, rtol, atol, dtype, True)
    on_host, none = _solve(prob, method, levy, shape, rtol, atol, dtype, False)
    assert none is None and stats is not None and stats["control"] == "device"
    assert torch.isfinite(on_device).all() and on_device.shape == on_host.shape
    if dtype == torch.float32:
        # times are float32: the two controllers' step sizes (doubles that agree to an ulp) round to the same times,
        # the same increments are queried and the solutions agree to rounding
        torch.testing.assert_close(on_device, on_host, rtol=2e-5, atol=2e-6)
    else:
        # In float64 an ulp of difference in a proposed step size moves the query times by 1e-17; the increments
        # follow (a Brownian path is nowhere smooth), the error estimate feeds that back into the next step size, and
        # after a few attempts the two runs are on different -- equally valid -- time grids. What must hold is that both
        # are solutions of the same accuracy on the SAME path: compare each with a fine fixed-step solve of that path.
        fine = _fine_fixed_step(prob, method, levy, shape, dtype)
        err_dev = (on_device - fine).pow(2).mean().sqrt().item()
        err_host = (on_host - fine).pow(2).mean().sqrt().item()
        scale = fine.abs().max().item()
        assert err_dev <= 3.0 * err_host + 1e-3 * scale, (err_dev, err_host, scale)
        assert err_host <= 3.0 * err_dev + 1e-3 * scale, (err_dev, err_host, scale)
    # one synchronisation per round of attempts: a few per output time, never one per attempt
    assert stats["host_syncs"] <= 3 * stats["output_times"] + 1, stats
    assert stats["accepted"] <= stats["attempts_used"] <= stats["attempts_enqueued"]


def test_c2_size_adaptive_solve_syncs_once_per_output_time():
    """The headline shape (65536 x 64, GBM), adaptive Milstein over [0, 1] with 4 output times: the number of host
    synchronisations is of the order of the output times, not of the attempted steps."""
    ys, stats = _solve("gbm_ito", "milstein", "none", (65536, 64, 64), 1e-3, 1e-4, torch.float32, True,
                       ts_list=(0.0, 0.25, 0.5, 0.75, 1.0))
    assert torch.isfinite(ys).all() and stats["output_times"] == 4
    assert stats["attempts_used"] >= 12 and stats["host_syncs"] <= 3 * stats["output_times"], stats
    assert stats["host_syncs"] < stats["attempts_used"] / 2 and stats["attempts_enqueued"] <= stats["attempts_used"] + 4
    host, _ = _solve("gbm_ito", "milstein", "none", (65536, 64, 64), 1e-3, 1e-4, torch.float32, False,
                     ts_list=(0.0, 0.25, 0.5, 0.75, 1.0))
    torch.testing.assert_close(ys, host, rtol=2e-5, atol=2e-6)


def test_two_output_times_inside_one_step_and_dt_min():
    """Output times closer together than a step (the second needs no further step: interpolation only), and a tolerance
    so tight that the controller runs into dt_min and warns like the reference (base_solver.py:134-137)."""
    ys, stats = _solve("gbm_ito", "milstein", "none", (32, 4, 4), 1e-2, 1e-2, torch.float64, True,
                       ts_list=(0.0, 0.01, 0.02, 0.5))
    host, _ = _solve("gbm_ito", "milstein", "none", (32, 4, 4), 1e-2, 1e-2, torch.float64, False,
                     ts_list=(0.0, 0.01, 0.02, 0.5))
    torch.testing.assert_close(ys, host, rtol=1e-2, atol=1e-3)       # float64: see the comment in the test above
    import torchsde_amd
    sde = problems.make("gbm_ito", dtype=torch.float64, d=4).to(DEV)
    y0 = torch.full((16, 4), 0.1, dtype=torch.float64, device=DEV)
    ts = torch.tensor([0.0, 0.05], dtype=torch.float64, device=DEV)

    def run(device_control):
        bm = torchsde_amd.BrownianInterval(0.0, 0.05, size=(16, 4), dtype=torch.float64, device=DEV, entropy=3)
        with torch.no_grad():
            return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="euler", dt=0.01, adaptive=True, rtol=1e-12,
                                       atol=1e-12, dt_min=2e-3, options={"device_adaptive": device_control})
    with pytest.warns(UserWarning, match="Hitting minimum allowed step size"):
        a = run(True)
    with pytest.warns(UserWarning, match="Hitting minimum allowed step size"):
        b = run(False)
    torch.testing.assert_close(a, b, rtol=1e-9, atol=1e-11)      # pinned at dt_min: both take the same steps




@pytest.mark.parametrize("method,levy", [("milstein", "none"), ("srk", "space-time"), ("euler", "none")])
def test_attempt_replayed_as_a_cached_graph_is_bit_identical(method, levy):
    """``options={"hip_graph": True}`` on a device-controlled adaptive solve: the attempt is captured once per SDE object
    and structure and replayed by later solves (other entropy, other y0, other output times and tolerances of the
    controller's state) -- the same launches, so the same bits as the eagerly issued solve."""
    import warnings
    import torchsde_amd
    from torchsde_amd import adaptive
    B, d = 96, 8
    sde = problems.make("gbm_ito", d=d).to(DEV)

    def solve(entropy, y_value, ts, graph):
        bm = torchsde_amd.BrownianInterval(0.0, 1.0, size=(B, d), device=DEV, dtype=torch.float32, entropy=entropy,
                                           levy_area_approximation=levy)
        y0 = torch.full((B, d), y_value, device=DEV)
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ys = torchsde_amd.sdeint(sde, y0, torch.tensor(ts, device=DEV), bm=bm, method=method, dt=0.05, adaptive=True,
                                     rtol=1e-3, atol=1e-4, options={"hip_graph": graph})
        return ys, dict(adaptive.last_stats)

    for k, (entropy, y_value, ts) in enumerate([(5, 0.1, [0.0, 0.5, 1.0]), (6, 0.3, [0.0, 0.25, 0.5, 1.0]),
                                                (7, 0.2, [0.0, 1.0])]):
        eager, stats_eager = solve(entropy, y_value, ts, False)
        graphed, stats_graph = solve(entropy, y_value, ts, True)
        assert stats_eager["launch"] == "eager" and stats_graph["launch"] == "graph replay"
        assert stats_graph["attempts_used"] == stats_eager["attempts_used"] > 3
        assert torch.equal(eager, graphed), (k, (eager - graphed).abs().max().item())
    from torchsde_amd import graph as graph_module
    cached = [key for key in graph_module._cache_of(sde) if key[0] == "adaptive-attempt"]
    assert len(cached) == 1                                     # three solves, one capture


def _solve_on(sde, ts_list, entropy, device_control, shape=(48, 4, 4), method="milstein", dt=0.05, rtol=1e-3, atol=1e-4):
    import torchsde_amd
    from torchsde_amd import adaptive
    B, d, m = shape
    y0 = torch.full((B, d), 0.1, device=DEV)
    bm = torchsde_amd.BrownianInterval(ts_list[0], ts_list[-1], size=(B, m), dtype=torch.float32, device=DEV, entropy=entropy)
    adaptive.last_stats = None
    with torch.no_grad():
        ys = torchsde_amd.sdeint(sde, y0, torch.tensor(ts_list, device=DEV), bm=bm, method=method, dt=dt, adaptive=True,
                                 rtol=rtol, atol=atol, options={"device_adaptive": device_control, "hip_graph": False})
    return ys, (dict(adaptive.last_stats) if adaptive.last_stats else None)


def test_output_times_on_the_device_one_or_two_synchronisations_per_solve():
    """The controller walks the list of output times and the emit kernel writes their rows (base_solver.py:117-145, both
    loops, on the device): a solve synchronises to learn that it is complete, not once per output time. First solve of an
    SDE object: the budget is the upper bound the initial step size gives (rejections may cost a second round); later
    solves: the attempts the previous one used."""
    ts_list = [0.0, 0.1, 0.25, 0.3, 0.5, 0.75, 0.8, 1.0]
    sde = problems.make("gbm_ito", d=4).to(DEV)
    first, stats1 = _solve_on(sde, ts_list, 11, True)
    again, stats2 = _solve_on(sde, ts_list, 11, True)
    other, stats3 = _solve_on(sde, ts_list, 12, True)
    host, _ = _solve_on(sde, ts_list, 11, False)
    host_other, _ = _solve_on(sde, ts_list, 12, False)
    assert stats1["output_times"] == 7 and stats1["host_syncs"] <= 3, stats1
    assert stats2["host_syncs"] == 1 and stats2["attempts_enqueued"] == stats2["attempts_used"] + 1, stats2
    assert stats3["host_syncs"] <= 2, stats3
    assert torch.equal(first, again)
    torch.testing.assert_close(first, host, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(other, host_other, rtol=2e-5, atol=2e-6)


def test_many_output_times_inside_single_steps():
    """Forty output times over a handful of steps (several rows written by one emit launch): the rows are the host loop's."""
    sde = problems.make("gbm_ito", d=4).to(DEV)
    dense = [0.0] + [round(0.025 * k, 6) for k in range(1, 40)] + [1.0]
    a, stats = _solve_on(sde, dense, 5, True, dt=0.2, rtol=1e-2, atol=1e-2)
    b, _ = _solve_on(sde, dense, 5, False, dt=0.2, rtol=1e-2, atol=1e-2)
    assert stats["output_times"] == len(dense) - 1 and stats["host_syncs"] <= 3 and stats["accepted"] < 30, stats
    torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-6)


@pytest.mark.filterwarnings("ignore:Numerical solution is not guaranteed")
@pytest.mark.parametrize("prob,method,levy,shape", [("gbm_ito", "milstein", "none", (48, 4, 4)),
                                                    ("gbm_ito", "srk", "space-time", (48, 4, 4)),
                                                    ("mlpdiag_ito", "euler", "none", (48, 4, 4)),
                                                    ("additive_ito", "srk", "space-time", (48, 4, 3)),
                                                    ("general_ito", "euler", "none", (48, 4, 4))])
def test_adaptive_solve_with_gradients_replays_the_accepted_steps(prob, method, levy, shape):
    """Autograd recording, `options={"adaptive_replay": True}`: the device-controlled loop finds the accepted steps under
    no_grad, then those alone are run again with autograd on (adaptive.integrate_with_grad) -- values and gradients (y0 and every parameter) equal the host-driven
    loop's, which records every attempt like the reference (base_solver.py:117-142); one synchronisation per round instead of
    one per attempt."""
    import torchsde_amd
    from torchsde_amd import adaptive
    B, d, m = shape
    sde = problems.make(prob, d=d, m=m).to(DEV)
    ts = torch.tensor([0.0, 0.3, 0.35, 1.0], device=DEV)
    weights = None

    def run(device_control):
        nonlocal weights
        y0 = torch.full((B, d), 0.1, device=DEV, requires_grad=True)
        bm = torchsde_amd.BrownianInterval(0.0, 1.0, size=(B, m), dtype=torch.float32, device=DEV, entropy=21,
                                           levy_area_approximation=levy)
        sde.zero_grad()
        adaptive.last_stats = None
        ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=0.1, adaptive=True, rtol=1e-3, atol=1e-3,
                                 options={"device_adaptive": device_control, "hip_graph": False, "adaptive_replay": True})
        if weights is None:
            weights = torch.cos(torch.arange(ys.numel(), device=DEV, dtype=torch.float32)).reshape(ys.shape)
        (ys * weights).sum().backward()
        return ys.detach(), y0.grad.clone(), {n: p.grad.clone() for n, p in sde.named_parameters()}, adaptive.last_stats

    ys_d, gy_d, gp_d, stats = run(True)
    ys_h, gy_h, gp_h, none = run(False)
    assert none is None and stats is not None and "replayed" in stats["control"], stats
    assert stats["replayed_steps"] == stats["accepted"] >= 3 and stats["host_syncs"] <= 3, stats
    # the replayed steps ARE the device-controlled solve: same kernels, same increments
    with torch.no_grad():
        y0 = torch.full((B, d), 0.1, device=DEV)
        bm = torchsde_amd.BrownianInterval(0.0, 1.0, size=(B, m), dtype=torch.float32, device=DEV, entropy=21,
                                           levy_area_approximation=levy)
        plain = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=0.1, adaptive=True, rtol=1e-3, atol=1e-3,
                                    options={"hip_graph": False})
    torch.testing.assert_close(ys_d, plain, rtol=2e-6, atol=2e-6)
    assert gp_h and set(gp_d) == set(gp_h)
    same_grid = torch.allclose(ys_d, ys_h, rtol=2e-5, atol=2e-6)
    # The host loop's step sizes agree with the controller kernel's to an ulp of a double; where that rounds to another
    # float32 time the two runs continue on different -- equally valid -- grids (profiles/r5_adaptive_one_sync.txt) and
    # values and gradients agree to the accuracy of the solve instead of to rounding.
    tol_y = dict(rtol=2e-5, atol=2e-6) if same_grid else dict(rtol=5e-3, atol=5e-4)
    torch.testing.assert_close(ys_d, ys_h, **tol_y)
    scale = max(gy_h.abs().max().item(), 1e-6)
    torch.testing.assert_close(gy_d, gy_h, rtol=2e-4 if same_grid else 2e-2, atol=(2e-5 if same_grid else 2e-2) * scale)
    for key in gp_h:
        scale = max(gp_h[key].abs().max().item(), 1e-6)
        torch.testing.assert_close(gp_d[key], gp_h[key], rtol=5e-4 if same_grid else 2e-2,
                                   atol=(5e-5 if same_grid else 2e-2) * scale)

"""CPU tests of everything that is not a kernel: the C ABI surface, the contract checks, the time grid, the
Brownian cell bookkeeping and the loud-failure rules. (No GPU compute is attempted.)"""
import ctypes
import os
import re
import warnings

import numpy as np
import pytest
import torch

import torchsde_amd
from tests import helpers
from workloads import problems
from torchsde_amd import _native, contract, solvers, timegrid
from torchsde_amd.brownian import BrownianInterval, uniform_edges
from torchsde_amd.sde import ForwardSDE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- C ABI ---------------------------------------------------------------------------------------------------
def _header_symbols():
    text = open(os.path.join(ROOT, "include", "torchsde_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tsde_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_native.LIB_PATH)
    names = _header_symbols()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/torchsde_amd.h but not exported"


def test_python_binding_covers_header():
    assert sorted(_native.SIGNATURES) == _header_symbols()
    assert _native.load().tsde_abi_version() == 2


def test_noise_struct_layout_matches_header():
    assert ctypes.sizeof(_native.Noise) == 64 and _native.Noise.entropy_dev.offset == 56
    assert _native.Noise.h.offset == 40 and _native.Noise.bcast_d.offset == 48
    assert ctypes.sizeof(_native.Seg) == 72


def test_cpu_tensors_fail_loudly():
    sde = problems.make("gbm_ito", d=4)
    with pytest.raises(_native.NativeLibraryError, match="no CPU fallback"):
        torchsde_amd.sdeint(sde, torch.full((3, 4), 0.1), torch.tensor([0.0, 0.1]), method="euler", dt=0.05)
    bm = BrownianInterval(0.0, 1.0, size=(3, 4))
    with pytest.raises(_native.NativeLibraryError):
        bm(0.0, 0.5)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", "/nonexistent/libtorchsde_amd.so")
    with pytest.raises(_native.NativeLibraryError, match="not built"):
        _native.load()


# ---- time grid --------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["f32_1e-3", "f32_dyadic", "f32_multi", "f64_multi", "f32_linspace20", "f64_1e-2",
                                  "f32_coarse"])
def test_time_grid_matches_reference_queries(name):
    """The host-side grid equals, query for query, what the reference's stepping loop asks its Brownian motion."""
    z = helpers.load("timegrid.npz")
    grid = timegrid.build(z[name + "__ts"], float(z[name + "__dt"]))
    t = grid.t_f64()
    assert np.array_equal(np.stack([t[:-1], t[1:]], axis=1), z[name + "__queries"])


def test_time_grid_fp32_remainder_and_interpolation():
    grid = timegrid.build(np.array([0.0, 1.0], dtype=np.float32), 1e-3)
    assert grid.n_steps == 1001                       # 1000 steps + a 9.3e-6 remainder
    assert grid.outputs[-1][1] == 1001 and grid.outputs[-1][2:] == (0.0, 1.0)
    grid = timegrid.build(np.array([0.0, 0.25, 0.5], dtype=np.float64), 0.1)
    assert grid.n_steps == 5                          # no clipping at 0.25: interpolated from the (0.2, 0.3) step
    kp, kc, w0, w1 = grid.outputs[0]
    assert (kp, kc) == (2, 3) and abs(w0 - 0.5) < 1e-12 and abs(w1 - 0.5) < 1e-12


# ---- contract ------------------------------------------------------------------------------------------------
class _NoNoiseType:
    sde_type = "ito"

    def f(self, t, y):
        return y

    def g(self, t, y):
        return y


def test_contract_errors():
    y0 = torch.zeros(3, 4)
    ts = torch.tensor([0.0, 1.0])
    sde = problems.make("gbm_ito", d=4)
    with pytest.raises(ValueError, match="noise_type"):
        contract.check_contract(_NoNoiseType(), y0, ts, None, None, False, None, None, False)
    with pytest.raises(ValueError, match="2-dimensional"):
        contract.check_contract(sde, torch.zeros(4), ts, None, None, False, None, None, False)
    with pytest.raises(ValueError, match="strictly increasing"):
        contract.check_contract(sde, y0, torch.tensor([0.0, 0.0]), None, None, False, None, None, False)
    with pytest.raises(ValueError, match="Expected method"):
        contract.check_contract(sde, y0, ts, None, "rk4", False, None, None, False)
    with pytest.raises(ValueError, match="Batch sizes not consistent"):
        bm = BrownianInterval(0.0, 1.0, size=(5, 4))
        contract.check_contract(sde, y0, ts, bm, None, False, None, None, False)
    with pytest.raises(ValueError, match="list/tuple of floats"):
        contract.check_contract(sde, y0, "0,1", None, None, False, None, None, False)
    scalar_bad = problems.ScalarTrig(4)
    scalar_bad.g = lambda t, y: torch.ones(y.size(0), 4, 2)
    with pytest.raises(ValueError, match="Scalar noise must have only one channel"):
        contract.check_contract(scalar_bad, y0, ts, None, None, False, None, None, False)


def test_contract_defaults():
    y0 = torch.zeros(3, 4)
    ts = [0.0, 1.0]
    for prob, method, levy in [("gbm_ito", "srk", "space-time"), ("general_ito", "euler", "none"),
                               ("gbm_strat", "midpoint", "none"), ("additive_ito", "srk", "space-time")]:
        sde, y, t, bm, m, opts = contract.check_contract(problems.make(prob), y0, ts, None, None, False, None, None,
                                                         False)
        assert isinstance(sde, ForwardSDE) and m == method and bm.levy_area_approximation == levy
        assert torch.is_tensor(t) and t.dtype == y0.dtype and opts == {}
        assert bm.shape == (3, 4 if prob != "additive_ito" else 3) and not bm.frozen


def test_forward_sde_dispatch_flags():
    assert not ForwardSDE(problems.GBMDiag(4)).user_product
    assert not ForwardSDE(problems.GBMViaFAndG(4)).user_product
    assert ForwardSDE(problems.GBMViaGProd(4)).user_product
    assert ForwardSDE(problems.GBMViaFAndGProd(4)).user_product


def test_solver_compatibility_errors():
    bm = BrownianInterval(0.0, 1.0, size=(3, 4))
    kw = dict(bm=bm, dt=0.1, adaptive=False, rtol=1e-5, atol=1e-4, dt_min=1e-5)
    with pytest.raises(ValueError, match="only supports noise types"):
        solvers.MilsteinIto(sde=ForwardSDE(problems.make("general_ito")), options={}, **kw)
    with pytest.raises(ValueError, match="levy_area_approximation"):
        solvers.SRK(sde=ForwardSDE(problems.make("gbm_ito")), options={}, **kw)
    with pytest.raises(ValueError, match="solver is for type"):
        solvers.Euler(sde=ForwardSDE(problems.make("gbm_strat")), options={}, **kw)
    with pytest.raises(ValueError, match="does not match any known method"):
        solvers.select("rk4", "ito")
    assert solvers.select("heun", "stratonovich") is solvers.Heun
    assert solvers.select("reversible_heun", "stratonovich") is solvers.ReversibleHeun


def test_unknown_kwargs_warn():
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        contract.handle_unused_kwargs({"foo": 1}, msg="`sdeint`")
    assert any("Unexpected arguments" in str(x.message) for x in w)


# ---- Brownian bookkeeping (host side only) -------------------------------------------------------------------------
def test_brownian_constructor_validation():
    with pytest.raises(ValueError, match="less than terminal"):
        BrownianInterval(1.0, 0.0, size=(2,))
    with pytest.raises(ValueError, match="`tol` should be positive"):
        BrownianInterval(0.0, 1.0, size=(2,), halfway_tree=True)
    with pytest.raises(ValueError, match="non-negative"):
        BrownianInterval(0.0, 1.0, size=(2,), tol=-1.0)
    with pytest.raises(ValueError, match="levy_area_approximation"):
        BrownianInterval(0.0, 1.0, size=(2,), levy_area_approximation="bogus")
    with pytest.raises(ValueError, match="Must either specify `size`"):
        BrownianInterval(0.0, 1.0)
    with pytest.raises(ValueError, match="Multiple sizes"):
        BrownianInterval(0.0, 1.0, size=(3,), W=torch.zeros(4))
    bm = BrownianInterval(0.0, 1.0, size=(2, 3), entropy=7, levy_area_approximation="space-time")
    assert bm.shape == (2, 3) and bm.size() == (2, 3) and bm.entropy == 7 and bm.dtype == torch.get_default_dtype()
    with pytest.raises(RuntimeError, match="must respect ta <= tb"):
        bm(0.7, 0.2)


def test_cell_bookkeeping():
    e = uniform_edges(0.0, 1.0, 0.25)
    assert np.array_equal(e, [0, 0.25, 0.5, 0.75, 1.0])
    e = uniform_edges(0.0, 1.0, 0.3)
    assert np.allclose(e, [0, 0.3, 0.6, 0.9, 1.0]) and e[-1] == 1.0
    bm = BrownianInterval(0.0, 1.0, size=(2, 2), dt=0.25)
    assert bm.frozen and list(bm.match_grid(np.array([0.25, 0.5, 0.75]))) == [1, 2]
    assert bm.match_grid(np.array([0.25, 0.75])) is None          # two cells in one step: general query
    assert bm.match_grid(np.array([0.1, 0.25])) is None
    assert bm.locate(0.3, 0.8) == (1, 3) and bm.locate(0.25, 0.5) == (1, 1) and bm.locate(0.0, 1.0) == (0, 3)
    lazy = BrownianInterval(0.0, 1.0, size=(2, 2))
    grid = np.array([0.0, 0.1, 0.2, 0.35])
    assert lazy.adopt_grid(grid) and np.array_equal(lazy._edges, [0.0, 0.1, 0.2, 0.35, 1.0])
    assert not lazy.adopt_grid(np.array([0.0, 0.5, 1.0]))          # already frozen: path is fixed
    assert list(lazy.match_grid(grid)) == [0, 1, 2]
    sharded = BrownianInterval(0.0, 1.0, size=(8, 3), row_offset=16)
    assert sharded._elem0 == 48


def test_adjoint_argument_errors():
    class Plain:
        noise_type, sde_type = "diagonal", "ito"

        def f(self, t, y):
            return y

        def g(self, t, y):
            return y
    with pytest.raises(ValueError, match="nn.Module"):
        torchsde_amd.sdeint_adjoint(Plain(), torch.zeros(2, 2), [0.0, 1.0])
    sde = problems.make("general_ito")
    with pytest.raises(ValueError, match="only supports noise types"):
        torchsde_amd.sdeint_adjoint(sde, torch.zeros(2, 4), [0.0, 1.0], adjoint_method="milstein")
    with pytest.raises(ValueError, match="Runge"):
        torchsde_amd.sdeint_adjoint(problems.make("gbm_ito"), torch.zeros(2, 4), [0.0, 1.0], adjoint_method="srk")


def test_output_times_reach_the_host_once():
    """A device `ts` is copied to the host once per tensor VERSION (a training loop reusing its `ts` never synchronises
    on it); the monotonicity check of the contract runs on that copy. (CPU tensors are simply read every time.)"""
    import numpy as np
    from torchsde_amd import contract, timegrid
    ts = torch.tensor([0.0, 0.25, 1.0])
    first = timegrid.ts_to_host(ts)
    assert not first.flags.writeable
    np.testing.assert_array_equal(first, np.asarray([0.0, 0.25, 1.0], dtype=np.float32))
    assert contract.is_strictly_increasing(ts)
    ts.numpy()[1] = 2.0                                # an edit the version counter does not see: still picked up
    assert timegrid.ts_to_host(ts)[1] == 2.0 and first[1] == 0.25
    assert not contract.is_strictly_increasing(ts)
    assert not contract.is_strictly_increasing(torch.tensor([0.0, 1.0, 1.0], dtype=torch.float64))
    assert contract.is_strictly_increasing([0.0, 0.5, 2.0]) and not contract.is_strictly_increasing((0.0, 0.0))

    class OnDevice:                                    # the caching rule, on a stand-in for a device tensor
        dtype, _version = torch.float32, 0
        device = torch.device("meta")

        def __init__(self):
            self.copies = 0

        def detach(self):
            return self

        def cpu(self):
            self.copies += 1
            return torch.tensor([0.0, 1.0])

    fake = OnDevice()
    a = timegrid.ts_to_host(fake)
    assert timegrid.ts_to_host(fake) is a and fake.copies == 1
    fake._version = 1                                  # an in-place op happened: the copy is redone
    assert timegrid.ts_to_host(fake) is not a and fake.copies == 2


def test_bench_latent_sde_statements_agree_on_cpu():
    """bench.py's two statements of the configs[4] latent SDE (user module / closed-form module) have the same
    parameters and the same f, g (plain torch on the CPU: this is about the modules, not the kernels)."""
    from workloads import configs
    user = configs.make_problem("latent_diag", 16, 16, "cpu")
    closed = configs.make_problem("latent_diag_closed_form", 16, 16, "cpu")
    y = torch.randn(5, 16, generator=torch.Generator().manual_seed(0))
    t = torch.tensor(0.0)
    assert torch.equal(user.f(t, y), closed.f(t, y)) and torch.equal(user.g(t, y), closed.g(t, y))
    assert closed.noise_type == user.noise_type == "diagonal" and closed.sde_type == user.sde_type == "ito"
    assert sum(p.numel() for p in user.parameters()) == sum(p.numel() for p in closed.parameters())
    spec = closed.closed_form(16, torch.float32, torch.device("cpu"))
    assert spec[0] == "mlp_diagonal" and spec[-1] == (1, 0.1) and spec[1].shape == (16, 16)


# ---- which backward sweep an `adjoint_method` string selects (no kernels involved) -----------------------------
def _fake_bm(levy="none"):
    class BM:
        levy_area_approximation = levy
    return BM()


@pytest.mark.parametrize("problem,adjoint_method,want", [
    ("gbm_ito", "euler", "euler"), ("gbm_ito", "milstein", "milstein"), ("general_ito", "euler", "euler"),
    ("gbm_strat", "midpoint", "midpoint"), ("gbm_strat", "milstein", "milstein"), ("gbm_strat", "heun", "heun"),
    ("gbm_strat", "euler_heun", "euler_heun"), ("general_strat", "heun", "heun"),
    ("general_strat", "euler_heun", "euler_heun"), ("scalar_strat", "heun", "heun"),
])
def test_adjoint_method_selects_its_own_backward_sweep(problem, adjoint_method, want):
    """Every solver the reference can run on an adjoint SDE (adjoint.py:83-93) maps to ITS backward sweep -- not to
    Milstein by default."""
    from torchsde_amd import adjoint
    sde = ForwardSDE(problems.make(problem))
    assert adjoint._backward_kind(sde, _fake_bm(), adjoint_method, {}, list(sde.parameters())) == want


@pytest.mark.parametrize("problem,adjoint_method,error,match", [
    ("gbm_strat", "log_ode", ValueError, "Log-ODE schemes cannot be used for adjoint SDEs"),      # log_ode.py:32-35
    ("gbm_strat", "reversible_heun", RuntimeError, "Adjoint `f_and_g` not defined"),      # adjoint_sde.py:262-264
    ("gbm_ito", "srk", ValueError, "Stochastic Runge"),                                          # srk.py:40-43
    ("gbm_ito", "heun", ValueError, "SDE is of type ito but solver is for type stratonovich"),
    ("general_strat", "milstein", ValueError, "noise type"),                                     # milstein.py:38-41
    ("gbm_strat", "adjoint_reversible_heun", None, None),
])
def test_unusable_adjoint_methods_fail_at_call_time(problem, adjoint_method, error, match):
    from torchsde_amd import adjoint
    sde = ForwardSDE(problems.make(problem))
    levy = "foster" if adjoint_method == "log_ode" else "none"
    if error is None:
        assert adjoint._backward_kind(sde, _fake_bm(), adjoint_method, {}, []) == "reversible_heun"
        return
    with pytest.raises(error, match=match):
        adjoint._check_adjoint_method(adjoint.AdjointSDE(sde, []), adjoint_method, {}, _fake_bm(levy))


# ---- the closed-form route is taken only for the dynamics the module itself states ---------------------------------
def test_closed_form_route_ignores_subclasses_that_change_the_dynamics():
    from torchsde_amd import closed_form

    class TimeDependentDrift(torchsde_amd.AffineDiagonalSDE):
        def f(self, t, y):
            return torch.sin(t) * y

    class OwnProduct(torchsde_amd.MLPDriftDiagonalSDE):
        def g_prod(self, t, y, v):
            return 2.0 * v

    class SameDynamics(torchsde_amd.AffineDiagonalSDE):
        def extra_report(self):
            return "nothing the solver sees"

    assert closed_form.publishes_its_own_dynamics(torchsde_amd.AffineDiagonalSDE(1.0, 0.0, 0.5, 0.0))
    assert closed_form.publishes_its_own_dynamics(torchsde_amd.MLPDriftDiagonalSDE(4, 4))
    assert closed_form.publishes_its_own_dynamics(SameDynamics(1.0, 0.0, 0.5, 0.0))
    assert not closed_form.publishes_its_own_dynamics(TimeDependentDrift(1.0, 0.0, 0.5, 0.0))
    assert not closed_form.publishes_its_own_dynamics(OwnProduct(4, 4))
    patched = torchsde_amd.AffineDiagonalSDE(1.0, 0.0, 0.5, 0.0)
    patched.g = lambda t, y: y * 0.0 + 1.0
    assert not closed_form.publishes_its_own_dynamics(patched)


# ---- an empty batch launches nothing (so it runs here): shapes of every return form -----------------------------------
def test_empty_batch_returns_the_reference_shapes():
    ts = torch.tensor([0.0, 0.25, 0.5])
    y = torch.zeros(0, 4)
    assert torchsde_amd.sdeint(problems.make("gbm_ito"), y, ts, dt=0.1, method="euler").shape == (3, 0, 4)
    ys, extras = torchsde_amd.sdeint(problems.make("gbm_strat"), y, ts, dt=0.1, method="reversible_heun", extra=True)
    assert ys.shape == (3, 0, 4) and [tuple(e.shape) for e in extras] == [(0, 4)] * 3
    ys, log_ratio = torchsde_amd.sdeint(problems.make("gbm_ito"), y, ts, dt=0.1, method="euler", logqp=True)
    assert ys.shape == (3, 0, 4) and log_ratio.shape == (2, 0)
    y_grad = torch.zeros(0, 4, requires_grad=True)
    sde = problems.make("general_ito", d=4, m=3)
    out = torchsde_amd.sdeint_adjoint(sde, y_grad, ts, dt=0.1)
    out.sum().backward()
    assert out.shape == (3, 0, 4) and y_grad.grad.shape == (0, 4)


def test_empty_brownian_interval_answers_without_a_device():
    bm = torchsde_amd.BrownianInterval(0.0, 1.0, size=(0, 3), levy_area_approximation="space-time")
    W, U = bm(0.1, 0.7, return_U=True)
    assert W.shape == (0, 3) and U.shape == (0, 3)


def test_bench_refuses_to_start_more_ranks_than_gpus():
    """`python bench.py --gpus N` on a node with fewer than N GPUs: a clear message and a non-zero exit code BEFORE any
    rank is started (RCCL would otherwise fail somewhere inside init with two ranks on one device)."""
    import subprocess
    import sys
    n = torch.cuda.device_count() + 1
    if n < 2:
        n = 2
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "TSDE_BENCH_SHARE_GPU")}
    proc = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", str(n), "--steps", "1"],
                          capture_output=True, text=True, timeout=300, env=env, cwd=root)
    assert proc.returncode not in (0, None)
    assert "not starting" in proc.stderr and not [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_srk_stage_decomposition_equals_the_reference_step(dtype):
    """The four SRID2 stage kernels hand partial sums to each other (H1_2 from stage 1, y1's running sum and H1_3's
    first two terms from stage 2, ...). Their torch twin (the differentiable form used when autograd runs through the
    solver; same operation order as csrc/steps.hip SrkDiagOp) must give the bits of the reference's step as the oracle
    restates it (oracle/solvers_ref.py srk_step <- methods/srk.py:57-88)."""
    from oracle import solvers_ref
    from torchsde_amd import kernels as K
    torch.manual_seed(5)
    B, d, dt = 37, 8, 2.0 ** -4
    sde = problems.make("gbm_ito", d=d).to(dtype)
    y0 = (0.1 + torch.rand(B, d)).to(dtype)
    W = (torch.randn(B, d) * dt ** 0.5).to(dtype)
    U = (dt * (0.5 * W + torch.randn(B, d).to(dtype) * (dt / 12) ** 0.5)).to(dtype)
    t0, t1 = torch.tensor(0.25, dtype=dtype), torch.tensor(0.25 + dt, dtype=dtype)

    def bm(ta, tb, return_U=False):
        return (W, U) if return_U else W
    with torch.no_grad():
        want = solvers_ref.srk_step(sde, bm, t0, t1, y0)
    noise = K.NoiseSpec.external(W, U)
    h = t1 - t0
    dtn, rdt, sqrt_dt = float(h), float(1 / h), float(h.sqrt())
    y = y0.clone().requires_grad_(True)      # grad mode: the torch twin of the stage kernels runs (CPU tensors)
    t_q, t_h = t0 + 0.25 * h, t0 + 0.5 * h
    f0, g0 = sde.f(t0, y), sde.g(t0, y)
    H0_1, H1_1, H1_2 = K.srk_diag_stage(1, (y, f0, g0), dtn, rdt, sqrt_dt, noise)
    f1, g1 = sde.f(t1, H0_1), sde.g(t_q, H1_1)
    H0_2, acc, P = K.srk_diag_stage(2, (y, f0, g0, f1, g1), dtn, rdt, sqrt_dt, noise)
    f2, g2 = sde.f(t_h, H0_2), sde.g(t1, H1_2)
    H1_3, acc = K.srk_diag_stage(3, (P, acc, f2, g2), dtn, rdt, sqrt_dt, noise)
    (y1,) = K.srk_diag_stage(4, (acc, sde.g(t_q, H1_3)), dtn, rdt, sqrt_dt, noise)
    assert torch.equal(y1.detach(), want)


def test_python_state_fingerprint_of_an_sde_object():
    """The key of the default ("auto") HIP-graph cache holds a fingerprint of the SDE object's Python-side state: stable
    across calls (also once this package's own cache sits on the object), changed by re-binding a plain attribute or a
    tensor, unchanged by in-place tensor updates (replays read tensor contents live)."""
    from torchsde_amd import graph
    sde = problems.make("gbm_ito", d=4)
    first = graph.python_state(sde)
    graph._cache_of(sde)
    sde.f(torch.tensor(0.0), torch.ones(2, 4))
    assert graph.python_state(sde) == first
    with torch.no_grad():
        for p in sde.parameters():
            p.mul_(0.5)
    assert graph.python_state(sde) == first
    sde.some_scale = 2.0
    second = graph.python_state(sde)
    assert second != first
    sde.some_scale = 3.0
    assert graph.python_state(sde) != second
    sde.some_scale = 2.0
    sde.ctx = (torch.zeros(3), [1, 2])
    third = graph.python_state(sde)
    sde.ctx[0].add_(1.0)
    assert graph.python_state(sde) == third
    sde.ctx = (torch.zeros(3), [1, 2])
    assert graph.python_state(sde) != third
    sde.train(False)
    assert graph.python_state(sde) != third
    big = problems.make("gbm_ito", d=4)
    big.table = list(range(10000))
    assert graph.python_state(big) is None          # too much state to fingerprint cheaply: "auto" stays eager
    assert graph.mode_of({}) == "auto" and graph.mode_of({"hip_graph": True}) is True
    assert graph.mode_of({"hip_graph": False}) is False and graph.mode_of(None) == "auto"
    with pytest.raises(ValueError):
        graph.mode_of({"hip_graph": "yes"})


def test_screening_tells_independent_drift_and_diffusion_from_code_that_shares_memory():
    """hip_graph="auto" may record drift and diffusion as parallel graph branches only if the screened eager run saw them
    touch disjoint memory (graph._OperatorRecorder, fed by ForwardSDE._f_then_g); and it only records operators it
    knows to be capture-safe."""
    from torch import nn
    from torchsde_amd import graph

    class Shared(nn.Module):
        noise_type, sde_type = "diagonal", "ito"

        def __init__(self):
            super().__init__()
            self.lin = nn.Linear(4, 4)

        def f(self, t, y):
            self.h = torch.tanh(self.lin(y))       # cached for g: one network evaluation serves both
            return -self.h

        def g(self, t, y):
            return 0.1 * self.h

    class Scratch(nn.Module):
        noise_type, sde_type = "diagonal", "ito"

        def __init__(self):
            super().__init__()
            self.buf = torch.zeros(3, 4)

        def f(self, t, y):
            return torch.mul(y, -0.5, out=self.buf) + 0.0     # both write the same scratch buffer

        def g(self, t, y):
            return torch.mul(y, 0.2, out=self.buf) + 0.0

    def screen(sde, extra=None):
        fs = ForwardSDE(sde)
        rec = graph._OperatorRecorder()
        graph._RECORDER = rec
        try:
            with rec, torch.no_grad():
                for _ in range(6):
                    fs.f_and_g(torch.tensor(0.0), torch.ones(3, 4))
                if extra is not None:
                    extra()
        finally:
            graph._RECORDER = None
        return rec

    for name in ("mlpdiag_ito", "gbm_ito"):
        rec = screen(problems.make(name, d=4))
        assert rec.independent() and not rec.unknown and rec.pairs == 6
    assert not screen(Shared()).independent()
    assert not screen(Scratch()).independent()
    rec = screen(problems.make("gbm_ito", d=4), extra=lambda: torch.linalg.pinv(torch.rand(3, 3)))
    assert any("pinv" in op or "svd" in op for op in rec.unknown)
    assert not graph._OperatorRecorder().independent()            # no drift / diffusion pair seen: not independent


def test_graph_cache_is_bounded_in_entries_and_bytes():
    from torchsde_amd import graph

    class Fake:
        def __init__(self, nbytes):
            self._t = torch.empty(nbytes // 4)

        def outputs(self):
            return [self._t]

    cache = graph._GraphCache()
    old_limit = graph._MAX_PINNED_BYTES
    graph._MAX_PINNED_BYTES = 4000
    try:
        for k in range(6):
            graph._remember(cache, ("sig", k), Fake(1000))
        assert list(cache) == [("sig", k) for k in (2, 3, 4, 5)]         # 4 x 1000 bytes fit, the oldest two went
        graph._remember(cache, ("big",), Fake(10000))                     # larger than the budget by itself: kept alone
        assert list(cache) == [("big",)]
    finally:
        graph._MAX_PINNED_BYTES = old_limit
    cache = graph._GraphCache()
    for k in range(graph._MAX_GRAPHS_PER_SDE + 5):
        graph._remember(cache, k, graph._Seen())
    assert len(cache) == graph._MAX_GRAPHS_PER_SDE
    import copy
    assert len(copy.deepcopy(cache)) == 0


def test_describe_cache_says_what_each_structure_does():
    from torchsde_amd import graph

    class FakeGraph:
        memset_nodes = (40, 40)

    class FakeSweep:
        graph = FakeGraph()
        tuning = {"kept": "sequential"}

    class Holder:
        pass

    sde = Holder()
    cache = graph._cache_of(sde)
    cache[("auto", 1, "Euler")] = graph._Seen(independent=True)
    cache[("auto", 2, "Midpoint")] = graph._Refused("the code synchronises with the host")
    cache[("adjoint-backward", "euler")] = FakeSweep()
    lines = graph.describe_cache(sde)
    assert lines[0].startswith("[auto] Euler: seen once") and "independent: yes" in lines[0]
    assert lines[1] == "[auto] Midpoint: stays eager: the code synchronises with the host"
    assert lines[2].startswith("[explicit] adjoint-backward: FakeSweep") and "40 of 40 memset nodes rewritten" in lines[2]
    assert graph.describe_cache(Holder()) == []


def test_bench_headline_is_the_last_line_and_fits_the_drivers_window(capsys, tmp_path, monkeypatch):
    """The driver keeps the tail of bench.py's stdout: round 3's line had grown to 21 KB (16 side measurements inside
    it) and nothing was parsed. The record of that very run (profiles/r3r_bench_c2_default.json) through this round's
    printer: side measurements as short lines BEFORE the headline and whole in bench_also.json, the headline LAST,
    under 4 KB, with `roofline.frac` and `cpu_baseline` in it."""
    import importlib
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    with open(os.path.join(root, "profiles", "r3r_bench_c2_default.json")) as fh:
        line = json.load(fh)
    also = line.pop("also")
    assert len(also) >= 16
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_digest", lambda: "0" * 16)
    # worst case: the prose of the old line kept, then doubled
    line["roofline"]["timing"] = line["roofline"]["timing"] * 2
    line["also_file"] = bench._emit_also(also)
    bench._print_headline(line)
    out = capsys.readouterr().out.splitlines()
    assert len(out) == len(also) + 1
    for text in out[:-1]:
        assert len(text) <= bench.ALSO_LINE_LIMIT and "also" in json.loads(text)
    assert len(out[-1]) < 4096
    head = json.loads(out[-1])
    assert head["roofline"]["frac"] > 0 and head["cpu_baseline"]["value"] > 0 and head["value"] > 0
    assert "also" not in head and head["also_file"] == "bench_also.json"
    with open(tmp_path / "bench_also.json") as fh:
        assert set(json.load(fh)["also"]) == set(also)
    assert sum(len(t) + 1 for t in out[-3:]) < 9000      # even a window of ~9 KB holds the headline whole


def test_lds_footprints_of_the_network_kernels():
    """What the host asks before it routes a module to a network kernel (host arithmetic of the C ABI: no GPU involved).
    General noise up to 32 state channels: the diffusion's last layer is laid out for the padded width (a compile-time row
    stride, csrc/mlp_general.hip PairLayout), so the footprint does not depend on d within a tile size; the BASELINE configs[2]
    shape fits the 160 KiB of a CU in both kernels (with 0.75 KiB to spare); above 32 channels it follows d."""
    from torchsde_amd import _native
    lib = _native.load()
    general, diagonal = 2, 0
    limit = 160 * 1024

    def euler(d, m, h, out):
        return lib.tsde_trajectory_mlp_general_lds(d, m, h, h, out, general)

    def rheun(d, m, h, out, mids=0):
        return lib.tsde_rheun_mlp_lds(d, m, h, h, out, general, mids, mids)
    assert 0 < euler(32, 16, 64, 512) <= limit and 0 < rheun(32, 16, 64, 512) <= limit
    assert euler(32, 16, 64, 512) == 4 * (2 * 32 * 68 + 64 * 36 + 64 * (512 + 8) + 4 * 64 + 32 + 512 + 32)
    assert euler(20, 16, 64, 320) == euler(32, 16, 64, 512) and rheun(17, 12, 40, 17 * 12) == rheun(32, 16, 64, 512)
    assert euler(36, 8, 40, 36 * 8) < euler(63, 7, 40, 63 * 7)                  # 64-channel tile: the real width counts
    assert rheun(32, 16, 64, 512, mids=1) > limit                              # (a hidden-to-hidden layer per net no longer fits)
    assert euler(65, 4, 16, 260) == 0 and rheun(16, 17, 16, 16 * 17) == 0      # no kernel for these shapes
    assert lib.tsde_trajectory_mlp_general_lds(64, 64, 128, 128, 64, diagonal) > 0


def test_bench_offline_counters_follow_the_sources_of_the_kernels_they_measured(tmp_path, monkeypatch):
    """profiles/traffic_latest.json is a measurement of the stepwise kernels: editing ANOTHER kernel's source (the digest over
    all sources changes) must not drop it from the bench line, editing steps.hip must; the side measurements are a list
    without repeats of known workloads, and their time budget leaves the rest marked as skipped."""
    import importlib
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys
    sys.path.insert(0, root)
    bench = importlib.import_module("bench")
    assert len(set(bench.ALSO)) == len(bench.ALSO) and all(w in bench.WORKLOADS for w in bench.ALSO)
    with open(os.path.join(root, "profiles", "traffic_latest.json")) as fh:
        rec = json.load(fh)
    files = dict(bench.csrc_file_digests())
    rec["files"], rec["csrc_sha"] = dict(files), "f" * 16        # collected at other sources, same per-file hashes
    (tmp_path / "profiles").mkdir()
    with open(tmp_path / "profiles" / "traffic_latest.json", "w") as fh:
        json.dump(rec, fh)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "csrc_digest", lambda: "0" * 16)
    monkeypatch.setattr(bench, "csrc_file_digests", lambda: dict(files, **{"mlp_general.hip": "changed"}))
    roof = {}
    bench._attach_offline_traffic(roof, "c2_srk_diag")
    assert roof.get("traffic_over_algorithmic", 0) > 1.0 and "unchanged" in roof["traffic_source"]
    monkeypatch.setattr(bench, "csrc_file_digests", lambda: dict(files, **{"steps.hip": "changed"}))
    roof = {}
    bench._attach_offline_traffic(roof, "c2_srk_diag")
    assert "traffic" not in roof and "steps.hip changed" in roof["traffic_source"]
    # the time budget of the side measurements
    ran = []
    monkeypatch.setattr(bench, "_side_measurement", lambda dev, name: ran.append(name) or {"ms_per_solve": 1.0})
    monkeypatch.setattr(bench.torch.cuda, "empty_cache", lambda: None)
    ticks = iter(range(0, 10 ** 6, 40))
    monkeypatch.setattr(bench.time, "time", lambda: next(ticks))
    also = bench._side_measurements(None, budget=150.0)
    assert ran == list(bench.ALSO[:3]) and set(also) == set(bench.ALSO)
    assert all("skipped" in also[w] for w in bench.ALSO[3:])


_FINGERPRINT_GLOBAL_SCALE = 1.5


def _fingerprint_helper(y):
    return _FINGERPRINT_GLOBAL_SCALE * y


def test_python_state_sees_globals_closures_class_attributes_and_strides(monkeypatch):
    """What round 3's fingerprint missed (VERDICT weak 2, ADVICE): a drift that reads a module-level number -- directly
    or through a helper function --, a closed-over list element, a class attribute; values whose CPython hashes collide
    (hash(-1) == hash(-2)); a square matrix re-bound to its transpose (same storage and shape, other strides)."""
    import sys
    from torchsde_amd import graph
    this = sys.modules[__name__]
    coefficients = [0.5, 2.0]

    class SDE(torch.nn.Module):
        noise_type, sde_type = "diagonal", "ito"
        gain = 3.0

        def __init__(self):
            super().__init__()
            self.sign = -1
            self.W = torch.eye(3)
            self.g = lambda t, y: coefficients[1] * y        # an instance attribute holding a closure

        def f(self, t, y):
            return _fingerprint_helper(y) * self.gain

    sde = SDE()
    state = graph.python_state(sde)
    assert state is not None and graph.python_state(sde) == state and hash(state) == hash(graph.python_state(sde))

    def changed(action, undo):
        action()
        differs = graph.python_state(sde) != state
        undo()
        assert graph.python_state(sde) == state
        return differs

    assert changed(lambda: monkeypatch.setattr(this, "_FINGERPRINT_GLOBAL_SCALE", 2.5),
                   lambda: monkeypatch.setattr(this, "_FINGERPRINT_GLOBAL_SCALE", 1.5))
    assert changed(lambda: coefficients.__setitem__(1, 4.0), lambda: coefficients.__setitem__(1, 2.0))
    assert changed(lambda: setattr(SDE, "gain", 4.0), lambda: setattr(SDE, "gain", 3.0))
    assert hash((1, -1, "a")) == hash((1, -2, "a"))           # the collision the old `hash(tuple(...))` key fell into
    assert changed(lambda: setattr(sde, "sign", -2), lambda: setattr(sde, "sign", -1))
    assert changed(lambda: setattr(sde, "sign", -1.0), lambda: setattr(sde, "sign", -1))       # 1 vs 1.0: other program
    W = sde.W
    assert changed(lambda: setattr(sde, "W", W.t()), lambda: setattr(sde, "W", W))
    sde.rate = float("nan")
    assert graph.python_state(sde) == graph.python_state(sde)       # nan is one value, not a miss on every solve
    assert graph.process_state() == graph.process_state()
    torch.set_float32_matmul_precision("high")
    try:
        assert graph.process_state() != ("x",) and graph.process_state()[2] == "high"
    finally:
        torch.set_float32_matmul_precision("highest")


def test_auto_key_refuses_an_object_whose_state_never_repeats_and_schedule_of_checks(monkeypatch):
    """An SDE object that changes on every call (a call counter) never reaches a replay: after 8 distinct states of one
    structure the structure is refused and later solves skip the fingerprint walk. Accepted graphs are
    re-checked against the eager path on replays 1, 2, 8, 64, 512, ... TSDE_HIP_GRAPH sets the default and 0 overrides."""
    from torchsde_amd import graph
    sde = problems.make("gbm_ito", d=4)
    cache, structure = graph._GraphCache(), ("Euler", (4, 4), "float32")
    first = graph.auto_key(cache, structure, sde)
    cache[first] = graph._Seen()
    assert first is not None and graph.auto_key(cache, structure, sde) == first
    keys = set()
    for i in range(graph._MAX_STATES_PER_STRUCTURE + 2):
        sde.calls = i
        key = graph.auto_key(cache, structure, sde)
        if key is not None:
            keys.add(key)
            cache[key] = graph._Seen()
    # (the hit above reset the count: misses are counted CONSECUTIVELY)
    assert len(keys) == graph._MAX_STATES_PER_STRUCTURE and graph.auto_key(cache, structure, sde) is None
    assert not [k for k in cache if k[0] == "auto"]                       # the churned entries are gone
    # a sweep over many legitimate states with cache hits in between is not churn, and a graph that has replayed survives
    sweep, swept = graph._GraphCache(), problems.make("gbm_ito", d=4)
    for i in range(3 * graph._MAX_STATES_PER_STRUCTURE):
        swept.scale = float(i)
        key = graph.auto_key(sweep, structure, swept)
        assert key is not None
        sweep[key] = graph._Seen()
        assert graph.auto_key(sweep, structure, swept) == key            # the second solve in this state: a hit
    walked = []
    monkeypatch.setattr(graph, "python_state", lambda base: walked.append(1))
    assert graph.auto_key(cache, structure, sde) is None and not walked   # refused before any fingerprinting
    assert graph.auto_key(cache, ("Milstein",) + structure[1:], sde) is None and walked     # (other structure: not refused)
    assert any("stays eager" in line for line in graph.describe_cache(type("S", (), {graph._CACHE_ATTR: cache})()))
    assert [n for n in range(1, 5000) if graph.due_for_a_check(n)] == [1, 2, 8, 64, 512, 4096]
    monkeypatch.setenv("TSDE_HIP_GRAPH", "0")
    assert graph.mode_of({}) is False and graph.mode_of({"hip_graph": True}) is False and graph.mode_of(None) is False
    monkeypatch.setenv("TSDE_HIP_GRAPH", "1")
    assert graph.mode_of({}) is True and graph.mode_of({"hip_graph": False}) is False and graph.mode_of({"hip_graph": "auto"}) == "auto"
    monkeypatch.setenv("TSDE_HIP_GRAPH", "auto")
    assert graph.mode_of({}) == "auto"
    monkeypatch.setenv("TSDE_HIP_GRAPH", "sometimes")
    with pytest.raises(ValueError):
        graph.mode_of({})


def test_modules_that_must_not_be_called_an_extra_time_are_not_interpreted():
    """recognise.py calls f and g once per solve on a two-row probe: a normalisation layer in training mode would take
    the probe into its running statistics, a compiled module would recompile under the dispatch mode."""
    from torch import nn
    from torchsde_amd.solvers import BaseSDESolver

    class WithBatchNorm(nn.Module):
        noise_type, sde_type = "diagonal", "ito"

        def __init__(self):
            super().__init__()
            self.norm = nn.BatchNorm1d(4)

    sde = WithBatchNorm()
    assert not BaseSDESolver._may_be_interpreted(sde)
    sde.eval()
    assert BaseSDESolver._may_be_interpreted(sde)
    assert BaseSDESolver._may_be_interpreted(problems.make("gbm_ito", d=4))
    assert not BaseSDESolver._may_be_interpreted(type("OptimizedModule", (), {})())


def test_matrix_precision_option_is_validated_before_any_route():
    """`options["matrix_precision"]` (opt-in split-bf16 products of the neural general-noise kernel): anything but "f32" /
    "bf16x3" raises on every route, CPU tensors included (before the no-CPU-fallback error)."""
    import torch
    import torchsde_amd
    from workloads import problems
    sde = problems.MLPGeneral(4, 2, "ito")
    with pytest.raises(ValueError, match="matrix_precision"):
        torchsde_amd.sdeint(sde, torch.zeros(8, 4), torch.tensor([0.0, 1.0]), dt=0.1, method="euler",
                            options={"matrix_precision": "fp8"})


def test_adaptive_round_budget():
    """The number of attempts the host enqueues before it synchronises (adaptive.round_budget): an upper bound from the
    current step size, the previous solve's count when there is one, nothing when every output time is met, capped."""
    from torchsde_amd.adaptive import round_budget
    assert round_budget(1.0, 0.1) == 9                  # ceil(10) - 1: steps only grow
    assert round_budget(0.15, 0.1) == 2 and round_budget(0.05, 0.1) == 1
    assert round_budget(0.0, 0.1) == 0 and round_budget(-1e-9, 0.1) == 0
    assert round_budget(1.0, 0.1, hint=25) == 26 and round_budget(0.0, 0.1, hint=25) == 0
    assert round_budget(100.0, 0.001) == 256 and round_budget(1.0, 0.1, hint=1000) == 256


def test_pure_call_counters_are_told_from_state_that_reaches_the_dynamics():
    """graph.call_counters: `self._nfe += 1` in f / g / h with a getter nothing else calls (the reference's Ex* test problems,
    tests/problems.py:60-66, 92-98, 118-124) is a pure counter; a counter that is read anywhere else -- in f itself, through its
    getter, in a lambda, in a module-level helper -- or that two classes of the MRO both advance, is state."""
    import torch
    from torch import nn
    from torchsde_amd import graph

    class Counted(nn.Module):
        def __init__(self):
            super().__init__()
            self._nfe, self.width = 0, 3
            self.p = nn.Parameter(torch.ones(4))

        def f(self, t, y):
            self._nfe += 1
            return -self.p * y

        def g(self, t, y):
            self._nfe += 1
            return 0.1 * y

        def h(self, t, y):
            self._nfe += 1
            return torch.zeros_like(y)

        @property
        def nfe(self):
            return self._nfe

    class Gate(Counted):
        def f(self, t, y):
            self._nfe += 1
            return -y if self._nfe > 100 else -2 * y

    class ThroughGetter(Counted):
        def g(self, t, y):
            self._nfe += 1
            return 0.1 * y * self.nfe

    class InLambda(Counted):
        def f(self, t, y):
            self._nfe += 1
            return (lambda: -y * self._nfe)()

    class ViaHelper(Counted):
        def f(self, t, y):
            self._nfe += 1
            return -y * _helper_that_reads_the_counter(self)

    class TwiceInTheMro(Counted):
        def g(self, t, y):
            self._nfe += 2
            return 0.1 * y

    assert graph.call_counters(Counted()) == {"_nfe": {"f": 1, "g": 1, "h": 1}}
    for cls in (Gate, ThroughGetter, InLambda, ViaHelper, TwiceInTheMro):
        assert graph.call_counters(cls()) == {}, cls.__name__
    sde = Counted()
    before = graph.python_state(sde, ignore=("_nfe",))
    sde.f(torch.tensor(0.0), torch.zeros(2, 4))
    assert graph.python_state(sde, ignore=("_nfe",)) == before and graph.python_state(sde) != before


def _helper_that_reads_the_counter(sde):
    return 1.0 if sde._nfe < 10 else 2.0

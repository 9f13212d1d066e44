"""GPU parity tests (run with ``-m gpu`` on an MI355X): the HIP path vs the golden reference outputs, the
oracle, and itself (fused vs materialised increments, sharded vs unsharded)."""
import numpy as np
import pytest
import torch

from oracle import counter, solvers_ref
from tests import helpers
from workloads import problems

pytestmark = pytest.mark.gpu

DEV = "cuda"
NATIVE_UNSUPPORTED = set()


def _exact_expected(case):
    """Problems whose f, g are pure elementwise chains: the HIP path must be bit-identical to the reference CPU."""
    return case.problem.startswith("gbm")


# ------------------------------------------------------------------------------------------------------------
# 1. Replayed increments: torchsde_amd.sdeint == the REAL reference (golden), through the foreign-bm seam.
@pytest.mark.parametrize("tag", ["f32", "f64"])
@pytest.mark.parametrize("name", [n for n in helpers.solver_cases() if n not in NATIVE_UNSUPPORTED])
def test_replay_matches_reference_golden(name, tag):
    import torchsde_amd
    case = helpers.Case(name, tag)
    sde = case.sde(DEV)
    bm = helpers.make_replay_bm(case.table(DEV), (case.B, case.m), case.dtype, DEV, case.levy)
    with torch.no_grad():
        ys = torchsde_amd.sdeint(sde, case.y0(DEV), case.ts.to(DEV), bm=bm, method=case.method, dt=case.dt,
                                 options=case.options)
    ys = ys.cpu()
    assert ys.shape == case.ys.shape
    if _exact_expected(case):
        assert torch.equal(ys, case.ys), f"max diff {(ys - case.ys).abs().max().item():.3e}"
    else:
        # f/g contain transcendental functions / matmuls evaluated by different libraries on CPU and GPU
        rtol, atol = (2e-5, 2e-6) if tag == "f32" else (1e-10, 1e-12)
        torch.testing.assert_close(ys, case.ys, rtol=rtol, atol=atol)


# ------------------------------------------------------------------------------------------------------------
# 2. The generator: Philox bits exact, normals and bridge queries within fp32 transcendental tolerance.
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 2e-5), (torch.float64, 1e-12)])
@pytest.mark.parametrize("n,elem0,cell,node,stream", [(4096, 0, 0, 0, 0), (1001, 3, 7, 5, 1), (64, 2 ** 33 + 4, 2 ** 31, 2 ** 35 + 1, 1)])
def test_normals_match_oracle(dtype, tol, n, elem0, cell, node, stream):
    from torchsde_amd import _native
    lib = _native.load()
    out = torch.empty(n, dtype=dtype, device=DEV)
    code = lib.tsde_brownian_normals(_native.ptr(out), n, 99991, elem0, cell, node, stream, _native.dtype_code(dtype),
                                     _native.stream_ptr())
    _native.check(code, "tsde_brownian_normals")
    ref = counter.normals(n, 99991, elem0, cell, node, stream)
    err = np.abs(out.cpu().double().numpy() - ref).max()
    assert err <= tol, err


def test_normal_moments_large():
    from torchsde_amd import _native
    lib = _native.load()
    n = 1 << 22
    out = torch.empty(n, dtype=torch.float32, device=DEV)
    _native.check(lib.tsde_brownian_normals(_native.ptr(out), n, 5, 0, 0, 0, 0, 0, _native.stream_ptr()), "normals")
    x = out.double()
    assert abs(x.mean().item()) < 4 / np.sqrt(n)
    assert abs(x.var().item() - 1) < 6 * np.sqrt(2 / n)
    assert abs((x ** 3).mean().item()) < 6 * np.sqrt(15 / n)
    assert abs((x ** 4).mean().item() - 3) < 6 * np.sqrt(96 / n)


QUERY_GRIDS = {
    "single": np.array([0.0, 1.0]),
    "uniform": np.arange(0, 17) / 16.0,
    "ragged": np.array([0.0, 0.11, 0.35, 0.36, 0.8, 1.0]),
}


@pytest.mark.parametrize("levy", ["none", "space-time"])
@pytest.mark.parametrize("grid", list(QUERY_GRIDS))
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_query_matches_oracle(levy, grid, dtype):
    import torchsde_amd
    edges = QUERY_GRIDS[grid]
    size = (37, 3)   # 111 elements: not a multiple of 4 -> masked path
    bm = torchsde_amd.BrownianInterval(0.0, 1.0, size=size, dtype=dtype, device=DEV, entropy=314159,
                                       levy_area_approximation=levy, row_offset=5)
    bm._freeze(edges)
    have_h = levy != "none"
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    tol = 3e-5 if dtype == torch.float32 else 1e-11
    rng = np.random.default_rng(0)
    queries = [(0.0, 1.0), (edges[-2], edges[-1]), (edges[0], edges[1]), (0.05, 0.07), (0.3, 0.99)]
    if len(edges) > 3:
        queries += [(edges[1], edges[2]), (edges[0], edges[-2]), (edges[1], edges[-1])]
    queries += [tuple(sorted(rng.uniform(0, 1, 2))) for _ in range(12)]
    for a, b in queries:
        W, U = bm.increment(a, b, want_U=have_h)
        Wr, Ur, _ = counter.query(size[0] * size[1], 314159, edges, a, b, dtype=np_dt, elem0=5 * 3, have_h=have_h)
        assert np.abs(W.cpu().numpy().ravel() - Wr).max() <= tol, (a, b)
        if have_h:
            assert np.abs(U.cpu().numpy().ravel() - Ur).max() <= tol, (a, b)


def test_query_pinned_root_and_halfway_tree_match_oracle():
    import torchsde_amd
    n = 64
    Wroot = torch.linspace(-1, 1, n, dtype=torch.float64, device=DEV)
    bm = torchsde_amd.BrownianInterval(0.0, 2.0, W=Wroot, levy_area_approximation="space-time", entropy=11)
    for a, b in [(0.0, 2.0), (0.0, 0.7), (0.7, 2.0), (0.3, 1.9)]:
        W, U = bm.increment(a, b, want_U=True)
        Wr, Ur, _ = counter.query(n, 11, [0.0, 2.0], a, b, dtype=np.float64, have_h=True, rootW=Wroot.cpu().numpy())
        assert np.abs(W.cpu().numpy() - Wr).max() < 1e-11
        assert np.abs(U.cpu().numpy() - Ur).max() < 1e-11
    W_full, _ = bm.increment(0.0, 2.0)
    assert torch.equal(W_full, Wroot)
    tree = torchsde_amd.BrownianInterval(0.0, 1.0, size=(n,), dtype=torch.float64, device=DEV, entropy=12, tol=1e-6,
                                         halfway_tree=True)
    for a, b in [(0.1234567, 0.7654321), (0.5, 0.75), (0.0, 0.3333333)]:
        W, _ = tree.increment(a, b)
        Wr, _, _ = counter.query(n, 12, [0.0, 1.0], round(a, 6), round(b, 6), dtype=np.float64,
                                 max_depth=tree._max_depth, snap=1)
        assert np.abs(W.cpu().numpy() - Wr).max() < 1e-11


# ------------------------------------------------------------------------------------------------------------
# 3. Fused in-register increments == materialised increments of the same Brownian motion (bit-exact), and
#    == the oracle's reference arithmetic driven by the oracle twin of the generator.
FUSED_CASES = [
    ("gbm_ito", "euler", None, "none", (64, 8, 8)),
    ("gbm_ito", "milstein", None, "none", (64, 8, 8)),
    ("gbm_ito", "milstein", {"grad_free": True}, "none", (64, 8, 8)),
    ("gbm_ito", "srk", None, "space-time", (64, 8, 8)),
    ("gbm_strat", "midpoint", None, "none", (64, 8, 8)),
    ("gbm_strat", "milstein", None, "none", (63, 5, 5)),
    ("general_ito", "euler", None, "none", (48, 4, 4)),
    ("general_odd_ito", "euler", None, "none", (48, 3, 5)),
    ("general_strat", "midpoint", None, "none", (48, 4, 4)),
    ("scalar_ito", "euler", None, "none", (48, 4, 1)),
    ("scalar_ito", "milstein", None, "none", (48, 4, 1)),
    ("scalar_ito", "srk", None, "space-time", (48, 4, 1)),
    ("additive_ito", "euler", None, "none", (48, 4, 3)),
    ("additive_ito", "srk", None, "space-time", (48, 4, 3)),
    ("additive_ito", "milstein", None, "none", (48, 4, 4)),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("prob,method,options,levy,shape", FUSED_CASES)
def test_fused_equals_materialised_and_oracle(prob, method, options, levy, shape, dtype):
    import torchsde_amd
    B, d, m = shape
    steps, dt = 12, 2.0 ** -5
    ts = torch.tensor([0.0, 5 * dt, steps * dt], dtype=dtype, device=DEV)
    sde = problems.make(prob, dtype=dtype, d=d, m=m).to(DEV)
    y0 = torch.full((B, d), 0.1, dtype=dtype, device=DEV)
    kw = dict(t0=0.0, t1=steps * dt, size=(B, m), dtype=dtype, device=DEV, entropy=777,
              levy_area_approximation=levy, dt=dt)
    opts = None if options is None else dict(options)
    with torch.no_grad():
        ys_fused = torchsde_amd.sdeint(sde, y0, ts, bm=torchsde_amd.BrownianInterval(**kw), method=method, dt=dt,
                                       options=opts)
        # the same Brownian motion, but seen by the solver as a foreign object -> materialised increments
        inner = torchsde_amd.BrownianInterval(**kw)

        class Foreign(torchsde_amd.BaseBrownian):
            def __call__(self, ta, tb=None, return_U=False, return_A=False):
                return inner(ta, tb, return_U=return_U)

            def __repr__(self):
                return "Foreign"
            dtype = property(lambda s: inner.dtype)
            device = property(lambda s: inner.device)
            shape = property(lambda s: inner.shape)
            levy_area_approximation = property(lambda s: inner.levy_area_approximation)

        ys_mat = torchsde_amd.sdeint(sde, y0, ts, bm=Foreign(), method=method, dt=dt,
                                     options=None if options is None else dict(options))
    assert torch.equal(ys_fused, ys_mat), (ys_fused - ys_mat).abs().max().item()

    # oracle: reference arithmetic on CPU + C twin of the generator
    np_dt = np.float32 if dtype == torch.float32 else np.float64
    edges = np.arange(steps + 1) * dt

    def bm_cpu(ta, tb, return_U=False):
        W, U, _ = counter.query(B * m, 777, edges, float(ta), float(tb), dtype=np_dt, have_h=(levy != "none"))
        W = torch.from_numpy(W).reshape(B, m)
        return (W, torch.from_numpy(U).reshape(B, m)) if return_U else W

    with torch.no_grad():
        ref = solvers_ref.integrate(sde.cpu(), bm_cpu, y0.cpu(), ts.cpu(), dt, method, options)
    rtol, atol = (3e-5, 3e-6) if dtype == torch.float32 else (1e-10, 1e-12)
    torch.testing.assert_close(ys_fused.cpu(), ref, rtol=rtol, atol=atol)


def test_sharded_rows_bit_identical():
    """Rows [r0, r1) solved alone with row_offset=r0 equal the same rows of the full batch (RNG uses global rows)."""
    import torchsde_amd
    B, d, steps, dt = 96, 8, 8, 2.0 ** -4
    ts = torch.tensor([0.0, steps * dt], device=DEV)
    sde = problems.make("gbm_ito", d=d).to(DEV)
    y0 = torch.rand(B, d, device=DEV, generator=torch.Generator(DEV).manual_seed(0)) * 0.2
    with torch.no_grad():
        full = torchsde_amd.sdeint(sde, y0, ts, bm=torchsde_amd.BrownianInterval(
            0.0, steps * dt, size=(B, d), device=DEV, dtype=torch.float32, entropy=3, dt=dt), method="euler", dt=dt)
        parts = []
        for r0 in range(0, B, 32):
            bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(32, d), device=DEV, dtype=torch.float32,
                                               entropy=3, dt=dt, row_offset=r0)
            parts.append(torchsde_amd.sdeint(sde, y0[r0:r0 + 32], ts, bm=bm, method="euler", dt=dt))
    assert torch.equal(full, torch.cat(parts, dim=1))


def test_nondyadic_fp32_grid_default_bm():
    """dt=1e-3 in float32: 1001 steps like the reference; the default bm adopts the solver grid (fused path)."""
    import torchsde_amd
    sde = problems.make("gbm_ito", d=8).to(DEV)
    y0 = torch.full((16, 8), 0.1, device=DEV)
    ts = torch.tensor([0.0, 0.4, 1.0], device=DEV)
    with torch.no_grad():
        ys = torchsde_amd.sdeint(sde, y0, ts, method="euler", dt=1e-3)
    assert ys.shape == (3, 16, 8)
    assert torch.isfinite(ys).all()
    assert torch.equal(ys[0], y0)


# ------------------------------------------------------------------------------------------------------------
# 4. Provider dispatch: the same SDE exposed through f/g, f_and_g, g_prod, f_and_g_prod gives bit-identical
#    results under the same entropy (reference tests/test_sdeint.py:79-98 `test_specialised_functions`).
@pytest.mark.parametrize("method,sde_type", [("euler", "ito"), ("midpoint", "stratonovich")])
def test_specialised_functions_bit_identical(method, sde_type):
    import torchsde_amd
    B, d, steps, dt = 32, 8, 8, 2.0 ** -4
    ts = torch.tensor([0.0, steps * dt], device=DEV)
    y0 = torch.full((B, d), 0.1, device=DEV)
    outs = []
    for cls in (problems.GBMDiag, problems.GBMViaFAndG, problems.GBMViaGProd, problems.GBMViaFAndGProd):
        sde = cls(d, sde_type).to(DEV)
        bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(B, d), device=DEV, dtype=torch.float32, entropy=8)
        with torch.no_grad():
            outs.append(torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=dt))
    for o in outs[1:]:
        assert torch.equal(outs[0], o)


@pytest.mark.parametrize("method,sde_type", [("euler", "ito"), ("midpoint", "stratonovich")])
def test_specialised_functions_general_noise(method, sde_type):
    """The same for GENERAL noise, with the six provider combinations of the reference's test (tests/problems.py:356-440):
    f + g, f_and_g, f + g_prod (no g at all), f_and_g_prod, and f_and_g together with either product. Where the module
    supplies the product, the user's `bmm` sums the m terms, elsewhere the contraction kernel does: same path, the sums
    in another order -- float32 rounding, not bit equality."""
    import torchsde_amd
    B, d, m, steps, dt = 32, 5, 3, 8, 2.0 ** -4
    vector = torch.randn(m, generator=torch.Generator().manual_seed(3)).to(DEV)

    def drift(y):
        return -y

    def diffusion(y):
        return y.unsqueeze(-1).sigmoid() * vector

    def product(y, v):
        return diffusion(y).bmm(v.unsqueeze(-1)).squeeze(-1)

    def module(**methods):
        cls = type("Provided", (torch.nn.Module,), dict(noise_type="general", sde_type=sde_type, **methods))
        return cls()

    forms = [
        module(f=lambda self, t, y: drift(y), g=lambda self, t, y: diffusion(y)),
        module(f_and_g=lambda self, t, y: (drift(y), diffusion(y))),
        module(f=lambda self, t, y: drift(y), g_prod=lambda self, t, y, v: product(y, v)),
        module(f_and_g_prod=lambda self, t, y, v: (drift(y), product(y, v))),
        module(f_and_g=lambda self, t, y: (drift(y), diffusion(y)), g_prod=lambda self, t, y, v: product(y, v)),
        module(f_and_g=lambda self, t, y: (drift(y), diffusion(y)),
               f_and_g_prod=lambda self, t, y, v: (drift(y), product(y, v))),
    ]
    ts = torch.tensor([0.0, steps * dt], device=DEV)
    y0 = torch.randn(B, d, generator=torch.Generator().manual_seed(4)).to(DEV)
    outs = []
    for sde in forms:
        bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(B, m), device=DEV, dtype=torch.float32, entropy=45678)
        with torch.no_grad():
            outs.append(torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=dt)[1])
    for o in outs[1:]:
        assert o.shape == outs[0].shape
        torch.testing.assert_close(o, outs[0], rtol=2e-6, atol=2e-6)


def test_logqp_and_names():
    """logqp=True appends the KL column and returns log-ratio increments; names= renames drift/diffusion
    (reference sdeint.py:142-144,284-295; base_sde.py:212-306). Numbers against the reference itself are in
    tests/test_gpu_logqp.py; here a case with a closed form: constant diffusion s, f = -theta y, h = -y, so
    u = (1 - theta) y / s and d(log-ratio)/dt = 0.5 |u|^2 on the solver's own Euler states."""
    import torchsde_amd

    class Latent(torch.nn.Module):
        noise_type, sde_type = "diagonal", "ito"

        def __init__(self):
            super().__init__()
            self.theta = torch.nn.Parameter(torch.tensor(0.5))

        def drift(self, t, y):
            return -self.theta * y

        def diffusion(self, t, y):
            return 0.3 + 0.0 * y

        def h(self, t, y):
            return -y

    sde = Latent().to(DEV)
    dt = 2.0 ** -5
    y0 = torch.full((16, 4), 0.2, device=DEV)
    ts = torch.tensor([k * dt for k in range(17)], device=DEV)
    bm = torchsde_amd.BrownianInterval(0.0, 0.5, size=(16, 5), device=DEV, dtype=torch.float32, entropy=4)
    with torch.no_grad():
        ys, logqp = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="euler", dt=dt, logqp=True,
                                        names={"drift": "drift", "diffusion": "diffusion"})
    assert ys.shape == (17, 16, 4) and logqp.shape == (16, 16)
    want = 0.5 * (((1 - 0.5) * ys[:-1] / 0.3) ** 2).sum(dim=2) * dt      # Euler: increment = integrand(y_k) * dt
    torch.testing.assert_close(logqp, want, rtol=2e-4, atol=1e-6)


# ------------------------------------------------------------------------------------------------------------
# 5. Adaptive stepping (step doubling on the virtual bridge tree) vs the oracle's restatement of the reference's
#    adaptive loop (pinned to the real reference on CPU, tests/test_oracle_solvers.py) driven by the C twin of the
#    generator. Accept/reject control flow must be identical; the proposed step sizes depend on the error norm to
#    the last ulp (GPU vs CPU reductions), so query times drift by ~1e-12 and the Brownian values with them:
#    and the error norm feeds back chaotically: times are compared at 1e-2, outputs at 2e-3.
@pytest.mark.parametrize("prob,method,levy", [("gbm_ito", "milstein", "none"), ("gbm_ito", "srk", "space-time"),
                                              ("gbm_strat", "midpoint", "none")])
def test_adaptive_matches_oracle(prob, method, levy):
    import torchsde_amd
    B, d = 16, 4
    dtype = torch.float64
    ts = torch.tensor([0.0, 0.4, 1.0], dtype=dtype, device=DEV)
    sde = problems.make(prob, dtype=dtype, d=d).to(DEV)
    y0 = torch.full((B, d), 0.1, dtype=dtype, device=DEV)
    inner = torchsde_amd.BrownianInterval(0.0, 1.0, size=(B, d), dtype=dtype, device=DEV, entropy=2718,
                                          levy_area_approximation=levy)
    gpu_queries = []

    class Recording(torchsde_amd.BaseBrownian):
        def __call__(self, ta, tb=None, return_U=False, return_A=False):
            gpu_queries.append((float(ta), float(tb)))
            return inner(ta, tb, return_U=return_U)

        def __repr__(self):
            return "Recording"
        dtype = property(lambda s: inner.dtype)
        device = property(lambda s: inner.device)
        shape = property(lambda s: inner.shape)
        levy_area_approximation = property(lambda s: inner.levy_area_approximation)

    with torch.no_grad():
        ys = torchsde_amd.sdeint(sde, y0, ts, bm=Recording(), method=method, dt=0.1, adaptive=True, rtol=1e-3,
                                 atol=1e-3)

    cpu_queries = []

    def bm_cpu(ta, tb, return_U=False):
        cpu_queries.append((float(ta), float(tb)))
        W, U, _ = counter.query(B * d, 2718, [0.0, 1.0], float(ta), float(tb), dtype=np.float64,
                                have_h=(levy != "none"))
        W = torch.from_numpy(W).reshape(B, d)
        return (W, torch.from_numpy(U).reshape(B, d)) if return_U else W

    with torch.no_grad():
        ref = solvers_ref.integrate(sde.cpu(), bm_cpu, y0.cpu(), ts.cpu(), 0.1, method, adaptive=True, rtol=1e-3,
                                    atol=1e-3)
    assert len(gpu_queries) == len(cpu_queries) >= 15
    assert len({round(b - a, 9) for a, b in cpu_queries}) >= 4          # the controller really adapted
    np.testing.assert_allclose(np.array(gpu_queries), np.array(cpu_queries), rtol=1e-2, atol=1e-3)
    torch.testing.assert_close(ys.cpu(), ref, rtol=1e-2, atol=2e-3)

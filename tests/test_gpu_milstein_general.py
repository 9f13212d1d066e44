"""General-noise Milstein (opt-in extension; the reference rejects it, milstein.py:25). No reference oracle exists,
so it is pinned by the reduction tests of SURVEY.md section 8, note N1."""
import math

import pytest
import torch
from scipy.stats import linregress
from torch import nn

from workloads import problems

pytestmark = pytest.mark.gpu
DEV = "cuda"
F64 = torch.float64


class DiagAsGeneral(nn.Module):
    """GBM written as a general-noise SDE: g[b, i, j] = delta_ij sigma_i y_i."""
    noise_type = "general"

    def __init__(self, base):
        super().__init__()
        self.base, self.sde_type = base, base.sde_type

    def f(self, t, y):
        return self.base.f(t, y)

    def g(self, t, y):
        return torch.diag_embed(self.base.g(t, y))


@pytest.mark.parametrize("sde_type", ["ito", "stratonovich"])
def test_reduces_to_diagonal_milstein(sde_type):
    """(i) diagonal g embedded as general == the reference-backed diagonal Milstein on the same increments."""
    import torchsde_amd
    B, d, steps, dt = 32, 4, 16, 2.0 ** -5
    base = problems.GBMDiag(d, sde_type, dtype=F64).to(DEV)
    y0 = torch.full((B, d), 0.1, dtype=F64, device=DEV)
    ts = torch.tensor([0.0, steps * dt], dtype=F64, device=DEV)
    kw = dict(t0=0.0, t1=steps * dt, size=(B, d), dtype=F64, device=DEV, entropy=5, dt=dt)
    with torch.no_grad():
        ref = torchsde_amd.sdeint(base, y0, ts, bm=torchsde_amd.BrownianInterval(**kw), method="milstein", dt=dt)
        gen = torchsde_amd.sdeint(DiagAsGeneral(base), y0, ts, bm=torchsde_amd.BrownianInterval(**kw),
                                  method="milstein", dt=dt, options={"general_noise": True})
    torch.testing.assert_close(gen, ref, rtol=1e-10, atol=1e-12)


def test_requires_opt_in():
    import torchsde_amd
    sde = problems.make("general_ito").to(DEV)
    with pytest.raises(ValueError, match="only supports noise types"):
        torchsde_amd.sdeint(sde, torch.zeros(2, 4, device=DEV), [0.0, 0.1], method="milstein", dt=0.05)


def test_levy_term_equals_dg_ga_jvp_column_sum():
    """(iii) one step: y1 - Euler step == dg_ga_jvp_column_sum(t0, y0, I) with I = (W W^T - dt Id)/2 + A."""
    import torchsde_amd
    from torchsde_amd.sde import ForwardSDE
    B, d, m, dt = 16, 4, 4, 2.0 ** -4
    sde = problems.make("general_ito", dtype=F64, d=d, m=m).to(DEV)
    y0 = torch.full((B, d), 0.1, dtype=F64, device=DEV)
    ts = torch.tensor([0.0, dt], dtype=F64, device=DEV)
    kw = dict(t0=0.0, t1=dt, size=(B, m), dtype=F64, device=DEV, entropy=9, dt=dt, levy_area_approximation="foster")
    with torch.no_grad():
        mil = torchsde_amd.sdeint(sde, y0, ts, bm=torchsde_amd.BrownianInterval(**kw), method="milstein", dt=dt,
                                  options={"general_noise": True})
        eul = torchsde_amd.sdeint(sde, y0, ts, bm=torchsde_amd.BrownianInterval(**kw), method="euler", dt=dt)
    W, _, A = torchsde_amd.BrownianInterval(**kw).increment_with_levy_area(0.0, dt)
    integrals = 0.5 * (W.unsqueeze(-1) * W.unsqueeze(-2) - dt * torch.eye(m, dtype=F64, device=DEV)) + A
    expected = ForwardSDE(sde).dg_ga_jvp_column_sum(ts[0], y0, integrals)
    torch.testing.assert_close(mil[-1] - eul[-1], expected.detach(), rtol=1e-9, atol=1e-12)


def test_strong_order_beats_euler():
    """(iv) on a commutative general-noise SDE (diagonal embedded) the slope is ~1.0 vs ~0.5 for Euler."""
    import torchsde_amd
    B, d, t1 = 4096, 4, 1.0
    base = problems.GBMDiag(d, "ito", dtype=F64).to(DEV)
    sde = DiagAsGeneral(base)
    y0 = torch.full((B, d), 0.1, dtype=F64, device=DEV)
    ts = torch.tensor([0.0, t1], dtype=F64, device=DEV)
    bm = torchsde_amd.BrownianInterval(0.0, t1, size=(B, d), dtype=F64, device=DEV, entropy=271, dt=2.0 ** -8)
    exact = base.exact(y0, t1, bm(0.0, t1))
    slopes = {}
    for method, opts in (("euler", None), ("milstein", {"general_noise": True})):
        xs, ys_ = [], []
        with torch.no_grad():
            for k in range(3, 9):
                dt = 2.0 ** -k
                out = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=dt, options=opts)
                xs.append(math.log(dt))
                ys_.append(0.5 * math.log(((out[-1] - exact) ** 2).sum(1).mean().item()))
        slopes[method] = linregress(xs, ys_).slope
    assert abs(slopes["milstein"] - 1.0) < 0.15 and slopes["euler"] < 0.75, slopes


# ---- the derivative-free form (options={"general_noise": True, "grad_free": True}) -----------------------------------
GF = {"general_noise": True, "grad_free": True}


class _ZeroDrift(nn.Module):
    """`base` with its drift removed (the supporting states of the Ito derivative-free scheme are y0 + dt*f + g_k*sqrt_dt:
    with f = 0 a diagonal g embedded as general is perturbed in channel k only, so the reduction is exact)."""

    def __init__(self, base):
        super().__init__()
        self.base, self.sde_type, self.noise_type = base, base.sde_type, base.noise_type

    def f(self, t, y):
        return torch.zeros_like(y)

    def g(self, t, y):
        return self.base.g(t, y)


@pytest.mark.parametrize("sde_type", ["ito", "stratonovich"])
@pytest.mark.parametrize("dtype,rtol", [(F64, 1e-9), (torch.float32, 2e-4)])
def test_grad_free_reduces_to_the_reference_backed_diagonal_grad_free_milstein(sde_type, dtype, rtol):
    """(i) diagonal g embedded as general, derivative-free == the diagonal derivative-free Milstein (milstein.py:58-67,
    pinned to reference goldens elsewhere) on the same increments. Stratonovich: any drift (its supporting state has no
    drift term); Ito: zero drift, see _ZeroDrift."""
    import torchsde_amd
    B, d, steps, dt = 32, 4, 16, 2.0 ** -5
    base = problems.GBMDiag(d, sde_type, dtype=dtype).to(DEV)
    if sde_type == "ito":
        base = _ZeroDrift(base)
    y0 = torch.full((B, d), 0.1, dtype=dtype, device=DEV)
    ts = torch.tensor([0.0, steps * dt], dtype=dtype, device=DEV)
    kw = dict(t0=0.0, t1=steps * dt, size=(B, d), dtype=dtype, device=DEV, entropy=5, dt=dt)
    with torch.no_grad():
        ref = torchsde_amd.sdeint(base, y0, ts, bm=torchsde_amd.BrownianInterval(**kw), method="milstein", dt=dt,
                                  options={"grad_free": True})
        gen = torchsde_amd.sdeint(DiagAsGeneral(base), y0, ts, bm=torchsde_amd.BrownianInterval(**kw),
                                  method="milstein", dt=dt, options=GF)
    torch.testing.assert_close(gen, ref, rtol=rtol, atol=rtol * 1e-2)


def test_grad_free_with_additive_diffusion_is_euler_bit_for_bit():
    """(ii) a state-independent g: every supporting evaluation returns g itself, the correction is exactly zero."""
    import torchsde_amd

    class AdditiveAsGeneral(nn.Module):
        noise_type, sde_type = "general", "ito"

        def __init__(self):
            super().__init__()
            self.sigma = nn.Parameter(torch.linspace(0.1, 0.6, 8 * 4).reshape(8, 4))

        def f(self, t, y):
            return -0.5 * y

        def g(self, t, y):
            return self.sigma.unsqueeze(0).repeat(y.shape[0], 1, 1)

    B, steps, dt = 64, 16, 2.0 ** -5
    sde = AdditiveAsGeneral().to(DEV)
    y0 = torch.full((B, 8), 0.1, device=DEV)
    ts = torch.tensor([0.0, steps * dt], device=DEV)
    kw = dict(t0=0.0, t1=steps * dt, size=(B, 4), dtype=torch.float32, device=DEV, entropy=5, dt=dt,
              levy_area_approximation="foster")
    with torch.no_grad():
        mil = torchsde_amd.sdeint(sde, y0, ts, bm=torchsde_amd.BrownianInterval(**kw), method="milstein", dt=dt, options=GF)
        eul = torchsde_amd.sdeint(sde, y0, ts, bm=torchsde_amd.BrownianInterval(**kw), method="euler", dt=dt)
    assert torch.equal(mil, eul)


@pytest.mark.parametrize("shape", [(16, 4, 4), (8, 32, 16), (5, 3, 2)])       # rows kernel (NC 1, 2) and the generic one
def test_grad_free_kernels_equal_their_torch_statement(shape):
    """The two kernels of the derivative-free step against the same sums written with torch ops in float64."""
    from torchsde_amd import kernels as K
    B, d, m = shape
    torch.manual_seed(3)
    dt, sqrt_dt = 2.0 ** -6, 2.0 ** -3
    for dtype, tol in ((F64, 1e-12), (torch.float32, 1e-5)):
        y0, f = torch.randn(B, d, dtype=dtype, device=DEV), torch.randn(B, d, dtype=dtype, device=DEV)
        g = torch.randn(B, d, m, dtype=dtype, device=DEV)
        gk = g.unsqueeze(0) + 0.1 * torch.randn(m, B, d, m, dtype=dtype, device=DEV)
        integrals = torch.randn(B, m, m, dtype=dtype, device=DEV) * dt
        for ito in (True, False):
            got = K.milstein_gf_general_support(y0, f, g, dt, sqrt_dt, ito)
            want = ((y0 + dt * f) if ito else y0).unsqueeze(0) + g.permute(2, 0, 1) * sqrt_dt
            assert torch.equal(got, want)
        got = K.milstein_gf_general_correction(g, gk, integrals, sqrt_dt)
        want = torch.einsum("kbil,bkl->bi", (gk - g.unsqueeze(0)).double(), integrals.double()) / sqrt_dt
        torch.testing.assert_close(got.double(), want, rtol=tol, atol=tol * want.abs().max().item())


def test_grad_free_levy_term_approximates_the_jvp_form():
    """(iii) one step on a smooth general-noise SDE: the derivative-free correction is a finite difference of the JVP
    form's (dg_ga_jvp_column_sum) with increment g_k*sqrt_dt (+ dt*f): they agree to O(sqrt_dt) relative."""
    import torchsde_amd
    B, d, m = 64, 4, 4
    sde = problems.make("general_ito", dtype=F64, d=d, m=m).to(DEV)
    y0 = torch.full((B, d), 0.1, dtype=F64, device=DEV)
    errs = []
    for dt in (2.0 ** -6, 2.0 ** -10):
        ts = torch.tensor([0.0, dt], dtype=F64, device=DEV)
        kw = dict(t0=0.0, t1=dt, size=(B, m), dtype=F64, device=DEV, entropy=9, dt=dt, levy_area_approximation="foster")
        with torch.no_grad():
            jvp = torchsde_amd.sdeint(sde, y0, ts, bm=torchsde_amd.BrownianInterval(**kw), method="milstein", dt=dt,
                                      options={"general_noise": True})
            gf = torchsde_amd.sdeint(sde, y0, ts, bm=torchsde_amd.BrownianInterval(**kw), method="milstein", dt=dt,
                                     options=GF)
            eul = torchsde_amd.sdeint(sde, y0, ts, bm=torchsde_amd.BrownianInterval(**kw), method="euler", dt=dt)
        term = (jvp[-1] - eul[-1]).norm().item()
        errs.append((gf[-1] - jvp[-1]).norm().item() / term)
    assert errs[0] < 0.6 and errs[1] < errs[0] / 2.5, errs       # 16x smaller dt -> ~4x smaller relative difference


def test_grad_free_strong_order_beats_euler():
    """(iv) strong order ~1.0 on the commutative general-noise SDE with drift (vs ~0.5 for Euler)."""
    import torchsde_amd
    B, d, t1 = 4096, 4, 1.0
    base = problems.GBMDiag(d, "ito", dtype=F64).to(DEV)
    sde = DiagAsGeneral(base)
    y0 = torch.full((B, d), 0.1, dtype=F64, device=DEV)
    ts = torch.tensor([0.0, t1], dtype=F64, device=DEV)
    bm = torchsde_amd.BrownianInterval(0.0, t1, size=(B, d), dtype=F64, device=DEV, entropy=271, dt=2.0 ** -8)
    exact = base.exact(y0, t1, bm(0.0, t1))
    xs, ys_ = [], []
    with torch.no_grad():
        for k in range(3, 9):
            dt = 2.0 ** -k
            out = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="milstein", dt=dt, options=GF)
            xs.append(math.log(dt))
            ys_.append(0.5 * math.log(((out[-1] - exact) ** 2).sum(1).mean().item()))
    assert abs(linregress(xs, ys_).slope - 1.0) < 0.15


def test_general_milstein_at_full_size_reduces_to_the_oracle_step():
    """configs[2] as worded, at its full shape (16384 x 32 x 16): a diagonal g embedded as general noise, JVP form and
    derivative-free form, sampled rows against the ORACLE's Milstein step (oracle/solvers_ref.py milstein_step <-
    milstein.py:52-94) on the same increments."""
    import torchsde_amd
    from oracle import solvers_ref
    B, d, m, steps, dt = 16384, 32, 16, 4, 2.0 ** -10

    class Embedded(nn.Module):
        """d = 32 state channels, m = 16 Brownian channels: channel j drives state channels j and j + 16."""
        noise_type, sde_type = "general", "ito"

        def __init__(self):
            super().__init__()
            gen = torch.Generator().manual_seed(0)
            self.mu = nn.Parameter(-torch.rand(d, generator=gen))
            self.sigma = nn.Parameter(0.2 + 0.5 * torch.rand(d, generator=gen))

        def f(self, t, y):
            return self.mu * y

        def g(self, t, y):
            s = self.sigma * y                                  # (B, d)
            out = y.new_zeros(y.shape[0], d, m)
            idx = torch.arange(d, device=y.device)
            out[:, idx, idx % m] = s
            return out

    sde = Embedded().to(DEV)
    y0 = torch.full((B, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, steps * dt], device=DEV)
    kw = dict(t0=0.0, t1=steps * dt, size=(B, m), dtype=torch.float32, device=DEV, entropy=77, dt=dt)
    bm = torchsde_amd.BrownianInterval(**kw)
    Ws = [bm(k * dt, (k + 1) * dt) for k in range(steps)]
    rows = torch.arange(0, B, 257, device=DEV)

    # oracle: the same SDE is DIAGONAL noise in the 32 channels with dW_i = W_{i mod 16} (commutative: A drops out)
    class AsDiagonal(nn.Module):
        noise_type, sde_type = "diagonal", "ito"

        def f(self, t, y):
            return sde.mu.detach().cpu().double() * y

        def g(self, t, y):
            return sde.sigma.detach().cpu().double() * y

    y = y0[rows].cpu().double()
    t = torch.tensor(0.0, dtype=F64)
    for k in range(steps):
        Wk = Ws[k][rows].cpu().double()
        y = solvers_ref.milstein_step(AsDiagonal(), lambda a, b, W=Wk: torch.cat([W, W], dim=1), t, t + dt, y)
        t = t + dt
    with torch.no_grad():
        jvp = torchsde_amd.sdeint(sde, y0, ts, bm=torchsde_amd.BrownianInterval(**kw), method="milstein", dt=dt,
                                  options={"general_noise": True})[-1][rows].cpu().double()
        gf = torchsde_amd.sdeint(sde, y0, ts, bm=torchsde_amd.BrownianInterval(**kw), method="milstein", dt=dt,
                                 options=GF)[-1][rows].cpu().double()
    torch.testing.assert_close(jvp, y, rtol=2e-5, atol=1e-7)
    # the derivative-free form differs from the derivative form by its O(dt^1.5) finite-difference error per step
    torch.testing.assert_close(gf, y, rtol=1e-3, atol=1e-6)
    assert (gf - y).abs().max() > 0


@pytest.mark.parametrize("dtype", [torch.float32, F64])
@pytest.mark.parametrize("shape,row_offset", [((64, 16), 0), ((33, 4), 0), ((20, 6), 2), ((16, 16), 7), ((9, 3), 0)])
@pytest.mark.parametrize("levy", ["davie", "foster"])
def test_levy_area_kernels_equal_their_definition_and_the_fused_integrals(dtype, shape, row_offset, levy):
    """The row-per-wave Levy-area kernel (one Philox call per four entries) and its fused form I = (W W^T - dt Id)/2 + A
    against the definition written with torch ops on the generator's raw normals (tsde_brownian_normals): the same
    bits as the per-entry kernel it replaces (odd m and unaligned shards still take that one)."""
    import torchsde_amd
    from torchsde_amd import _native, kernels as K
    B, m = shape
    h = 2.0 ** -5
    bm = torchsde_amd.BrownianInterval(0.0, 1.0, size=(B, m), dtype=dtype, device=DEV, entropy=1234, dt=h,
                                       levy_area_approximation=levy, row_offset=row_offset)
    W, U, A = bm.increment_with_levy_area(3 * h, 4 * h)
    H = torch.empty(B, m, dtype=dtype, device=DEV)
    bm.increment(3 * h, 4 * h, want_U=True, out_H=H)
    # the node key of the interval, as increment_with_levy_area forms it
    import struct
    bits_a, bits_b = (struct.unpack("<Q", struct.pack("<d", x))[0] for x in (bm._round(3 * h), bm._round(4 * h)))
    mix = (bits_a * 0x9E3779B97F4A7C15 + ((bits_b << 31) | (bits_b >> 33)) * 0xBF58476D1CE4E5B9) & ((1 << 64) - 1)
    node = (mix ^ (mix >> 29)) & ((1 << 42) - 1)
    ca, _ = bm.locate(bm._round(3 * h), bm._round(4 * h))
    N = torch.empty(B * m * m, dtype=dtype, device=DEV)
    lib = _native.load()
    _native.check(lib.tsde_brownian_normals(_native.ptr(N), N.numel(), bm._key, bm._elem0 * m, ca, node, 2,
                                            _native.dtype_code(dtype), _native.stream_ptr(N.device)), "normals")
    N = N.reshape(B, m, m)
    Wi, Wj, Hi, Hj = W.unsqueeze(2), W.unsqueeze(1), H.unsqueeze(2), H.unsqueeze(1)
    if levy == "foster":
        tenth = torch.tensor(0.1 * h, dtype=dtype, device=DEV)
        sd = (tenth * (tenth + (Hi * Hi + Hj * Hj))).double().sqrt().to(dtype)
    else:
        sd = torch.full((B, m, m), (h * h / 12.0) ** 0.5, dtype=dtype, device=DEV)
    want = (Hi * Wj - Wi * Hj) + sd * (N - N.transpose(1, 2))
    eye = torch.eye(m, dtype=torch.bool, device=DEV)
    want = torch.where(eye, Hi * Wj - Wi * Hj, want)
    assert torch.equal(A, want)
    assert torch.equal(A, -A.transpose(1, 2))
    for ito in (True, False):
        _, _, integrals = bm.increment_with_levy_area(3 * h, 4 * h, iterated=(h, ito))
        assert torch.equal(integrals, K.iterated_integrals(W, A, h, ito))

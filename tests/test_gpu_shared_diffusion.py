"""``tsde_step_shared``: the step for a diffusion that is ONE (d, m) matrix for the whole batch, as one product on the
matrix cores with the increments generated in registers (csrc/steps.hip shared_mfma_kernel; ``-m gpu``).

Checked against float64 torch arithmetic on the SAME increments (the counter generator materialises them through
``tsde_cell_increment``), for every tile shape class, both dtypes, the three weight modes of SRA1, external increments,
and -- bit for bit -- against itself under row sharding and against the per-row contraction kernel's results through
the solvers (tolerance: the two sum over the Brownian channels in different orders)."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _spec(B, m, dtype, cell=3, h=2.0 ** -7, elem0=0):
    from torchsde_amd.kernels import NoiseSpec
    return NoiseSpec((B, m), dtype, torch.device(DEV), entropy=0x1234567, elem0=elem0, cell=cell, h=h)


def _reference(y0, f, S, ca, cf, cg, mode, cw, cu, rdt, W, U):
    y0, f, S, W = (x.double() for x in (y0, f, S, W))
    w = W if mode == 0 else ((cu * U.double()) * rdt if mode == 1 else (cw * W) + (cu * U.double()) * rdt)
    return (y0 + (ca * f) * cf) + cg * (w @ S.t())


SHAPES = [(16384, 32, 16), (1000, 3, 4), (33, 20, 8), (4096, 128, 64), (257, 48, 12), (5000, 64, 32), (17, 16, 4),
          (300, 6, 2), (64, 130, 8), (100, 5, 3)]            # the last three: outside the tiles (generic kernel, S shared)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("B,d,m", SHAPES)
def test_step_shared_against_float64_arithmetic_on_the_same_increments(B, d, m, dtype):
    from torchsde_amd import kernels as K
    gen = torch.Generator().manual_seed(B + d + m)
    y0 = torch.randn(B, d, generator=gen).to(DEV, dtype)
    f = torch.randn(B, d, generator=gen).to(DEV, dtype)
    S = (torch.randn(d, m, generator=gen) / m ** 0.5).to(DEV, dtype)
    tol = dict(rtol=2e-5, atol=2e-5) if dtype == torch.float32 else dict(rtol=1e-12, atol=1e-12)
    dt = 2.0 ** -7
    for mode, (ca, cw, cu) in ((0, (1.0, 0.0, 0.0)), (1, (0.75, 0.0, 1.5)), (2, (1 / 3, 1.0, -1.0))):
        spec = _spec(B, m, dtype)
        W, U = spec.materialise(need_U=True)
        got = K._raw_step_shared(y0, f, S, ca, dt, 0.5, mode, cw, cu, 1.0 / dt, spec, None)
        want = _reference(y0, f, S, ca, dt, 0.5, mode, cw, cu, 1.0 / dt, W, U)
        torch.testing.assert_close(got.double(), want, **tol)
        # the same launch on increments handed over as tensors
        ext = K.NoiseSpec.external(W, U)
        torch.testing.assert_close(K._raw_step_shared(y0, f, S, ca, dt, 0.5, mode, cw, cu, 1.0 / dt, ext, None).double(),
                                   want, **tol)
    # in place into a caller's buffer, and an unaligned view of one (the scalar path of the epilogue)
    buf = torch.empty(B * d + 1, device=DEV, dtype=dtype)
    spec = _spec(B, m, dtype)
    W, _ = spec.materialise()
    out = K._raw_step_shared(y0, f, S, 1.0, dt, 1.0, 0, 0.0, 0.0, 0.0, spec, buf[1:].view(B, d))
    torch.testing.assert_close(out.double(), _reference(y0, f, S, 1.0, dt, 1.0, 0, 0, 0, 0, W, None), **tol)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
def test_rows_solved_alone_are_the_rows_of_the_full_launch(dtype):
    """Row sharding moves a row to another lane and another tile; its sum over the channels keeps its order."""
    from torchsde_amd import kernels as K
    B, d, m = 4096, 32, 16
    y0 = torch.randn(B, d, device=DEV, dtype=dtype)
    f = torch.randn(B, d, device=DEV, dtype=dtype)
    S = torch.randn(d, m, device=DEV, dtype=dtype)
    full = K._raw_step_shared(y0, f, S, 1.0, 0.01, 1.0, 0, 0.0, 0.0, 0.0, _spec(B, m, dtype), None)
    for lo, n in ((0, 16), (1028, 333), (4000, 96)):
        part = K._raw_step_shared(y0[lo:lo + n].contiguous(), f[lo:lo + n].contiguous(), S, 1.0, 0.01, 1.0, 0, 0.0, 0.0,
                                  0.0, _spec(n, m, dtype, elem0=lo * m), None)
        assert torch.equal(part, full[lo:lo + n])


class _Additive(nn.Module):
    """Additive noise as the reference's examples return it: one matrix, expanded over the batch."""
    noise_type, sde_type = "additive", "ito"

    def __init__(self, d, m):
        super().__init__()
        gen = torch.Generator().manual_seed(11)
        self.A = nn.Parameter(-torch.rand(d, generator=gen))
        self.sigma = nn.Parameter(0.3 * torch.randn(d, m, generator=gen))

    def f(self, t, y):
        return self.A * y + torch.sin(t)

    def g(self, t, y):
        return self.sigma.expand(y.shape[0], -1, -1)


@pytest.mark.parametrize("method,levy", [("euler", "none"), ("milstein", "none"), ("srk", "space-time")])
def test_solvers_on_a_batch_broadcast_diffusion_match_the_per_row_contraction(method, levy):
    """`sigma.expand(B, d, m)` takes the matrix-core kernel; the same numbers materialised per row (`.contiguous()`)
    take the per-row contraction kernel (tsde_step_general): same increments, sums in another order."""
    import torchsde_amd
    B, d, m, steps, dt = 2048, 32, 16, 24, 2.0 ** -6
    sde = _Additive(d, m).to(DEV)

    class PerRow(_Additive):
        def g(self, t, y):
            return self.sigma.expand(y.shape[0], -1, -1).contiguous()

    per_row = PerRow(d, m).to(DEV)
    y0 = torch.full((B, d), 0.2, device=DEV)
    ts = torch.tensor([0.0, 7 * dt, steps * dt], device=DEV)

    def solve(module):
        bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(B, m), device=DEV, entropy=5,
                                           levy_area_approximation=levy)
        from torchsde_amd import kernels as K
        K.prof_begin(12, 4 * steps + 8)          # TSDE_KID_STEP_SHARED
        with torch.no_grad():
            out = torchsde_amd.sdeint(module, y0, ts, bm=bm, method=method, dt=dt, options={"hip_graph": False})
        torch.cuda.synchronize()
        return out, K.prof_end()[1]

    shared, launches = solve(sde)
    rows, none = solve(per_row)
    assert launches >= steps and none == 0
    torch.testing.assert_close(shared, rows, rtol=1e-5, atol=1e-6)

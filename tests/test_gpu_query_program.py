"""The Brownian query kernel in its two forms -- every lane walking the bridge tree itself (csrc/tsde_bridge.h, the
original), and the walk done once per block and run by the lanes as a list of operations (csrc/tsde_query_program.h, the
default since round 3) -- must give the same bits for every kind of query: the second is the first with its
wave-uniform bookkeeping taken out of the lanes, nothing else. (tests/golden/query_kernel_r1.pt pins the values
themselves.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _both(bm, ta, tb, want_U):
    from torchsde_amd import _native
    lib = _native.load()
    out = []
    try:
        for legacy in (1, 0):
            lib.tsde_set_query_walk(legacy)
            H = torch.full(bm.shape, float("nan"), dtype=bm.dtype, device=DEV) if want_U else None
            W, U = bm.increment(ta, tb, want_U=want_U, out_H=H)
            out.append((W.clone(), None if U is None else U.clone(), H))
    finally:
        lib.tsde_set_query_walk(0)
    return out


QUERIES = [(0.0, 1.0), (0.25, 0.5), (5 / 64, 6 / 64), (5 / 64, 5.5 / 64), (5.3 / 64 + 1e-7, 6.3 / 64 + 1e-7),
           (5.1 / 64 + 1e-7, 5.7 / 64), (0.1, 0.9), (0.0, 0.3337), (0.61803, 1.0), (0.5, 0.5 + 1e-9), (1 / 3, 2 / 3),
           (5 / 64, 5 / 64 + 2.0 ** -40), (0.123456789, 0.123456789 + 1e-6), (3 / 64 + 1e-12, 40 / 64 - 1e-12)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64])
@pytest.mark.parametrize("levy", ["none", "space-time"])
@pytest.mark.parametrize("grid", ["dt", "one cell", "halfway tree", "tol"])
def test_program_and_walk_give_the_same_bits(dtype, levy, grid):
    import torchsde_amd
    kw = dict(t0=0.0, t1=1.0, size=(37, 6), dtype=dtype, device=DEV, entropy=4242, levy_area_approximation=levy,
              row_offset=3)          # 222 elements, first element at 18: neither a multiple of 4 (scalar stores)
    if grid == "dt":
        kw["dt"] = 1 / 64
    elif grid == "halfway tree":
        kw.update(halfway_tree=True, tol=1e-4)
    elif grid == "tol":
        kw.update(dt=1 / 64, tol=1e-5)
    bm = torchsde_amd.BrownianInterval(**kw)
    for ta, tb in QUERIES:
        walk, program = _both(bm, ta, tb, want_U=levy != "none")
        for a, b, name in zip(walk, program, "WUH"):
            if a is not None:
                assert torch.equal(a, b), (name, ta, tb)
                assert torch.isfinite(b).all(), (name, ta, tb)


@pytest.mark.parametrize("levy", ["none", "space-time"])
def test_program_and_walk_agree_on_vector_stores_and_pinned_roots(levy):
    import torchsde_amd
    torch.manual_seed(0)
    shape = (64, 16)
    bm = torchsde_amd.BrownianInterval(0.0, 1.0, size=shape, dtype=torch.float32, device=DEV, entropy=7, dt=1 / 32,
                                       levy_area_approximation=levy)
    pinned = torchsde_amd.BrownianInterval(0.0, 1.0, dtype=torch.float32, device=DEV, entropy=7,
                                           levy_area_approximation=levy, W=torch.randn(shape, device=DEV),
                                           H=torch.randn(shape, device=DEV) * 0.3 if levy != "none" else None)
    for which in (bm, pinned):
        for ta, tb in QUERIES:
            walk, program = _both(which, ta, tb, want_U=levy != "none")
            for a, b in zip(walk, program):
                if a is not None:
                    assert torch.equal(a, b), (ta, tb)


def test_adaptive_solve_is_unchanged_by_the_query_form():
    """The device-controlled adaptive loop reads its query bounds from device memory (tsde_brownian_query_dev): the
    program is then built in the kernel from those; a whole adaptive solve must not move by a bit."""
    import torchsde_amd
    from torchsde_amd import _native
    from workloads import problems
    lib = _native.load()
    sde = problems.make("gbm_ito", d=8).to(DEV)
    y0 = torch.full((128, 8), 0.1, device=DEV)
    ts = torch.tensor([0.0, 0.1, 0.25], device=DEV)
    outs = []
    try:
        for legacy in (1, 0):
            lib.tsde_set_query_walk(legacy)
            bm = torchsde_amd.BrownianInterval(0.0, 0.25, size=(128, 8), device=DEV, dtype=torch.float32, entropy=11,
                                               levy_area_approximation="space-time")
            with torch.no_grad():
                outs.append(torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="srk", dt=0.01, adaptive=True, rtol=1e-3,
                                                atol=1e-4, options={"hip_graph": False}))
    finally:
        lib.tsde_set_query_walk(0)
    assert torch.equal(outs[0], outs[1])

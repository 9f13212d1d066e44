"""Randomised end-to-end parity (run with ``-m gpu``): random batch / state / channel sizes (odd ones included: they
take the scalar, unaligned and generic kernel paths), random output times and step sizes, every method that fits the
drawn noise type -- `sdeint` on the GPU vs the oracle's restatement of the reference's solver on the C twin of the
generator; float64, fixed seed."""
import numpy as np
import pytest
import torch

from oracle import counter, solvers_ref
from workloads import problems
from torchsde_amd import timegrid

pytestmark = pytest.mark.gpu
DEV = "cuda"

METHODS = {
    ("gbm", "ito"): ["euler", "milstein", "srk"], ("gbm", "strat"): ["midpoint", "heun", "euler_heun", "milstein"],
    ("scalar", "ito"): ["euler", "milstein", "srk"], ("scalar", "strat"): ["midpoint", "heun", "euler_heun"],
    ("additive", "ito"): ["euler", "milstein", "srk"], ("additive", "strat"): ["midpoint", "heun"],
    ("general", "ito"): ["euler"], ("general", "strat"): ["midpoint", "heun", "euler_heun"],
}


def _cases(n=48, seed=20240917):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        kind = ["gbm", "scalar", "additive", "general"][i % 4]
        tag = ["ito", "strat"][int(rng.integers(2))]
        method = METHODS[(kind, tag)][int(rng.integers(len(METHODS[(kind, tag)])))]
        B, d = int(rng.integers(1, 41)), int(rng.integers(1, 10))
        m = {"gbm": d, "scalar": 1}.get(kind, int(rng.integers(1, 7)))
        n_ts = int(rng.integers(2, 5))
        ts = np.cumsum(np.concatenate([[0.0], rng.uniform(0.05, 0.4, size=n_ts - 1)]))
        dt = float(rng.uniform(0.02, 0.15))
        out.append((f"{i:02d}-{kind}_{tag}-{method}-B{B}-d{d}-m{m}", kind, tag, method, B, d, m, ts, dt, int(rng.integers(1 << 30))))
    return out


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c[0])
def test_random_configuration_matches_oracle(case):
    import torchsde_amd
    _, kind, tag, method, B, d, m, ts_np, dt, entropy = case
    dtype = torch.float64
    levy = method == "srk"
    y0 = torch.linspace(0.05, 0.3, B * d, dtype=dtype).reshape(B, d)
    ts = torch.tensor(ts_np, dtype=dtype)
    edges = timegrid.build(ts_np.astype(np.float64), dt).t_f64()      # the grid the generator adopts from the solver

    def bm_cpu(ta, tb, return_U=False):
        W, U, _ = counter.query(B * m, entropy, edges, float(ta), float(tb), dtype=np.float64, have_h=levy)
        W = torch.from_numpy(W).reshape(B, m)
        return (W, torch.from_numpy(U).reshape(B, m)) if return_U else W

    sde_cpu = problems.make(f"{kind}_{tag}", dtype=dtype, d=d, m=m)
    with torch.no_grad():
        ref = solvers_ref.integrate(sde_cpu, bm_cpu, y0, ts, dt, method)
    sde = problems.make(f"{kind}_{tag}", dtype=dtype, d=d, m=m).to(DEV)
    bm = torchsde_amd.BrownianInterval(float(ts_np[0]), float(ts_np[-1]), size=(B, m), dtype=dtype, device=DEV,
                                       entropy=entropy, levy_area_approximation="space-time" if levy else "none")
    with torch.no_grad():
        got = torchsde_amd.sdeint(sde, y0.to(DEV), ts.to(DEV), bm=bm, method=method, dt=dt)
    torch.testing.assert_close(got.cpu(), ref, rtol=1e-9, atol=1e-11)


def _adjoint_cases(n=16, seed=77):
    rng = np.random.default_rng(seed)
    combos = [("gbm", "ito", "euler", "euler"), ("gbm", "ito", "milstein", None), ("gbm", "strat", "midpoint", None),
              ("mlpdiag", "ito", "srk", None), ("general", "ito", "euler", None), ("general", "strat", "midpoint", None),
              ("additive", "ito", "euler", None), ("scalar", "ito", "euler", None)]
    out = []
    for i in range(n):
        kind, tag, method, adjoint_method = combos[i % len(combos)]
        B, d = int(rng.integers(2, 25)), int(rng.integers(1, 7))
        m = {"gbm": d, "mlpdiag": d, "scalar": 1}.get(kind, int(rng.integers(1, 5)))
        n_ts = int(rng.integers(2, 4))
        ts = np.cumsum(np.concatenate([[0.0], rng.uniform(0.1, 0.3, size=n_ts - 1)]))
        dt = float(rng.uniform(0.03, 0.1))
        out.append((f"{i:02d}-{kind}_{tag}-{method}-B{B}-d{d}-m{m}", kind, tag, method, adjoint_method, B, d, m, ts, dt,
                    int(rng.integers(1 << 30))))
    return out


@pytest.mark.parametrize("case", _adjoint_cases(), ids=lambda c: c[0])
def test_random_adjoint_matches_oracle(case):
    """sdeint_adjoint on random shapes and NON-dyadic step sizes (the reverse sweep's steps then straddle the
    generator's cells: every reverse increment is a bridge query) vs the oracle's restatement of the reference's
    adjoint on the C twin of the generator."""
    import torchsde_amd
    from oracle import adjoint_ref
    _, kind, tag, method, adjoint_method, B, d, m, ts_np, dt, entropy = case
    dtype = torch.float64
    levy = method == "srk"
    ts = torch.tensor(ts_np, dtype=dtype)
    edges = timegrid.build(ts_np.astype(np.float64), dt).t_f64()
    wt = torch.linspace(-1, 1, len(ts_np) * B * d, dtype=dtype).reshape(len(ts_np), B, d)

    def bm_cpu(ta, tb, return_U=False):
        W, U, _ = counter.query(B * m, entropy, edges, float(ta), float(tb), dtype=np.float64, have_h=levy)
        W = torch.from_numpy(W).reshape(B, m)
        return (W, torch.from_numpy(U).reshape(B, m)) if return_U else W

    sde_cpu = problems.make(f"{kind}_{tag}", dtype=dtype, d=d, m=m)
    ys_ref, gy_ref, gp_ref = adjoint_ref.adjoint_gradients(sde_cpu, torch.full((B, d), 0.1, dtype=dtype), ts, bm_cpu, dt,
                                                           method, adjoint_method, wt)
    sde = problems.make(f"{kind}_{tag}", dtype=dtype, d=d, m=m).to(DEV)
    y0 = torch.full((B, d), 0.1, dtype=dtype, device=DEV, requires_grad=True)
    bm = torchsde_amd.BrownianInterval(float(ts_np[0]), float(ts_np[-1]), size=(B, m), dtype=dtype, device=DEV,
                                       entropy=entropy, levy_area_approximation="space-time" if levy else "none")
    ys = torchsde_amd.sdeint_adjoint(sde, y0, ts.to(DEV), bm=bm, method=method, adjoint_method=adjoint_method, dt=dt)
    (ys * wt.to(DEV)).sum().backward()
    torch.testing.assert_close(ys.detach().cpu(), ys_ref, rtol=1e-9, atol=1e-11)
    torch.testing.assert_close(y0.grad.cpu(), gy_ref, rtol=1e-7, atol=1e-9)
    for p, ref in zip(sde.parameters(), gp_ref):
        got = torch.zeros_like(ref) if p.grad is None else p.grad.cpu()
        torch.testing.assert_close(got, ref, rtol=1e-7, atol=1e-8)

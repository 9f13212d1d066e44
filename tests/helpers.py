"""Shared test helpers: golden fixture loading and replay Brownian motions."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TORCH_DT = {"f32": torch.float32, "f64": torch.float64}


def load(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def solver_cases():
    return sorted(f[len("solver_"):-4] for f in os.listdir(GOLDEN) if f.startswith("solver_") and f.endswith(".npz"))


def adjoint_cases():
    return sorted(f[len("adjoint_"):-4] for f in os.listdir(GOLDEN)
                  if f.startswith("adjoint_") and f.endswith(".npz") and not f.startswith("adjoint_adaptive_"))


class Case:
    """One golden solver case in one precision."""

    def __init__(self, name, tag, prefix="solver_"):
        z = load(f"{prefix}{name}.npz")
        self.z = z
        self.name, self.tag, self.dtype = name, tag, TORCH_DT[tag]
        self.problem, self.method, self.levy = str(z["problem"]), str(z["method"]), str(z["levy"])
        self.dt = float(z["dt"])
        self.options = {"grad_free": True} if bool(z["grad_free"]) else None
        self.B, self.d, self.m = (int(v) for v in z["shape"])
        self.ts = torch.tensor(z[f"{tag}__ts"], dtype=self.dtype)
        self.queries = z[f"{tag}__queries"]
        self.W = z[f"{tag}__W"]
        self.U = z[f"{tag}__U"]
        self.A = z[f"{tag}__A"] if f"{tag}__A" in z.files else None
        self.ys = torch.tensor(z[f"{tag}__ys"], dtype=self.dtype)
        self.param_checksum = float(z[f"{tag}__param_checksum"])

    def sde(self, device="cpu"):
        from workloads import problems
        sde = problems.make(self.problem, dtype=self.dtype, d=self.d, m=self.m)
        got = float(sum(p.detach().double().abs().sum() for p in sde.parameters()))
        assert abs(got - self.param_checksum) <= 1e-9 * max(1.0, abs(got)), "test problem parameters drifted"
        return sde.to(device)

    def y0(self, device="cpu"):
        return torch.full((self.B, self.d), 0.1, dtype=self.dtype, device=device)

    def table(self, device="cpu"):
        return {(float(a), float(b)): (torch.tensor(self.W[i], dtype=self.dtype, device=device),
                                       torch.tensor(self.U[i], dtype=self.dtype, device=device),
                                       None if self.A is None else torch.tensor(self.A[i], dtype=self.dtype,
                                                                                device=device))
                for i, (a, b) in enumerate(self.queries)}


def make_replay_bm(table, shape, dtype, device, levy):
    """A foreign BaseBrownian (seam S3) that replays stored increments. Intervals are matched to 1e-10: an adaptive
    solve proposes its step sizes from an error norm whose last bits depend on the reduction order, so its query
    times agree with the recorded ones to rounding, not bit for bit."""
    from torchsde_amd import BaseBrownian
    _dtype, _device, _shape, _levy = dtype, device, shape, levy
    nearby = {}
    for (a, b), v in table.items():
        nearby.setdefault((round(a, 10), round(b, 10)), v)

    class Replay(BaseBrownian):
        def __call__(self, ta, tb=None, return_U=False, return_A=False):
            key = (float(ta), float(tb))
            W, U, A = table[key] if key in table else nearby[(round(key[0], 10), round(key[1], 10))]
            if return_U:
                return (W, U, A) if return_A else (W, U)
            return (W, A) if return_A else W

        def __repr__(self):
            return "Replay"

        @property
        def dtype(self):
            return _dtype

        @property
        def device(self):
            return torch.device(_device)

        @property
        def shape(self):
            return tuple(_shape)

        @property
        def levy_area_approximation(self):
            return _levy

    return Replay()


def has_gpu():
    return torch.cuda.is_available()


def mlp_module_from(z, dtype, device):
    """The perceptron-drift module of a closed_form_mlp_*.npz fixture, with the fixture's parameter values."""
    import torchsde_amd
    B, d, hidden, steps = (int(v) for v in z["shape"])
    sde = torchsde_amd.MLPDriftDiagonalSDE(d, hidden, activation=str(z["activation"]), sde_type=str(z["sde_type"]),
                                           diffusion=str(z["diffusion"]), diff_scale=float(z["diff_scale"]),
                                           diff_rate=torch.tensor(z["param__diff_rate"]),
                                           diff_shift=torch.tensor(z["param__diff_shift"]), dtype=dtype)
    with torch.no_grad():
        for name, p in sde.named_parameters():
            p.copy_(torch.tensor(z["param__" + name]).to(dtype))
    return sde.to(device)


def sampled_rows(B, n=64, seed=0, seams=()):
    """~n global rows of a B-row batch: the first and last rows, both sides of every kernel / shard seam handed in
    (`seams`), both sides of the 256-row tile boundaries next to them, and random rows in between."""
    rows = {0, 1, B - 2, B - 1}
    for s in tuple(seams) + (B // 2, 256, 512, B - 256):
        rows.update(r for r in (s - 1, s, s + 1) if 0 <= r < B)
    gen = np.random.default_rng(seed)
    while len(rows) < n:
        rows.add(int(gen.integers(0, B)))
    return np.array(sorted(rows), dtype=np.int64)


def counter_rows_bm(rows, m, entropy, edges, dtype, levy=False):
    """The counter-RNG Brownian path of GLOBAL batch rows `rows` (m channels each) from the oracle's C twin of the
    generator, as the callable the oracle's solvers take: ``bm(ta, tb, return_U=False) -> (len(rows), m)`` tensors.
    Row r of an unsharded (B, m) BrownianInterval is elements r*m .. r*m + m - 1 of the counter field."""
    from oracle import counter
    npdt = np.float32 if dtype == torch.float32 else np.float64
    edges = np.ascontiguousarray(edges, dtype=np.float64)

    def bm(ta, tb, return_U=False, return_A=False):
        W = np.empty((len(rows), m), dtype=npdt)
        U = np.empty((len(rows), m), dtype=npdt) if levy else None
        for k, r in enumerate(rows):
            w, u, _ = counter.query(m, entropy, edges, float(ta), float(tb), dtype=npdt, elem0=int(r) * m, have_h=levy)
            W[k] = w
            if levy:
                U[k] = u
        W = torch.from_numpy(W)
        return (W, torch.from_numpy(U)) if return_U else W

    return bm


def assert_within_reference_rounding(new32, ref32, ref64, what="", factor=4.0, floor=1e-6):
    """the float32 HIP result may differ from the float64 oracle by at most `factor` times what
    the oracle's own float32 run differs from it, plus `floor` (scaled by the magnitude of the compared quantity)."""
    new32, ref32, ref64 = (torch.as_tensor(x).detach().double().cpu() for x in (new32, ref32, ref64))
    scale = max(1.0, ref64.abs().max().item())
    err_new = (new32 - ref64).abs().max().item()
    err_ref = (ref32 - ref64).abs().max().item()
    assert err_new <= factor * err_ref + floor * scale, \
        f"{what}: |hip32 - ref64| = {err_new:.3e} > {factor} * |ref32 - ref64| ({err_ref:.3e}) + {floor * scale:.1e}"
    return err_new, err_ref

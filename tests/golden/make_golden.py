"""Generates the golden fixtures in this directory by running the REAL reference (read-only checkout at
/root/reference, CPU) -- run in the dev container only:

    python tests/golden/make_golden.py

The GPU box has no /root/reference; tests read the committed .npz files. Fixtures:
  timegrid.npz        the (t0, t1) pairs the reference's stepping loop queries for several (ts, dt, dtype)
  solver_<case>.npz   reference `sdeint` outputs under replayed Brownian increments (increments stored)
  adjoint_<case>.npz  reference `sdeint_adjoint` outputs + gradients under replayed increments
  bridge.npz          (parent, normals, children) tuples recorded inside the reference's bridge code and
                      multi-interval merge results
  brownian_seq.npz    reference BrownianInterval outputs for fixed entropy and query sequences
  closed_form_mlp_<case>.npz   reference `sdeint` (+ autograd gradients) of the perceptron-drift module in float64, on
                      the counter-RNG path the trajectory kernels generate for themselves
  closed_form_affine_<case>.npz   the same for the affine diagonal module (all five schemes of its trajectory kernel)
  recognised_poly3_<case>.npz   reference `sdeint` of plain user modules with polynomial drift / diffusion (double well,
                                logistic growth), float64, counter-RNG path: pins TSDE_FN_POLY3 to the reference
  recognised_additive_<case>.npz   reference `sdeint` of plain additive-noise modules (the shapes of ExAdditive, a constant
                                matrix, NeuralAdditive), float64, counter-RNG path: pins the additive trajectory kernels
  closed_form_expr_<case>.npz   reference `sdeint` of the elementwise-expression module (incl. the reference's own
                      benchmark SDE f = y, g = exp(-y)), float64, counter-RNG path
  closed_form_adjoint_<case>.npz   reference `sdeint_adjoint(adjoint_method="euler")` of the perceptron-drift module, float64,
                      counter-RNG path: ys and all gradients
  logqp_<case>.npz    reference `sdeint(..., logqp=True)` / `sdeint_adjoint(..., logqp=True[, names=])`: ys, log-ratio, gradients
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "_ref_shim"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import torchsde  # noqa: E402  (the reference)
from torchsde._brownian import brownian_interval as ref_bi  # noqa: E402

from workloads import problems  # noqa: E402

DT = {"f32": torch.float32, "f64": torch.float64}


class ReplayBM(torchsde.BaseBrownian):
    """Draws an increment the first time an interval is queried and replays it afterwards."""

    def __init__(self, shape, dtype, seed, levy="none"):
        super().__init__()
        self._shape, self._dtype, self._levy = tuple(shape), dtype, levy
        self.rng = np.random.default_rng(seed)
        self.table, self.order = {}, []

    def __call__(self, ta, tb=None, return_U=False, return_A=False):
        key = (float(ta), float(tb))
        if key not in self.table:
            h = key[1] - key[0]
            W = torch.tensor(self.rng.standard_normal(self._shape) * np.sqrt(h), dtype=self._dtype)
            H = torch.tensor(self.rng.standard_normal(self._shape) * np.sqrt(h / 12), dtype=self._dtype)
            U = h * (.5 * W + H)
            A = None
            if self._levy in ("davie", "foster"):
                N = torch.tensor(self.rng.standard_normal(self._shape + self._shape[-1:]) * h, dtype=self._dtype)
                A = 0.5 * (N - N.transpose(-1, -2))
            self.table[key] = (W, U, A)
            self.order.append(key)
        W, U, A = self.table[key]
        if return_U:
            return (W, U, A) if return_A else (W, U)
        return (W, A) if return_A else W

    def __repr__(self):
        return "ReplayBM"

    dtype = property(lambda self: self._dtype)
    device = property(lambda self: torch.device("cpu"))
    shape = property(lambda self: self._shape)
    levy_area_approximation = property(lambda self: self._levy)

    def dump(self):
        keys = np.array(self.order, dtype=np.float64).reshape(-1, 2)
        W = np.stack([self.table[k][0].numpy() for k in self.order])
        U = np.stack([self.table[k][1].numpy() for k in self.order])
        self.A_dump = None
        if len(self.table[self.order[0]]) > 2 and self.table[self.order[0]][2] is not None:
            self.A_dump = np.stack([self.table[k][2].numpy() for k in self.order])
        return keys, W, U


def param_checksum(sde):
    return float(sum(p.detach().double().abs().sum() for p in sde.parameters()))


# ------------------------------------------------------------------------------------------------- timegrid
def gen_timegrid():
    class Rec(ReplayBM):
        pass
    out = {}
    cases = {
        "f32_1e-3": (torch.tensor([0., 1.], dtype=torch.float32), 1e-3),
        "f32_dyadic": (torch.tensor([0., 1000 * 2.0 ** -10], dtype=torch.float32), 2.0 ** -10),
        "f32_multi": (torch.tensor([0., 0.25, 0.5, 0.77], dtype=torch.float32), 0.1),
        "f64_multi": (torch.tensor([0., 0.25, 0.5, 0.77], dtype=torch.float64), 0.1),
        "f32_linspace20": (torch.linspace(0, 1, 20, dtype=torch.float32), 1e-3),
        "f64_1e-2": (torch.tensor([0.3, 1.7], dtype=torch.float64), 1e-2),
        "f32_coarse": (torch.tensor([0., 0.05, 0.1, 0.15, 1.0], dtype=torch.float32), 0.4),
    }
    for name, (ts, dt) in cases.items():
        sde = problems.make("gbm_ito", dtype=ts.dtype)
        bm = Rec((2, 4), ts.dtype, seed=0)
        y0 = torch.full((2, 4), 0.1, dtype=ts.dtype)
        with torch.no_grad():
            torchsde.sdeint(sde, y0, ts, bm=bm, method="euler", dt=dt)
        keys, _, _ = bm.dump()
        out[name + "__ts"] = ts.numpy()
        out[name + "__dt"] = np.float64(dt)
        out[name + "__queries"] = keys
    np.savez(os.path.join(HERE, "timegrid.npz"), **out)
    print("timegrid.npz:", {k: v.shape for k, v in out.items() if k.endswith("queries")})


# --------------------------------------------------------------------------------------------------- solver
SOLVER_CASES = [
    # name, problem, method, options, levy, (B, d, m), ts, dt
    ("euler_gbm", "gbm_ito", "euler", None, "none", (5, 4, 4), [0., 0.25, 0.5, 0.77], 0.1),
    ("euler_gbm_odd", "gbm_ito", "euler", None, "none", (5, 3, 3), [0., 0.5], 2.0 ** -4),
    ("euler_scalar", "scalar_ito", "euler", None, "none", (5, 4, 1), [0., 0.3, 0.6], 0.05),
    ("euler_additive", "additive_ito", "euler", None, "none", (5, 4, 3), [0., 0.5], 0.05),
    ("euler_general", "general_ito", "euler", None, "none", (6, 4, 4), [0., 0.5], 0.05),
    ("euler_general_odd", "general_odd_ito", "euler", None, "none", (6, 3, 5), [0., 0.5], 0.05),
    ("euler_readme", "readme", "euler", None, "none", (32, 3, 2), list(np.linspace(0, 1, 20)), 1e-3),
    ("milstein_gbm_ito", "gbm_ito", "milstein", None, "none", (5, 4, 4), [0., 0.25, 0.5, 0.77], 0.1),
    ("milstein_gbm_strat", "gbm_strat", "milstein", None, "none", (5, 4, 4), [0., 0.5], 0.05),
    ("milstein_gf_gbm_ito", "gbm_ito", "milstein", {"grad_free": True}, "none", (5, 4, 4), [0., 0.5], 0.05),
    ("milstein_gf_gbm_strat", "gbm_strat", "milstein", {"grad_free": True}, "none", (5, 4, 4), [0., 0.5], 0.05),
    ("milstein_scalar", "scalar_ito", "milstein", None, "none", (5, 4, 1), [0., 0.5], 0.05),
    ("milstein_gf_scalar", "scalar_ito", "milstein", {"grad_free": True}, "none", (5, 4, 1), [0., 0.5], 0.05),
    ("milstein_additive", "additive_ito", "milstein", None, "none", (5, 4, 3), [0., 0.5], 0.05),
    ("milstein_mlpdiag", "mlpdiag_ito", "milstein", None, "none", (5, 4, 4), [0., 0.5], 0.05),
    ("srk_gbm", "gbm_ito", "srk", None, "space-time", (5, 4, 4), [0., 0.25, 0.5, 0.77], 0.1),
    ("srk_scalar", "scalar_ito", "srk", None, "space-time", (5, 4, 1), [0., 0.5], 0.05),
    ("srk_additive", "additive_ito", "srk", None, "space-time", (5, 4, 3), [0., 0.5], 0.05),
    ("srk_mlpdiag", "mlpdiag_ito", "srk", None, "space-time", (5, 4, 4), [0., 0.5], 0.05),
    ("midpoint_gbm", "gbm_strat", "midpoint", None, "none", (5, 4, 4), [0., 0.25, 0.5, 0.77], 0.1),
    ("midpoint_scalar", "scalar_strat", "midpoint", None, "none", (5, 4, 1), [0., 0.5], 0.05),
    ("midpoint_additive", "additive_strat", "midpoint", None, "none", (5, 4, 3), [0., 0.5], 0.05),
    ("midpoint_general", "general_strat", "midpoint", None, "none", (6, 4, 4), [0., 0.5], 0.05),
    ("heun_gbm", "gbm_strat", "heun", None, "none", (5, 4, 4), [0., 0.25, 0.5, 0.77], 0.1),
    ("heun_general", "general_strat", "heun", None, "none", (6, 4, 4), [0., 0.5], 0.05),
    ("heun_scalar", "scalar_strat", "heun", None, "none", (5, 4, 1), [0., 0.5], 0.05),
    ("euler_heun_gbm", "gbm_strat", "euler_heun", None, "none", (5, 4, 4), [0., 0.25, 0.5, 0.77], 0.1),
    ("euler_heun_general", "general_strat", "euler_heun", None, "none", (6, 4, 4), [0., 0.5], 0.05),
    ("euler_heun_additive", "additive_strat", "euler_heun", None, "none", (5, 4, 3), [0., 0.5], 0.05),
    ("log_ode_gbm", "gbm_strat", "log_ode", None, "foster", (5, 4, 4), [0., 0.5], 0.05),
    ("log_ode_general", "general_strat", "log_ode", None, "davie", (6, 4, 4), [0., 0.5], 0.05),
    ("rheun_gbm", "gbm_strat", "reversible_heun", None, "none", (5, 4, 4), [0., 0.25, 0.5, 0.77], 0.1),
    ("rheun_scalar", "scalar_strat", "reversible_heun", None, "none", (5, 4, 1), [0., 0.5], 0.05),
    ("rheun_additive", "additive_strat", "reversible_heun", None, "none", (5, 4, 3), [0., 0.5], 0.05),
    ("rheun_general", "general_strat", "reversible_heun", None, "none", (6, 4, 4), [0., 0.5], 0.05),
]


def gen_solver():
    for name, prob, method, options, levy, (B, d, m), ts, dt in SOLVER_CASES:
        out = {"problem": prob, "method": method, "levy": levy, "dt": np.float64(dt),
               "grad_free": bool(options and options.get("grad_free")), "shape": np.array([B, d, m])}
        for tag, dtype in DT.items():
            sde = problems.make(prob, dtype=dtype, d=d, m=m)
            y0 = torch.full((B, d), 0.1, dtype=dtype)
            tst = torch.tensor(ts, dtype=dtype)
            bm = ReplayBM((B, m), dtype, seed=sum(map(ord, name)), levy=levy)
            with torch.no_grad():
                ys = torchsde.sdeint(sde, y0, tst, bm=bm, method=method, dt=dt,
                                     options=None if options is None else dict(options))
            keys, W, U = bm.dump()
            out[f"{tag}__ts"] = tst.numpy()
            out[f"{tag}__queries"] = keys
            out[f"{tag}__W"] = W
            out[f"{tag}__U"] = U
            if getattr(bm, "A_dump", None) is not None:
                out[f"{tag}__A"] = bm.A_dump
            out[f"{tag}__ys"] = ys.numpy()
            out[f"{tag}__param_checksum"] = np.float64(param_checksum(sde))
        np.savez_compressed(os.path.join(HERE, f"solver_{name}.npz"), **out)
        print(f"solver_{name}.npz  steps={len(keys)}  ys[-1,0,:2]={out['f32__ys'][-1, 0, :2]}")


# ------------------------------------------------------------------------------------------------- adaptive
class RecordingBM(torchsde.BaseBrownian):
    """Wraps the reference's own BrownianInterval (a consistent path) and records every query it answers."""

    def __init__(self, inner):
        super().__init__()
        self.inner, self.table, self.order = inner, {}, []

    def __call__(self, ta, tb=None, return_U=False, return_A=False):
        key = (float(ta), float(tb))
        if key not in self.table:
            if self.inner.levy_area_approximation != "none":
                W, U = self.inner(ta, tb, return_U=True)
            else:
                W, U = self.inner(ta, tb), None
            self.table[key] = (W, torch.zeros_like(W) if U is None else U)
            self.order.append(key)
        W, U = self.table[key]
        return (W, U) if return_U else W

    def __repr__(self):
        return "RecordingBM"

    dtype = property(lambda self: self.inner.dtype)
    device = property(lambda self: self.inner.device)
    shape = property(lambda self: self.inner.shape)
    levy_area_approximation = property(lambda self: self.inner.levy_area_approximation)

    dump = ReplayBM.dump


ADAPTIVE_CASES = [
    ("adaptive_milstein_gbm", "gbm_ito", "milstein", "none", (5, 4, 4), [0., 0.5, 1.0], 0.05, 1e-3, 1e-3),
    ("adaptive_srk_gbm", "gbm_ito", "srk", "space-time", (5, 4, 4), [0., 1.0], 0.1, 1e-4, 1e-4),
    ("adaptive_midpoint_gbm", "gbm_strat", "midpoint", "none", (5, 4, 4), [0., 0.4, 1.0], 0.1, 1e-3, 1e-3),
    ("adaptive_euler_additive", "additive_ito", "euler", "none", (5, 4, 3), [0., 1.0], 0.1, 1e-3, 1e-3),
]


def gen_adaptive():
    for name, prob, method, levy, (B, d, m), ts, dt, rtol, atol in ADAPTIVE_CASES:
        out = {"problem": prob, "method": method, "levy": levy, "dt": np.float64(dt), "grad_free": False,
               "shape": np.array([B, d, m]), "rtol": np.float64(rtol), "atol": np.float64(atol)}
        for tag, dtype in DT.items():
            sde = problems.make(prob, dtype=dtype, d=d, m=m)
            y0 = torch.full((B, d), 0.1, dtype=dtype)
            tst = torch.tensor(ts, dtype=dtype)
            inner = torchsde.BrownianInterval(t0=ts[0], t1=ts[-1], size=(B, m), dtype=dtype, entropy=99,
                                              levy_area_approximation=levy)
            bm = RecordingBM(inner)
            with torch.no_grad():
                ys = torchsde.sdeint(sde, y0, tst, bm=bm, method=method, dt=dt, adaptive=True, rtol=rtol, atol=atol)
            keys, W, U = bm.dump()
            out[f"{tag}__ts"] = tst.numpy()
            out[f"{tag}__queries"] = keys
            out[f"{tag}__W"] = W
            out[f"{tag}__U"] = U
            out[f"{tag}__ys"] = ys.numpy()
            out[f"{tag}__param_checksum"] = np.float64(param_checksum(sde))
        np.savez_compressed(os.path.join(HERE, f"adaptive_{name[len('adaptive_'):]}.npz"), **out)
        print(f"adaptive_{name[len('adaptive_'):]}.npz  distinct queries={len(keys)}")


# -------------------------------------------------------------------------------------------------- adjoint
ADJOINT_CASES = [
    # name, problem, method, adjoint_method, levy, (B, d, m), ts, dt
    ("gbm_ito_euler_euler", "gbm_ito", "euler", "euler", "none", (5, 4, 4), [0., 0.5, 1.0], 2.0 ** -4),
    ("gbm_ito_default", "gbm_ito", "euler", None, "none", (5, 4, 4), [0., 0.5, 1.0], 2.0 ** -4),
    ("gbm_ito_srk_fwd", "gbm_ito", "srk", "milstein", "space-time", (5, 4, 4), [0., 0.5, 1.0], 2.0 ** -4),
    ("gbm_strat_midpoint", "gbm_strat", "midpoint", None, "none", (5, 4, 4), [0., 0.5, 1.0], 2.0 ** -4),
    ("mlpdiag_ito_milstein", "mlpdiag_ito", "milstein", None, "none", (5, 4, 4), [0., 0.5, 1.0], 2.0 ** -4),
    ("mlpdiag_strat_midpoint", "mlpdiag_strat", "midpoint", None, "none", (5, 4, 4), [0., 1.0], 2.0 ** -4),
    ("general_ito_euler", "general_ito", "euler", None, "none", (6, 4, 4), [0., 0.5, 1.0], 2.0 ** -4),
    ("general_strat_midpoint", "general_strat", "midpoint", None, "none", (6, 4, 4), [0., 1.0], 2.0 ** -4),
    ("scalar_ito_euler", "scalar_ito", "euler", None, "none", (5, 4, 1), [0., 1.0], 2.0 ** -4),
    ("additive_ito_euler", "additive_ito", "euler", None, "none", (5, 4, 3), [0., 1.0], 2.0 ** -4),
    ("gbm_ito_nondyadic", "gbm_ito", "euler", "euler", "none", (5, 4, 4), [0., 0.35, 0.9], 0.1),
    ("gbm_strat_rheun", "gbm_strat", "reversible_heun", None, "none", (5, 4, 4), [0., 0.5, 1.0], 2.0 ** -4),
    ("mlpdiag_strat_rheun", "mlpdiag_strat", "reversible_heun", None, "none", (5, 4, 4), [0., 0.5, 1.0], 2.0 ** -4),
    ("general_strat_rheun", "general_strat", "reversible_heun", None, "none", (6, 4, 4), [0., 1.0], 2.0 ** -4),
    ("scalar_strat_rheun", "scalar_strat", "reversible_heun", None, "none", (5, 4, 1), [0., 1.0], 2.0 ** -4),
    # the other solvers the reference can run on an adjoint SDE (they only need f_and_g_prod / g_prod)
    ("gbm_strat_heun", "gbm_strat", "heun", "heun", "none", (5, 4, 4), [0., 0.5, 1.0], 2.0 ** -4),
    ("mlpdiag_strat_euler_heun", "mlpdiag_strat", "euler_heun", "euler_heun", "none", (5, 4, 4), [0., 0.5, 1.0],
     2.0 ** -4),
    ("general_strat_adj_heun", "general_strat", "midpoint", "heun", "none", (6, 4, 4), [0., 1.0], 2.0 ** -4),
    ("scalar_strat_adj_euler_heun", "scalar_strat", "midpoint", "euler_heun", "none", (5, 4, 1), [0., 1.0], 2.0 ** -4),
]


def gen_adjoint():
    for name, prob, method, adjoint_method, levy, (B, d, m), ts, dt in ADJOINT_CASES:
        out = {"problem": prob, "method": method, "adjoint_method": adjoint_method or "", "levy": levy,
               "dt": np.float64(dt), "grad_free": False, "shape": np.array([B, d, m])}
        for tag, dtype in DT.items():
            sde = problems.make(prob, dtype=dtype, d=d, m=m)
            y0 = torch.full((B, d), 0.1, dtype=dtype, requires_grad=True)
            tst = torch.tensor(ts, dtype=dtype)
            bm = ReplayBM((B, m), dtype, seed=sum(map(ord, name)), levy=levy)
            ys = torchsde.sdeint_adjoint(sde, y0, tst, bm=bm, method=method, adjoint_method=adjoint_method, dt=dt)
            wt = torch.tensor(np.random.default_rng(5).standard_normal(tuple(ys.shape)), dtype=dtype)
            (ys * wt).sum().backward()
            keys, W, U = bm.dump()
            out[f"{tag}__ts"] = tst.numpy()
            out[f"{tag}__queries"] = keys
            out[f"{tag}__W"] = W
            out[f"{tag}__U"] = U
            out[f"{tag}__ys"] = ys.detach().numpy()
            out[f"{tag}__loss_weights"] = wt.numpy()
            out[f"{tag}__grad_y0"] = y0.grad.numpy()
            for j, p in enumerate(sde.parameters()):
                out[f"{tag}__grad_p{j}"] = (torch.zeros_like(p) if p.grad is None else p.grad).numpy()
            out[f"{tag}__param_checksum"] = np.float64(param_checksum(sde))
        np.savez_compressed(os.path.join(HERE, f"adjoint_{name}.npz"), **out)
        print(f"adjoint_{name}.npz  queries={len(keys)}  |grad_y0|={np.abs(out['f32__grad_y0']).sum():.4f}")


# adjoint_adaptive=True: step doubling on the flat augmented state, restarted from `dt` on every output interval
# (adjoint.py:83-112 builds one solver and calls its integrate() per interval; base_solver.py:105,117-142)
ADJOINT_ADAPTIVE_CASES = [
    # name, problem, method, adjoint_method, (B, d, m), ts, dt, adjoint_rtol, adjoint_atol
    # (tolerances chosen so that attempts are both accepted and rejected without the whole sweep sitting on dt_min)
    ("gbm_ito_euler", "gbm_ito", "euler", "euler", (5, 4, 4), [0., 0.5, 1.0], 2.0 ** -3, 1e-1, 1e-2),
    ("mlpdiag_ito_milstein", "mlpdiag_ito", "milstein", None, (5, 4, 4), [0., 0.5, 1.0], 2.0 ** -3, 3e-2, 3e-3),
    ("gbm_strat_midpoint", "gbm_strat", "midpoint", None, (5, 4, 4), [0., 0.25, 1.0], 2.0 ** -3, 1e-1, 1e-2),
    ("general_strat_heun", "general_strat", "midpoint", "heun", (6, 4, 4), [0., 1.0], 2.0 ** -3, 1.0, 1e-1),
    ("mlpdiag_strat_rheun", "mlpdiag_strat", "reversible_heun", None, (5, 4, 4), [0., 0.5, 1.0], 2.0 ** -3, 1e-1,
     1e-2),
    ("general_strat_rheun", "general_strat", "reversible_heun", None, (6, 4, 4), [0., 1.0], 2.0 ** -3, 1.0, 1e-1),
]
ADJOINT_ADAPTIVE_DT_MIN = 2.0 ** -9


def gen_adjoint_adaptive():
    import warnings
    dtype = torch.float64
    for name, prob, method, adjoint_method, (B, d, m), ts, dt, rtol, atol in ADJOINT_ADAPTIVE_CASES:
        sde = problems.make(prob, dtype=dtype, d=d, m=m)
        y0 = torch.full((B, d), 0.1, dtype=dtype, requires_grad=True)
        tst = torch.tensor(ts, dtype=dtype)
        bm = ReplayBM((B, m), dtype, seed=sum(map(ord, name)), levy="none")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            ys = torchsde.sdeint_adjoint(sde, y0, tst, bm=bm, method=method, adjoint_method=adjoint_method, dt=dt,
                                         adjoint_adaptive=True, adjoint_rtol=rtol, adjoint_atol=atol,
                                         dt_min=ADJOINT_ADAPTIVE_DT_MIN)
            wt = torch.tensor(np.random.default_rng(5).standard_normal(tuple(ys.shape)), dtype=dtype)
            forward_queries = len(bm.order)
            (ys * wt).sum().backward()
        keys, W, U = bm.dump()
        out = {"problem": prob, "method": method, "adjoint_method": adjoint_method or "", "levy": "none",
               "dt": np.float64(dt), "grad_free": False, "shape": np.array([B, d, m]),
               "adjoint_rtol": np.float64(rtol), "adjoint_atol": np.float64(atol),
               "dt_min": np.float64(ADJOINT_ADAPTIVE_DT_MIN),
               "f64__ts": tst.numpy(), "f64__queries": keys, "f64__W": W, "f64__U": U, "f64__ys": ys.detach().numpy(),
               "f64__loss_weights": wt.numpy(), "f64__grad_y0": y0.grad.numpy(),
               "f64__param_checksum": np.float64(param_checksum(sde)),
               "backward_queries": np.int64(len(keys) - forward_queries)}
        for j, p in enumerate(sde.parameters()):
            out[f"f64__grad_p{j}"] = (torch.zeros_like(p) if p.grad is None else p.grad).numpy()
        np.savez_compressed(os.path.join(HERE, f"adjoint_adaptive_{name}.npz"), **out)
        print(f"adjoint_adaptive_{name}.npz  forward queries={forward_queries}  backward queries="
              f"{len(keys) - forward_queries}  |grad_y0|={np.abs(out['f64__grad_y0']).sum():.4f}")


# second derivatives through sdeint_adjoint: the reference nests a second adjoint solve (adjoint.py:97-112)
DOUBLE_BACKWARD_CASES = [
    # name, problem, method, adjoint_method, (B, d, m), ts, dt
    # (for an Ito SDE the reference's nested solve stops with "Adjoint `f_and_g` not defined": the corrected drift of
    # the adjoint of the adjoint asks for it, adjoint_sde.py:308,318 -> :271; only Stratonovich SDEs get through)
    ("gbm_strat_midpoint", "gbm_strat", "midpoint", None, (5, 4, 4), [0., 0.25, 0.5], 2.0 ** -7),
    ("mlpdiag_strat_midpoint", "mlpdiag_strat", "midpoint", None, (5, 4, 4), [0., 0.5], 2.0 ** -7),
    ("general_strat_heun", "general_strat", "heun", "heun", (6, 4, 4), [0., 0.5], 2.0 ** -7),
    ("scalar_strat_euler_heun", "scalar_strat", "midpoint", "euler_heun", (5, 4, 1), [0., 0.5], 2.0 ** -7),
]


def gen_double_backward():
    dtype = torch.float64
    for name, prob, method, adjoint_method, (B, d, m), ts, dt in DOUBLE_BACKWARD_CASES:
        sde = problems.make(prob, dtype=dtype, d=d, m=m)
        params = list(sde.parameters())
        y0 = torch.full((B, d), 0.1, dtype=dtype, requires_grad=True)
        tst = torch.tensor(ts, dtype=dtype)
        bm = ReplayBM((B, m), dtype, seed=sum(map(ord, name)), levy="none")
        ys = torchsde.sdeint_adjoint(sde, y0, tst, bm=bm, method=method, adjoint_method=adjoint_method, dt=dt)
        rng = np.random.default_rng(11)
        wt = torch.tensor(rng.standard_normal(tuple(ys.shape)), dtype=dtype)
        loss = (ys ** 2 * wt).sum()
        first = torch.autograd.grad(loss, [y0] + params, create_graph=True, allow_unused=True)
        mix = [torch.tensor(rng.standard_normal(tuple(x.shape)), dtype=dtype) for x in [y0] + params]
        phi = sum((g * w).sum() for g, w in zip(first, mix) if g is not None)
        second = torch.autograd.grad(phi, [y0] + params, allow_unused=True)
        keys, W, U = bm.dump()
        out = {"problem": prob, "method": method, "adjoint_method": adjoint_method or "", "levy": "none",
               "dt": np.float64(dt), "grad_free": False, "shape": np.array([B, d, m]),
               "f64__ts": tst.numpy(), "f64__queries": keys, "f64__W": W, "f64__U": U, "f64__ys": ys.detach().numpy(),
               "f64__loss_weights": wt.numpy(), "f64__param_checksum": np.float64(param_checksum(sde))}
        for j, (g, w, h, x) in enumerate(zip(first, mix, second, [y0] + params)):
            out[f"f64__first{j}"] = (torch.zeros_like(x) if g is None else g.detach()).numpy()
            out[f"f64__mix{j}"] = w.numpy()
            out[f"f64__second{j}"] = (torch.zeros_like(x) if h is None else h).numpy()
        np.savez_compressed(os.path.join(HERE, f"double_backward_{name}.npz"), **out)
        print(f"double_backward_{name}.npz  queries={len(keys)}  |second|="
              f"{sum(float(np.abs(out[f'f64__second{j}']).sum()) for j in range(len(first))):.4f}")


# --------------------------------------------------------------------------------------------------- bridge
def gen_bridge():
    """Record what the reference's bridge code does with known normals, and multi-interval merges."""
    recs = []
    orig_randn = ref_bi._randn
    log = {}

    def spy_randn(size, dtype, device, seed):
        x = orig_randn(size, dtype, device, seed)
        log[int(seed)] = x
        return x

    ref_bi._randn = spy_randn
    try:
        out = {}
        for levy in ("none", "space-time"):
            for tag, dtype in DT.items():
                bm = torchsde.BrownianInterval(t0=0., t1=1., size=(6,), dtype=dtype, entropy=1234,
                                               levy_area_approximation=levy)
                W0, H0 = bm._w_h
                # one split at x: children are (0,x) and (x,1)
                x = 0.3
                res_l = bm(0., x, return_U=(levy != "none"))
                res_r = bm(x, 1., return_U=(levy != "none"))
                Wl, Ul = res_l if levy != "none" else (res_l, None)
                Wr, Ur = res_r if levy != "none" else (res_r, None)
                X1 = log[int(bm._W_seed)]
                X2 = log.get(int(bm._H_seed))
                pre = f"{levy}__{tag}__"
                out[pre + "x"] = np.float64(x)
                out[pre + "W"] = W0.numpy()
                out[pre + "H"] = (H0 if H0 is not None else torch.zeros_like(W0)).numpy()
                out[pre + "X1"] = X1.numpy()
                out[pre + "X2"] = (X2 if X2 is not None else torch.zeros_like(X1)).numpy()
                out[pre + "Wl"] = Wl.numpy()
                out[pre + "Wr"] = Wr.numpy()
                if levy != "none":
                    out[pre + "Ul"] = Ul.numpy()
                    out[pre + "Ur"] = Ur.numpy()
                    # merged query across the split: pieces (0.3->0.55 after another split) etc.
                    W_a, U_a = bm(0.1, 0.3, return_U=True)
                    W_b, U_b = bm(0.3, 0.8, return_U=True)
                    W_ab, U_ab = bm(0.1, 0.8, return_U=True)
                    out[pre + "merge_ta_u_t"] = np.array([0.1, 0.3, 0.8])
                    for k, v in (("W_a", W_a), ("U_a", U_a), ("W_b", W_b), ("U_b", U_b), ("W_ab", W_ab),
                                 ("U_ab", U_ab)):
                        out[pre + k] = v.numpy()
    finally:
        ref_bi._randn = orig_randn
    np.savez(os.path.join(HERE, "bridge.npz"), **out)
    print("bridge.npz:", len(out), "arrays")


# --------------------------------------------------------------------------------------------- brownian seq
def gen_brownian_seq():
    out = {}
    rng = np.random.default_rng(7)
    seqs = {
        "seq_steps": [(k * 0.01, (k + 1) * 0.01) for k in range(100)] + [(1 - (k + 1) * 0.01, 1 - k * 0.01)
                                                                           for k in range(100)],
        "seq_random": [tuple(sorted(rng.uniform(0, 1, 2))) for _ in range(150)],
    }
    for sname, seq in seqs.items():
        for levy in ("none", "space-time"):
            for hint in (None, 0.01):
                bm = torchsde.BrownianInterval(t0=0., t1=1., size=(3, 2), dtype=torch.float64, entropy=4321,
                                               levy_area_approximation=levy, dt=hint)
                Ws, Us = [], []
                for (a, b) in seq:
                    if levy == "none":
                        Ws.append(bm(a, b).numpy())
                    else:
                        W, U = bm(a, b, return_U=True)
                        Ws.append(W.numpy())
                        Us.append(U.numpy())
                pre = f"{sname}__{levy}__{'hint' if hint else 'nohint'}__"
                out[pre + "queries"] = np.array(seq, dtype=np.float64)
                out[pre + "W"] = np.stack(Ws)
                if Us:
                    out[pre + "U"] = np.stack(Us)
    np.savez_compressed(os.path.join(HERE, "brownian_seq.npz"), **out)
    print("brownian_seq.npz:", len(out), "arrays")


# ------------------------------------------------------------------------------------- closed-form neural SDE
# The perceptron-drift module (torchsde_amd.MLPDriftDiagonalSDE: an ordinary nn.Module with f, g, noise_type,
# sde_type) solved and differentiated by the REAL reference in float64, on the increments of the counter-RNG path
# (entropy, one cell per step) served by the C twin of the generator -- the path the trajectory kernels generate for
# themselves on the GPU. Pins tsde_trajectory_mlp_diag and its reverse sweep against the reference itself.
CLOSED_FORM_CASES = [
    # name, activation, diffusion, sde_type, method, with gradients
    ("euler_tanh_affine", "tanh", "affine", "ito", "euler", True),
    ("euler_softplus_sigmoid", "softplus", "sigmoid", "ito", "euler", True),
    ("milstein_softplus_affine", "softplus", "affine", "ito", "milstein", True),
    ("milstein_strat_tanh_affine", "tanh", "affine", "stratonovich", "milstein", True),
    ("midpoint_tanh_sigmoid", "tanh", "sigmoid", "stratonovich", "midpoint", False),
    # SRK (SRID2) needs the space-time Levy area of the path: the C twin serves (W, U)
    ("srk_softplus_sigmoid", "softplus", "sigmoid", "ito", "srk", False),
    ("srk_tanh_affine", "tanh", "affine", "ito", "srk", False),
]


def gen_closed_form():
    import torchsde_amd
    from oracle import counter
    B, d, hidden, steps, dt, entropy = 48, 32, 64, 16, 2.0 ** -5, 4242
    edges = np.arange(steps + 1) * dt
    ts = [0.0, 5 * dt, steps * dt]

    class CounterPath(torchsde.BaseBrownian):
        def __init__(self, levy):
            super().__init__()
            self.levy = levy

        def __call__(self, ta, tb=None, return_U=False, return_A=False):
            W, U, _ = counter.query(B * d, entropy, edges, float(ta), float(tb), dtype=np.float32,
                                    have_h=self.levy != "none")
            W = torch.from_numpy(W).reshape(B, d).double()
            return (W, torch.from_numpy(U).reshape(B, d).double()) if return_U else W

        def __repr__(self):
            return "CounterPath"

        dtype = property(lambda self: torch.float64)
        device = property(lambda self: torch.device("cpu"))
        shape = property(lambda self: (B, d))
        levy_area_approximation = property(lambda self: self.levy)

    for name, activation, diffusion, sde_type, method, with_grads in CLOSED_FORM_CASES:
        levy = "space-time" if method == "srk" else "none"
        gen = torch.Generator().manual_seed(sum(map(ord, name)))
        sigmoid = diffusion == "sigmoid"
        sde = torchsde_amd.MLPDriftDiagonalSDE(
            d, hidden, activation=activation, sde_type=sde_type, diffusion=diffusion,
            diff_scale=0.4 if sigmoid else 1.0, dtype=torch.float64,
            diff_rate=(2.0 if sigmoid else 0.2) * torch.rand(d, generator=gen, dtype=torch.float64) - 0.1,
            diff_shift=0.1 + 0.2 * torch.rand(d, generator=gen, dtype=torch.float64))
        with torch.no_grad():
            sde.lin1.weight.copy_(torch.randn(hidden, d, generator=gen, dtype=torch.float64) / d ** 0.5)
            sde.lin2.weight.copy_(torch.randn(d, hidden, generator=gen, dtype=torch.float64) / hidden ** 0.5)
            sde.lin1.bias.copy_(0.3 * torch.randn(hidden, generator=gen, dtype=torch.float64))
            sde.lin2.bias.copy_(0.3 * torch.randn(d, generator=gen, dtype=torch.float64))
        y0 = (0.5 * torch.randn(B, d, generator=gen, dtype=torch.float64)).requires_grad_(True)
        weights = torch.randn(len(ts), B, d, generator=gen, dtype=torch.float64)
        ys = torchsde.sdeint(sde, y0, torch.tensor(ts, dtype=torch.float64), bm=CounterPath(levy), method=method, dt=dt)
        out = {"activation": activation, "diffusion": diffusion, "sde_type": sde_type, "method": method, "levy": levy,
               "diff_scale": np.float64(sde.diff_scale), "with_grads": with_grads, "entropy": np.int64(entropy),
               "dt": np.float64(dt), "ts": np.asarray(ts), "shape": np.array([B, d, hidden, steps]),
               "y0": y0.detach().numpy(), "weights": weights.numpy(), "ys": ys.detach().numpy()}
        for pname, p in sde.named_parameters():
            out["param__" + pname] = p.detach().numpy()
        if with_grads:
            (ys * weights).sum().backward()
            out["grad__y0"] = y0.grad.numpy()
            for pname, p in sde.named_parameters():
                out["grad__" + pname] = p.grad.numpy()
        np.savez_compressed(os.path.join(HERE, f"closed_form_mlp_{name}.npz"), **out)
        print(f"closed_form_mlp_{name}.npz  |ys|={np.abs(out['ys']).mean():.4f}"
              + (f"  |grad lin1.weight|={np.abs(out['grad__lin1.weight']).mean():.4f}" if with_grads else ""))


AFFINE_CASES = [
    # name, sde_type, method, levy
    ("euler", "ito", "euler", "none"),
    ("milstein", "ito", "milstein", "none"),
    ("milstein_strat", "stratonovich", "milstein", "none"),
    ("srk", "ito", "srk", "space-time"),
    ("midpoint", "stratonovich", "midpoint", "none"),
]


def gen_closed_form_affine():
    """torchsde_amd.AffineDiagonalSDE (per-channel affine drift and diffusion) solved and differentiated by the REAL
    reference in float64 on the counter-RNG path: pins tsde_trajectory_affine_diag and its sensitivity variant."""
    import torchsde_amd
    from oracle import counter
    B, d, steps, dt, entropy = 40, 12, 16, 2.0 ** -5, 777001
    edges = np.arange(steps + 1) * dt
    ts = [0.0, 5 * dt, steps * dt]

    for name, sde_type, method, levy in AFFINE_CASES:
        class CounterPath(torchsde.BaseBrownian):
            def __call__(self, ta, tb=None, return_U=False, return_A=False):
                W, U, _ = counter.query(B * d, entropy, edges, float(ta), float(tb), dtype=np.float64,
                                        have_h=levy != "none")
                W = torch.from_numpy(W).reshape(B, d)
                return (W, torch.from_numpy(U).reshape(B, d)) if return_U else W

            def __repr__(self):
                return "CounterPath"

            dtype = property(lambda self: torch.float64)
            device = property(lambda self: torch.device("cpu"))
            shape = property(lambda self: (B, d))
            levy_area_approximation = property(lambda self: levy)

        gen = torch.Generator().manual_seed(sum(map(ord, name)))
        rnd = lambda lo, hi: lo + (hi - lo) * torch.rand(d, generator=gen, dtype=torch.float64)   # noqa: E731
        sde = torchsde_amd.AffineDiagonalSDE(rnd(-0.8, 0.2), rnd(-0.3, 0.3), rnd(0.1, 0.5), rnd(0.0, 0.2),
                                             sde_type=sde_type, dtype=torch.float64)
        y0 = (0.5 + torch.rand(B, d, generator=gen, dtype=torch.float64)).requires_grad_(True)
        weights = torch.randn(len(ts), B, d, generator=gen, dtype=torch.float64)
        ys = torchsde.sdeint(sde, y0, torch.tensor(ts, dtype=torch.float64), bm=CounterPath(), method=method, dt=dt)
        (ys * weights).sum().backward()
        out = {"sde_type": sde_type, "method": method, "levy": levy, "entropy": np.int64(entropy), "dt": np.float64(dt),
               "ts": np.asarray(ts), "shape": np.array([B, d, steps]), "y0": y0.detach().numpy(),
               "weights": weights.numpy(), "ys": ys.detach().numpy(), "grad__y0": y0.grad.numpy()}
        for pname, p in sde.named_parameters():
            out["param__" + pname] = p.detach().numpy()
            out["grad__" + pname] = p.grad.numpy()
        np.savez_compressed(os.path.join(HERE, f"closed_form_affine_{name}.npz"), **out)
        print(f"closed_form_affine_{name}.npz  |ys|={np.abs(out['ys']).mean():.4f}  "
              f"|grad drift_rate|={np.abs(out['grad__drift_rate']).mean():.4f}")


# ---------------------------------------------------------------------------------------------------- logqp
# `logqp=True` (reference sdeint.py:142-144, 284-295; base_sde.py:240-306): the state carries one more column, the
# running KL integrand 0.5 |u|^2 with u = (f - h) / g (diagonal) or pinv(g) (f - h) (general); the Brownian motion
# therefore has d + 1 channels for diagonal noise. Forward values, and for the adjoint cases the gradients of
# sum(ys * w) + sum(log_ratio * v), from the real reference under replayed increments.
LOGQP_CASES = [
    # name, problem, method, adjoint (None = plain sdeint), names, (B, d, m), ts, dt
    ("euler_gbm", "gbm_ito", "euler", None, None, (5, 4, 5), [0., 0.25, 0.5, 0.77], 0.1),
    ("midpoint_mlpdiag", "mlpdiag_strat", "midpoint", None, None, (5, 4, 5), [0., 0.5, 1.0], 2.0 ** -4),
    ("euler_general", "general_ito", "euler", None, None, (6, 4, 4), [0., 0.5], 0.05),
    ("srk_gbm", "gbm_ito", "srk", None, None, (5, 4, 5), [0., 0.5], 0.05),
    ("adjoint_mlpdiag_names", "mlpdiag_ito", "euler", "default", {"prior_drift": "prior"}, (5, 4, 5), [0., 0.5, 1.0],
     2.0 ** -4),
    ("adjoint_gbm_strat", "gbm_strat", "midpoint", "default", None, (5, 4, 5), [0., 0.5, 1.0], 2.0 ** -4),
    ("backprop_mlpdiag", "mlpdiag_ito", "euler", "backprop", None, (5, 4, 5), [0., 0.5, 1.0], 2.0 ** -4),
]


def gen_logqp():
    for name, prob, method, adjoint, names, (B, d, m), ts, dt in LOGQP_CASES:
        levy = "space-time" if method == "srk" else "none"
        out = {"problem": prob, "method": method, "levy": levy, "dt": np.float64(dt), "grad_free": False,
               "shape": np.array([B, d, m]), "adjoint": adjoint or "",
               "names": "" if names is None else ",".join(f"{k}={v}" for k, v in names.items())}
        for tag, dtype in DT.items():
            sde = problems.make(prob, dtype=dtype, d=d, m=m)
            y0 = torch.full((B, d), 0.1, dtype=dtype, requires_grad=adjoint is not None)
            tst = torch.tensor(ts, dtype=dtype)
            bm = ReplayBM((B, m), dtype, seed=sum(map(ord, "logqp_" + name)), levy=levy)
            if adjoint == "default":
                ys, log_ratio = torchsde.sdeint_adjoint(sde, y0, tst, bm=bm, method=method, dt=dt, logqp=True, names=names)
            elif adjoint == "backprop":
                ys, log_ratio = torchsde.sdeint(sde, y0, tst, bm=bm, method=method, dt=dt, logqp=True, names=names)
            else:
                with torch.no_grad():
                    ys, log_ratio = torchsde.sdeint(sde, y0, tst, bm=bm, method=method, dt=dt, logqp=True, names=names)
            if adjoint is not None:
                rng = np.random.default_rng(11)
                wy = torch.tensor(rng.standard_normal(tuple(ys.shape)), dtype=dtype)
                wl = torch.tensor(rng.standard_normal(tuple(log_ratio.shape)), dtype=dtype)
                ((ys * wy).sum() + (log_ratio * wl).sum()).backward()
                out[f"{tag}__wy"], out[f"{tag}__wl"] = wy.numpy(), wl.numpy()
                out[f"{tag}__grad_y0"] = y0.grad.numpy()
                for j, p in enumerate(sde.parameters()):
                    out[f"{tag}__grad_p{j}"] = (torch.zeros_like(p) if p.grad is None else p.grad).numpy()
            keys, W, U = bm.dump()
            out[f"{tag}__ts"] = tst.numpy()
            out[f"{tag}__queries"] = keys
            out[f"{tag}__W"] = W
            out[f"{tag}__U"] = U
            out[f"{tag}__ys"] = ys.detach().numpy()
            out[f"{tag}__log_ratio"] = log_ratio.detach().numpy()
            out[f"{tag}__param_checksum"] = np.float64(param_checksum(sde))
        np.savez_compressed(os.path.join(HERE, f"logqp_{name}.npz"), **out)
        print(f"logqp_{name}.npz  queries={len(keys)}  log_ratio[:, 0]={out['f32__log_ratio'][:, 0]}")


# ------------------------------------------------------------------------------ closed-form neural SDE, adjoint
# `torchsde.sdeint_adjoint(..., adjoint_method="euler")` of the REAL reference on the perceptron-drift module, float64,
# on the counter-RNG path (forward and time-reversed queries served by the C twin of the generator): pins
# tsde_adjoint_mlp_diag, the stochastic adjoint on the matrix cores.
CLOSED_FORM_ADJOINT_CASES = [
    # name, activation, diffusion, forward method, ts (multiples of dt)
    ("euler_softplus_sigmoid", "softplus", "sigmoid", "euler", [0.0, 16]),
    ("euler_tanh_affine_outputs", "tanh", "affine", "euler", [0.0, 5, 11, 16]),
    ("milstein_softplus_affine", "softplus", "affine", "milstein", [0.0, 8, 16]),
    # adjoint_method="milstein" (the default for diagonal Ito noise, adjoint.py:281-296), Ito and Stratonovich:
    # name, activation, diffusion, forward method, ts, sde_type, adjoint_method
    ("adjmil_softplus_sigmoid", "softplus", "sigmoid", "milstein", [0.0, 16], "ito", "milstein"),
    ("adjmil_tanh_affine_outputs", "tanh", "affine", "euler", [0.0, 5, 11, 16], "ito", "milstein"),
    ("adjmil_strat_softplus_sigmoid", "softplus", "sigmoid", "midpoint", [0.0, 7, 16], "stratonovich", "milstein"),
    ("adjmil_strat_tanh_affine", "tanh", "affine", "milstein", [0.0, 16], "stratonovich", "milstein"),
]


def gen_closed_form_adjoint():
    import torchsde_amd
    from oracle import counter
    B, d, hidden, steps, dt, entropy = 48, 32, 64, 16, 2.0 ** -5, 515151
    edges = np.arange(steps + 1) * dt

    class CounterPath(torchsde.BaseBrownian):
        def __call__(self, ta, tb=None, return_U=False, return_A=False):
            W, _, _ = counter.query(B * d, entropy, edges, float(ta), float(tb), dtype=np.float32, have_h=False)
            return torch.from_numpy(W).reshape(B, d).double()

        def __repr__(self):
            return "CounterPath"

        dtype = property(lambda self: torch.float64)
        device = property(lambda self: torch.device("cpu"))
        shape = property(lambda self: (B, d))
        levy_area_approximation = property(lambda self: "none")

    for name, activation, diffusion, method, marks, *rest in CLOSED_FORM_ADJOINT_CASES:
        sde_type, adjoint_method = rest if rest else ("ito", "euler")
        gen = torch.Generator().manual_seed(sum(map(ord, "adjoint_" + name)))
        sigmoid = diffusion == "sigmoid"
        sde = torchsde_amd.MLPDriftDiagonalSDE(
            d, hidden, activation=activation, sde_type=sde_type, diffusion=diffusion,
            diff_scale=0.4 if sigmoid else 1.0, dtype=torch.float64,
            diff_rate=(2.0 if sigmoid else 0.2) * torch.rand(d, generator=gen, dtype=torch.float64) - 0.1,
            diff_shift=0.1 + 0.2 * torch.rand(d, generator=gen, dtype=torch.float64))
        with torch.no_grad():
            sde.lin1.weight.copy_(torch.randn(hidden, d, generator=gen, dtype=torch.float64) / d ** 0.5)
            sde.lin2.weight.copy_(torch.randn(d, hidden, generator=gen, dtype=torch.float64) / hidden ** 0.5)
            sde.lin1.bias.copy_(0.3 * torch.randn(hidden, generator=gen, dtype=torch.float64))
            sde.lin2.bias.copy_(0.3 * torch.randn(d, generator=gen, dtype=torch.float64))
        ts = [float(marks[0])] + [k * dt for k in marks[1:]]
        y0 = (0.5 * torch.randn(B, d, generator=gen, dtype=torch.float64)).requires_grad_(True)
        weights = torch.randn(len(ts), B, d, generator=gen, dtype=torch.float64)
        ys = torchsde.sdeint_adjoint(sde, y0, torch.tensor(ts, dtype=torch.float64), bm=CounterPath(), method=method,
                                     adjoint_method=adjoint_method, dt=dt)
        (ys * weights).sum().backward()
        out = {"activation": activation, "diffusion": diffusion, "sde_type": sde_type, "method": method,
               "adjoint_method": adjoint_method,
               "diff_scale": np.float64(sde.diff_scale), "entropy": np.int64(entropy), "dt": np.float64(dt),
               "ts": np.asarray(ts), "shape": np.array([B, d, hidden, steps]), "y0": y0.detach().numpy(),
               "weights": weights.numpy(), "ys": ys.detach().numpy(), "grad__y0": y0.grad.numpy()}
        for pname, p in sde.named_parameters():
            out["param__" + pname] = p.detach().numpy()
            out["grad__" + pname] = p.grad.numpy()
        np.savez_compressed(os.path.join(HERE, f"closed_form_adjoint_{name}.npz"), **out)
        print(f"closed_form_adjoint_{name}.npz  |ys|={np.abs(out['ys']).mean():.4f}  "
              f"|grad lin1.weight|={np.abs(out['grad__lin1.weight']).mean():.4f}")


# --------------------------------------------------------------------------------- closed-form elementwise SDE
# torchsde_amd.ElementwiseDiagonalSDE solved by the REAL reference in float64 on the counter-RNG path: pins
# tsde_trajectory_expr_diag. The first three cases are the SDE of the reference's own benchmark
# (benchmarks/brownian.py:131-139: f = y, g = exp(-y)).
EXPR_CASES = [
    # name, drift fn, diffusion fn, drift coefs (scale, rate, shift, offset), diffusion coefs, sde_type, method, levy
    ("benchmark_euler", "identity", "exp", (1.0, 1.0, 0.0, 0.0), (1.0, -1.0, 0.0, 0.0), "ito", "euler", "none"),
    ("benchmark_milstein", "identity", "exp", (1.0, 1.0, 0.0, 0.0), (1.0, -1.0, 0.0, 0.0), "ito", "milstein", "none"),
    ("benchmark_srk", "identity", "exp", (1.0, 1.0, 0.0, 0.0), (1.0, -1.0, 0.0, 0.0), "ito", "srk", "space-time"),
    ("tanh_sigmoid_midpoint", "tanh", "sigmoid", None, None, "stratonovich", "midpoint", "none"),
    ("sin_softplus_milstein_strat", "sin", "softplus", None, None, "stratonovich", "milstein", "none"),
    ("softplus_cos_euler", "softplus", "cos", None, None, "ito", "euler", "none"),
]


def gen_closed_form_expr():
    import torchsde_amd
    from oracle import counter
    B, d, steps, entropy = 40, 12, 16, 909090

    for name, fk, gk, fc, gc, sde_type, method, levy in EXPR_CASES:
        # (explicit schemes on f = y, g = exp(-y) blow up once a path wanders to y << 0, where exp(-y) explodes: the
        #  benchmark SDE gets a short horizon)
        dt = 2.0 ** -8 if name.startswith("benchmark") else 2.0 ** -5
        edges = np.arange(steps + 1) * dt
        ts = [0.0, 5 * dt, 7.5 * dt, steps * dt]

        class CounterPath(torchsde.BaseBrownian):
            def __call__(self, ta, tb=None, return_U=False, return_A=False):
                W, U, _ = counter.query(B * d, entropy, edges, float(ta), float(tb), dtype=np.float64,
                                        have_h=levy != "none")
                W = torch.from_numpy(W).reshape(B, d)
                return (W, torch.from_numpy(U).reshape(B, d)) if return_U else W

            def __repr__(self):
                return "CounterPath"

            dtype = property(lambda self: torch.float64)
            device = property(lambda self: torch.device("cpu"))
            shape = property(lambda self: (B, d))
            levy_area_approximation = property(lambda self: levy)

        gen = torch.Generator().manual_seed(sum(map(ord, "expr_" + name)))
        rnd = lambda lo, hi: lo + (hi - lo) * torch.rand(d, generator=gen, dtype=torch.float64)   # noqa: E731
        if fc is None:
            fc = (rnd(-0.8, 0.8), rnd(0.5, 1.5), rnd(-0.3, 0.3), rnd(-0.2, 0.2))
            gc = (rnd(0.2, 0.6), rnd(-1.5, 1.5), rnd(-0.3, 0.3), rnd(0.05, 0.2))
        sde = torchsde_amd.ElementwiseDiagonalSDE(fk, gk, fc, gc, sde_type=sde_type, dtype=torch.float64)
        y0 = 0.6 * torch.rand(B, d, generator=gen, dtype=torch.float64) - 0.3
        with torch.no_grad():
            ys = torchsde.sdeint(sde, y0, torch.tensor(ts, dtype=torch.float64), bm=CounterPath(), method=method, dt=dt)
        out = {"drift": fk, "diffusion": gk, "sde_type": sde_type, "method": method, "levy": levy,
               "entropy": np.int64(entropy), "dt": np.float64(dt), "ts": np.asarray(ts), "shape": np.array([B, d, steps]),
               "y0": y0.numpy(), "ys": ys.numpy()}
        for pname, p in sde.named_parameters():
            out["param__" + pname] = p.detach().numpy()
        np.savez_compressed(os.path.join(HERE, f"closed_form_expr_{name}.npz"), **out)
        print(f"closed_form_expr_{name}.npz  |ys|={np.abs(out['ys']).mean():.4f}")


# Plain USER modules whose drift / diffusion are polynomials of the state (workloads.problems.DoubleWell, Logistic: nothing
# of this package in them), solved by the REAL reference in float64 on the counter-RNG path: pins the cubic mode of
# tsde_trajectory_expr_diag (TSDE_FN_POLY3) that recognise.py routes such modules to -- against the reference itself, not
# against this package's stepwise route.
POLY3_CASES = [
    # name, problem class, sde_type, method, levy
    ("doublewell_euler", "DoubleWell", "ito", "euler", "none"),
    ("doublewell_milstein", "DoubleWell", "ito", "milstein", "none"),
    ("doublewell_srk", "DoubleWell", "ito", "srk", "space-time"),
    ("logistic_euler", "Logistic", "ito", "euler", "none"),
    ("logistic_midpoint", "Logistic", "stratonovich", "midpoint", "none"),
    ("logistic_milstein_strat", "Logistic", "stratonovich", "milstein", "none"),
]


def gen_poly3():
    from oracle import counter
    from workloads import problems
    B, d, steps, entropy, dt = 40, 12, 16, 515151, 2.0 ** -5

    for name, cls, sde_type, method, levy in POLY3_CASES:
        edges = np.arange(steps + 1) * dt
        ts = [0.0, 5 * dt, 7.5 * dt, steps * dt]

        class CounterPath(torchsde.BaseBrownian):
            def __call__(self, ta, tb=None, return_U=False, return_A=False):
                W, U, _ = counter.query(B * d, entropy, edges, float(ta), float(tb), dtype=np.float64,
                                        have_h=levy != "none")
                W = torch.from_numpy(W).reshape(B, d)
                return (W, torch.from_numpy(U).reshape(B, d)) if return_U else W

            def __repr__(self):
                return "CounterPath"

            dtype = property(lambda self: torch.float64)
            device = property(lambda self: torch.device("cpu"))
            shape = property(lambda self: (B, d))
            levy_area_approximation = property(lambda self: levy)

        sde = (problems.DoubleWell(d) if cls == "DoubleWell" else problems.Logistic(d, sde_type)).double()
        sde.sde_type = sde_type
        gen = torch.Generator().manual_seed(sum(map(ord, "poly3_" + name)))
        # (double well: both wells and the barrier; logistic growth: positive populations around the carrying capacity)
        y0 = (2.4 * torch.rand(B, d, generator=gen, dtype=torch.float64) - 1.2 if cls == "DoubleWell"
              else 0.2 + 1.6 * torch.rand(B, d, generator=gen, dtype=torch.float64))
        with torch.no_grad():
            ys = torchsde.sdeint(sde, y0, torch.tensor(ts, dtype=torch.float64), bm=CounterPath(), method=method, dt=dt)
        out = {"problem": cls, "sde_type": sde_type, "method": method, "levy": levy, "entropy": np.int64(entropy),
               "dt": np.float64(dt), "ts": np.asarray(ts), "shape": np.array([B, d, steps]), "y0": y0.numpy(), "ys": ys.numpy()}
        for pname, p in sde.named_parameters():
            out["param__" + pname] = p.detach().numpy()
        np.savez_compressed(os.path.join(HERE, f"recognised_poly3_{name}.npz"), **out)
        print(f"recognised_poly3_{name}.npz  |ys|={np.abs(out['ys']).mean():.4f}")


ADDITIVE_CASES = [
    # name, problem class, sde_type, method, levy, m
    ("exadditive_euler", "AdditiveDecay", "ito", "euler", "none", 3),
    ("exadditive_milstein", "AdditiveDecay", "ito", "milstein", "none", 4),
    ("exadditive_srk", "AdditiveDecay", "ito", "srk", "space-time", 3),
    ("exadditive_midpoint", "AdditiveDecay", "stratonovich", "midpoint", "none", 8),
    ("shared_srk", "AdditiveShared", "ito", "srk", "space-time", 4),
    ("netadditive_euler", "MLPNetAdditive", "ito", "euler", "none", 3),
    ("netadditive_srk", "MLPNetAdditive", "ito", "srk", "space-time", 4),
    ("netadditive_midpoint", "MLPNetAdditive", "stratonovich", "midpoint", "none", 5),
]


def additive_module(cls, d, m, sde_type):
    from workloads import problems
    return {"AdditiveDecay": lambda: problems.AdditiveDecay(d, m, sde_type),
            "AdditiveShared": lambda: problems.AdditiveShared(d, m, sde_type),
            "MLPNetAdditive": lambda: problems.MLPNetAdditive(d, m, sde_type, hidden=8)}[cls]()


def gen_additive():
    """The REAL reference on unchanged additive-noise modules (the shapes of its ExAdditive / NeuralAdditive,
    tests/problems.py:106-132,195-224), float64, increments of the counter generator: what the additive trajectory kernels
    (tsde_trajectory_prog_additive / _mlp_additive) and the oracle must reproduce."""
    from oracle import counter
    B, d, steps, entropy, dt = 40, 12, 16, 626262, 2.0 ** -5

    for name, cls, sde_type, method, levy, m in ADDITIVE_CASES:
        edges = np.arange(steps + 1) * dt
        ts = [0.0, 5 * dt, 7.5 * dt, steps * dt]

        class CounterPath(torchsde.BaseBrownian):
            def __call__(self, ta, tb=None, return_U=False, return_A=False):
                W, U, _ = counter.query(B * m, entropy, edges, float(ta), float(tb), dtype=np.float64,
                                        have_h=levy != "none")
                W = torch.from_numpy(W).reshape(B, m)
                return (W, torch.from_numpy(U).reshape(B, m)) if return_U else W

            def __repr__(self):
                return "CounterPath"

            dtype = property(lambda self: torch.float64)
            device = property(lambda self: torch.device("cpu"))
            shape = property(lambda self: (B, m))
            levy_area_approximation = property(lambda self: levy)

        sde = additive_module(cls, d, m, sde_type).double()
        sde.sde_type = sde_type
        gen = torch.Generator().manual_seed(sum(map(ord, "additive_" + name)))
        y0 = 2.0 * torch.rand(B, d, generator=gen, dtype=torch.float64) - 1.0
        with torch.no_grad():
            ys = torchsde.sdeint(sde, y0, torch.tensor(ts, dtype=torch.float64), bm=CounterPath(), method=method, dt=dt)
        out = {"problem": cls, "sde_type": sde_type, "method": method, "levy": levy, "entropy": np.int64(entropy),
               "dt": np.float64(dt), "ts": np.asarray(ts), "shape": np.array([B, d, steps, m]), "y0": y0.numpy(),
               "ys": ys.numpy()}
        for pname, p in sde.named_parameters():
            out["param__" + pname] = p.detach().numpy()
        np.savez_compressed(os.path.join(HERE, f"recognised_additive_{name}.npz"), **out)
        print(f"recognised_additive_{name}.npz  |ys|={np.abs(out['ys']).mean():.4f}")


if __name__ == "__main__":
    which = sys.argv[1:] or ["timegrid", "solver", "adaptive", "adjoint", "bridge", "brownian_seq", "closed_form",
                             "closed_form_affine", "logqp", "closed_form_adjoint", "closed_form_expr", "double_backward",
                             "adjoint_adaptive", "poly3", "additive"]
    torch.manual_seed(0)
    for w in which:
        globals()["gen_" + w]()

"""Argument validation and defaulting shared by ``sdeint`` and ``sdeint_adjoint``.

Behavioural restatement of the reference's ``check_contract`` / ``parse_return``
(torchsde/_core/sdeint.py:115-300): the same checks in the same order raise ``ValueError`` with the same
messages, the same default method table is applied (:147-156), ``ts`` is converted the same way, each
provider the SDE has is probe-called once for its shapes (:199-236), and a default Brownian motion is
created for ``bm=None`` (:262-270). Runs once per solve; not a performance path.
"""
import warnings

import numpy as np

import torch

from . import sde as sde_lib
from . import timegrid
from .brownian import BrownianInterval
from .settings import LEVY_AREA_APPROXIMATIONS, METHODS, NOISE_TYPES, SDE_TYPES

_RENAMABLE = ("drift", "diffusion", "prior_drift", "drift_and_diffusion", "drift_and_diffusion_prod")

_DEFAULT_METHOD = {
    SDE_TYPES.ito: {NOISE_TYPES.diagonal: METHODS.srk, NOISE_TYPES.additive: METHODS.srk,
                    NOISE_TYPES.scalar: METHODS.srk, NOISE_TYPES.general: METHODS.euler},
    SDE_TYPES.stratonovich: {n: METHODS.midpoint for n in NOISE_TYPES.all()},
}


def handle_unused_kwargs(unused_kwargs, msg=None):
    if len(unused_kwargs) > 0:
        prefix = f"{msg}: " if msg is not None else ""
        warnings.warn(f"{prefix}Unexpected arguments {unused_kwargs}")


def assert_no_grad(names, maybe_tensors):
    for name, value in zip(names, maybe_tensors):
        if torch.is_tensor(value) and value.requires_grad:
            raise ValueError(f"Argument {name} must not require gradient.")


def is_strictly_increasing(ts):
    """misc.is_strictly_increasing of the reference, on ONE host copy of `ts` (the reference's element-wise Python
    comparison of a device tensor synchronises once per output time)."""
    if torch.is_tensor(ts):
        if ts.dim() != 1:
            return all(x < y for x, y in zip(ts[:-1], ts[1:]))       # let the reference's semantics decide
        host = timegrid.ts_to_host(ts) if ts.dtype in timegrid._NP else ts.detach().cpu().double().numpy()
        return bool((host[:-1] < host[1:]).all())
    return all(x < y for x, y in zip(ts[:-1], ts[1:]))


class _ShapeLedger:
    """Collects the batch / state / noise sizes every provider reports and cross-checks them."""

    def __init__(self, noise_type):
        self.diagonal = noise_type == NOISE_TYPES.diagonal
        self.batch, self.state, self.noise = [], [], []

    def vector(self, name, shape):
        if len(shape) != 2:
            raise ValueError(f"{name} must be of shape (batch, state_channels), but got {shape}.")
        self.batch.append(shape[0])
        self.state.append(shape[1])

    def diffusion(self, name, shape):
        if self.diagonal:
            if len(shape) != 2:
                raise ValueError(f"{name} must be of shape (batch, state_channels), but got {shape}.")
            self.batch.append(shape[0])
            self.state.append(shape[1])
            self.noise.append(shape[1])
        else:
            if len(shape) != 3:
                raise ValueError(f"{name} must be of shape (batch, state_channels, noise_channels), but got {shape}.")
            self.batch.append(shape[0])
            self.state.append(shape[1])
            self.noise.append(shape[2])

    def require_noise_size(self):
        if not self.noise:
            raise ValueError("Cannot infer noise size (i.e. number of Brownian motion channels). Either pass `bm` "
                             "explicitly, or specify one of the `g`, `f_and_g` functions.`")

    def verify(self):
        for label, values in (("Batch", self.batch), ("State", self.state), ("Noise", self.noise)):
            if any(v != values[0] for v in values[1:]):
                raise ValueError(f"{label} sizes not consistent.")


def default_method(sde):
    """sdeint.py:146-157."""
    return _DEFAULT_METHOD[sde.sde_type][sde.noise_type]


def default_levy_area_approximation(method):
    """The Levy-area approximation of the Brownian motion `sdeint` builds for `bm=None` (sdeint.py:260-267)."""
    if method == METHODS.srk:
        return LEVY_AREA_APPROXIMATIONS.space_time
    if method == METHODS.log_ode_midpoint:
        return LEVY_AREA_APPROXIMATIONS.foster
    return LEVY_AREA_APPROXIMATIONS.none


def check_contract(sde, y0, ts, bm, method, adaptive, options, names, logqp, bm_dt=None, bm_row_offset=0):
    """Returns (ForwardSDE, y0, ts, bm, method, options) or raises ValueError.

    ``bm_dt`` / ``bm_row_offset`` only affect the default Brownian motion built for ``bm=None``.
    """
    renames = {} if names is None else {k: names[k] for k in _RENAMABLE if k in names}
    if renames:
        sde = sde_lib.RenameMethodsSDE(sde, **renames)

    if not hasattr(sde, "noise_type"):
        raise ValueError("sde does not have the attribute noise_type.")
    if sde.noise_type not in NOISE_TYPES:
        raise ValueError(f"Expected noise type in {NOISE_TYPES}, but found {sde.noise_type}.")
    if not hasattr(sde, "sde_type"):
        raise ValueError("sde does not have the attribute sde_type.")
    if sde.sde_type not in SDE_TYPES:
        raise ValueError(f"Expected sde type in {SDE_TYPES}, but found {sde.sde_type}.")
    if not torch.is_tensor(y0):
        raise ValueError("`y0` must be a torch.Tensor.")
    if y0.dim() != 2:
        raise ValueError("`y0` must be a 2-dimensional tensor of shape (batch, channels).")

    if logqp:  # v0.1.1 compatibility: carry the KL integrand as an extra state column
        sde = sde_lib.SDELogqp(sde)
        y0 = torch.cat((y0, y0.new_zeros(size=(y0.size(0), 1))), dim=1)

    if method is None:
        method = _DEFAULT_METHOD[sde.sde_type][sde.noise_type]
    if method not in METHODS:
        raise ValueError(f"Expected method in {METHODS}, but found {method}.")

    if not torch.is_tensor(ts):
        if not isinstance(ts, (tuple, list)) or not all(isinstance(t, (float, int)) for t in ts):
            raise ValueError("Evaluation times `ts` must be a 1-D Tensor or list/tuple of floats.")
        if y0.dtype in timegrid._NP:
            host = np.asarray(ts, dtype=timegrid._NP[y0.dtype])
            ts = torch.from_numpy(host.copy()).to(y0.device)
            timegrid.remember(ts, host)
        else:
            ts = torch.tensor(ts, dtype=y0.dtype, device=y0.device)
    if not is_strictly_increasing(ts):
        raise ValueError("Evaluation times `ts` must be strictly increasing.")

    ledger = _ShapeLedger(sde.noise_type)
    ledger.batch.append(y0.size(0))
    ledger.state.append(y0.size(1))
    if bm is not None:
        if len(bm.shape) != 2:
            raise ValueError("`bm` must be of shape (batch, noise_channels).")
        ledger.batch.append(bm.shape[0])
        ledger.noise.append(bm.shape[1])

    has_drift = has_diffusion = False
    if hasattr(sde, "f"):
        has_drift = True
        ledger.vector("Drift", tuple(sde.f(ts[0], y0).size()))
    if hasattr(sde, "g"):
        has_diffusion = True
        ledger.diffusion("Diffusion", tuple(sde.g(ts[0], y0).size()))
    if hasattr(sde, "f_and_g"):
        has_drift = has_diffusion = True
        f_probe, g_probe = sde.f_and_g(ts[0], y0)
        ledger.vector("Drift", tuple(f_probe.size()))
        ledger.diffusion("Diffusion", tuple(g_probe.size()))
    if hasattr(sde, "g_prod"):
        has_diffusion = True
        ledger.require_noise_size()
        v = torch.randn(ledger.batch[0], ledger.noise[0], dtype=y0.dtype, device=y0.device)
        ledger.vector("Diffusion-vector product", tuple(sde.g_prod(ts[0], y0, v).size()))
    if hasattr(sde, "f_and_g_prod"):
        has_drift = has_diffusion = True
        ledger.require_noise_size()
        v = torch.randn(ledger.batch[0], ledger.noise[0], dtype=y0.dtype, device=y0.device)
        f_probe, gp_probe = sde.f_and_g_prod(ts[0], y0, v)
        ledger.vector("Drift", tuple(f_probe.size()))
        ledger.vector("Diffusion-vector product", tuple(gp_probe.size()))

    if not has_drift:
        raise ValueError("sde must define at least one of `f`, `f_and_g`, or `f_and_g_prod`. (Or possibly more "
                         "depending on the method chosen.)")
    if not has_diffusion:
        raise ValueError("sde must define at least one of `g`, `f_and_g`, `g_prod` or `f_and_g_prod`. (Or possibly "
                         "more depending on the method chosen.)")
    ledger.verify()
    if sde.noise_type == NOISE_TYPES.scalar and ledger.noise[0] != 1:
        raise ValueError(f"Scalar noise must have only one channel; the diffusion has {ledger.noise[0]} noise "
                         f"channels.")

    sde = sde_lib.ForwardSDE(sde)

    if bm is None:
        levy = default_levy_area_approximation(method)
        # Like the reference, no `dt` hint: a fixed-step solve will hand its own time grid to the
        # Brownian motion (BrownianInterval.adopt_grid), which is exact for any dt and dtype.
        bm = BrownianInterval(t0=ts[0], t1=ts[-1], size=(ledger.batch[0], ledger.noise[0]), dtype=y0.dtype,
                              device=y0.device, levy_area_approximation=levy, dt=bm_dt, row_offset=bm_row_offset)

    options = {} if options is None else options.copy()
    if not options.get("overlap_f_g", True):
        sde.overlap_f_g = False     # drift and diffusion recorded in sequence inside a captured graph (sde.py)

    if adaptive and method == METHODS.euler and sde.noise_type != NOISE_TYPES.additive:
        warnings.warn("Numerical solution is not guaranteed to converge to the correct solution when using adaptive "
                      "time-stepping with the Euler--Maruyama method with non-additive noise.")

    return sde, y0, ts, bm, method, options


def parse_return(y0, ys, extra_solver_state, extra, logqp):
    """Split off the log-ratio column if ``logqp`` and append the solver state if ``extra``."""
    if not logqp:
        return (ys, extra_solver_state) if extra else ys
    ys, log_ratio = ys.split(split_size=(y0.size(1) - 1, 1), dim=2)
    increments = (log_ratio[1:] - log_ratio[:-1]).squeeze(dim=2)
    return (ys, increments, extra_solver_state) if extra else (ys, increments)

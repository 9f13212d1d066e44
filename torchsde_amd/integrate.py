"""``sdeint``: the forward solve entry point (same signature and return convention as the reference's
torchsde/_core/sdeint.py:27-112)."""
import torch

from . import _native
from . import contract
from . import solvers


def sdeint(sde, y0, ts, bm=None, method=None, dt=1e-3, adaptive=False, rtol=1e-5, atol=1e-4, dt_min=1e-5,
           options=None, names=None, logqp=False, extra=False, extra_solver_state=None, **unused_kwargs):
    """Numerically integrate an SDE on the GPU.

    Args and returns are those of ``torchsde.sdeint``: ``ys`` of shape (T, batch, d); with ``logqp=True``
    also the (T-1, batch) log-ratio increments; with ``extra=True`` also the solver's extra state.
    Raises ``ValueError`` for every contract violation the reference rejects.
    """
    contract.handle_unused_kwargs(unused_kwargs, msg="`sdeint`")
    del unused_kwargs
    with _native.on_device_of(y0 if torch.is_tensor(y0) else "cpu"):
        return _sdeint(sde, y0, ts, bm, method, dt, adaptive, rtol, atol, dt_min, options, names, logqp, extra,
                       extra_solver_state)


def _sdeint(sde, y0, ts, bm, method, dt, adaptive, rtol, atol, dt_min, options, names, logqp, extra,
            extra_solver_state):
    sde, y0, ts, bm, method, options = contract.check_contract(sde, y0, ts, bm, method, adaptive, options, names,
                                                               logqp)
    contract.assert_no_grad(["ts", "dt", "rtol", "atol", "dt_min"], [ts, dt, rtol, atol, dt_min])

    solver_cls = solvers.select(method=method, sde_type=sde.sde_type)
    solver = solver_cls(sde=sde, bm=bm, dt=dt, adaptive=adaptive, rtol=rtol, atol=atol, dt_min=dt_min,
                        options=options)
    if hasattr(solver, "wants_extra"):
        solver.wants_extra = bool(extra)
    if extra_solver_state is None:
        extra_solver_state = solver.init_extra_solver_state(ts[0], y0)
    if y0.numel() == 0:      # an empty batch: nothing to launch (the reference's loop runs on empty tensors)
        ys = y0.unsqueeze(0).repeat(len(ts), *([1] * y0.dim()))
        return contract.parse_return(y0, ys, tuple(extra_solver_state), extra, logqp)
    ys, extra_solver_state = solver.integrate(y0, ts, extra_solver_state)
    return contract.parse_return(y0, ys, extra_solver_state, extra, logqp)

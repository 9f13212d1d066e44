"""``sdeint_adjoint`` on the perceptron-drift module (closed_form.MLPDriftDiagonalSDE) through the matrix-core kernels.

What the reference computes for ``sdeint_adjoint(sde, y0, ts, method=..., adjoint_method="euler")`` on a diagonal-noise
SDE (torchsde/_core/adjoint.py:31-127 + adjoint_sde.py:177-230, 296-323 + methods/euler.py:29-37): a forward solve that
keeps only the states at the output times, and a backward Euler-Maruyama solve of the augmented state (y, a_y, a_theta)
on the same Brownian path, y reset to the stored state and the output's cotangent added to a_y at every output time
(adjoint.py:114-116). Here the forward solve is one launch of the sampling kernel (``tsde_trajectory_mlp_diag``) and
the backward solve is ``tsde_adjoint_mlp_diag`` -- y reconstructed and a_y advanced in registers, four matrix products
per step on the MFMA units -- per chunk of steps, each followed by the two tall-K weight-gradient products
(``tsde_gram_partials``) over that chunk's stash. Memory: the outputs plus O(chunk) stash, independent of the step
count. Same Brownian path and the same arithmetic structure as the stepwise adjoint (adjoint.py), which remains the
route for every other module, method and grid, and the parity reference for this one (tests/test_gpu_mlp_adjoint.py).
"""
import numpy as np
import torch

from . import _native
from . import closed_form
from . import kernels as K
from . import timegrid
from .brownian import BrownianInterval
from .settings import METHOD_OPTIONS, METHODS, SDE_TYPES

# forward methods the sampling kernel has. (Backward: Euler-Maruyama is an Ito scheme -- the reference rejects it for a
# Stratonovich adjoint SDE, adjoint.py:83-93, and `_check_adjoint_method` has already done the same -- Milstein steps
# exist for both SDE types.)
_FORWARD_CODES = {
    (METHODS.euler, SDE_TYPES.ito): _native.TRAJ_EULER,
    (METHODS.milstein, SDE_TYPES.ito): _native.TRAJ_MILSTEIN_ITO,
    (METHODS.milstein, SDE_TYPES.stratonovich): _native.TRAJ_MILSTEIN_STRAT,
    (METHODS.midpoint, SDE_TYPES.stratonovich): _native.TRAJ_MIDPOINT,
    (METHODS.srk, SDE_TYPES.ito): _native.TRAJ_SRK,        # the reference's default for diagonal Ito noise
}
_BACKWARD_KINDS = {METHODS.euler: "euler", METHODS.milstein: "milstein"}


def adjoint_mlp_diag(y, a, stashes, row_rate, row_shift, w1, b1, w2, b2, rate, shift, diffusion, activation, ito,
                     schedule, k_lo, k_hi, bm, milstein=False):
    """One launch of ``tsde_adjoint_mlp_diag`` over steps k_hi-1 ... k_lo (y, a updated in place)."""
    stash_a, stash_hid, stash_delta, stash_y = stashes
    rows, d = y.shape
    lib, dt_code, stream = K._launch_env(y)
    entropy_dev = bm._entropy_dev
    code = lib.tsde_adjoint_mlp_diag(
        y.data_ptr(), a.data_ptr(), stash_a.data_ptr(), stash_hid.data_ptr(), stash_delta.data_ptr(), stash_y.data_ptr(),
        row_rate.data_ptr(), row_shift.data_ptr(), rows, d, b1.numel(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
        b2.data_ptr(), rate.data_ptr(), shift.data_ptr(), int(diffusion[0]), float(diffusion[1]), int(activation),
        (1 if ito else 0) | (2 if milstein else 0), schedule.struct(), int(k_lo), int(k_hi), bm._key, bm._elem0,
        None if entropy_dev is None else entropy_dev.data_ptr(), dt_code, stream)
    _native.check(code, "tsde_adjoint_mlp_diag")


class _MlpAdjointFn(torch.autograd.Function):
    """Forward: the sampling kernel, outputs only. Backward: the stochastic adjoint (Euler or Milstein steps) on the
    matrix cores."""

    STASH_BYTES = 3 << 30      # per-chunk stash of the backward sweep (four (steps, rows, width) float32 arrays)

    @staticmethod
    def forward(ctx, activation, diffusion, method_code, ito, backward_kind, schedule, backward_schedule, out_steps, bm,
                y0, w1, b1, w2, b2, rate, shift):
        rows, d = y0.shape
        y0c = _native.contiguous(y0.detach())
        coefs = [p.detach().reshape(-1).expand(d).contiguous() for p in (rate, shift)]
        w1_in = w1.detach().t().contiguous()                # (d, hidden): input-major, as the kernels read it
        w2_in = w2.detach().t().contiguous()                # (hidden, d)
        b1c, b2c = b1.detach().contiguous(), b2.detach().contiguous()
        ys = torch.empty((len(out_steps) + 1, rows, d), dtype=y0.dtype, device=y0.device)
        ys[0].copy_(y0c)
        K.trajectory_mlp_diag(ys[1:], y0c, w1_in, b1c, w2_in, b2c, coefs[0], coefs[1], activation, diffusion, method_code,
                              schedule, bm)
        ctx.save_for_backward(ys, w1_in, b1c, w2_in, b2c, coefs[0], coefs[1])
        ctx.activation, ctx.diffusion, ctx.ito = int(activation), (int(diffusion[0]), float(diffusion[1])), bool(ito)
        ctx.backward_kind = backward_kind
        ctx.schedule, ctx.bm, ctx.out_steps = backward_schedule, bm, tuple(out_steps)
        ctx.param_shapes = (tuple(rate.shape), tuple(shift.shape))
        ctx.generic = None
        return ys

    @staticmethod
    def backward(ctx, gys):
        if torch.is_grad_enabled():    # create_graph=True: the kernels below leave no graph; see adjoint_double.py
            return _MlpAdjointFn._backward_with_graph(ctx, gys)
        ys, w1_in, b1c, w2_in, b2c, rate, shift = ctx.saved_tensors
        rows, d = ys.shape[1], ys.shape[2]
        hidden = b1c.numel()
        dev = ys.device
        gys = _native.contiguous(gys)
        per_step = rows * (2 * d + 2 * hidden) * 4
        chunk = int(max(1, min(ctx.schedule.n_steps, _MlpAdjointFn.STASH_BYTES // max(per_step, 1))))
        stashes = (torch.empty((chunk, rows, d), dtype=torch.float32, device=dev),
                   torch.empty((chunk, rows, hidden), dtype=torch.float32, device=dev),
                   torch.empty((chunk, rows, hidden), dtype=torch.float32, device=dev),
                   torch.empty((chunk, rows, d), dtype=torch.float32, device=dev))
        row_rate = torch.zeros((rows, d), dtype=torch.float32, device=dev)
        row_shift = torch.zeros_like(row_rate)
        g_w1 = torch.zeros((hidden, d), dtype=torch.float32, device=dev)
        g_w2 = torch.zeros((d, hidden), dtype=torch.float32, device=dev)
        g_b1 = torch.zeros(hidden, dtype=torch.float32, device=dev)
        g_b2 = torch.zeros(d, dtype=torch.float32, device=dev)
        boundaries = (0,) + ctx.out_steps                   # step boundary of output i
        y = ys[-1].clone()
        a = gys[-1].clone()
        for i in range(len(boundaries) - 1, 0, -1):
            for k_hi in range(boundaries[i], boundaries[i - 1], -chunk):
                k_lo = max(boundaries[i - 1], k_hi - chunk)
                n = k_hi - k_lo
                adjoint_mlp_diag(y, a, stashes, row_rate, row_shift, w1_in, b1c, w2_in, b2c, rate, shift, ctx.diffusion,
                                 ctx.activation, ctx.ito, ctx.schedule, k_lo, k_hi, ctx.bm,
                                 milstein=ctx.backward_kind == "milstein")
                flat_a = stashes[0][:n].reshape(n * rows, d)
                flat_hid = stashes[1][:n].reshape(n * rows, hidden)
                flat_delta = stashes[2][:n].reshape(n * rows, hidden)
                flat_y = stashes[3][:n].reshape(n * rows, d)
                for g_w, g_b, lhs, rhs in ((g_w2, g_b2, flat_a, flat_hid), (g_w1, g_b1, flat_delta, flat_y)):
                    weight, bias = K.gram(lhs, rhs, column_sums=True)
                    g_w += weight
                    g_b += bias
            # adjoint.py:114-116: the forward state is known again at an output time; its cotangent joins a_y
            y.copy_(ys[i - 1])
            a += gys[i - 1]
        diffusion = []
        for acc, shape in zip((row_rate, row_shift), ctx.param_shapes):
            per_channel = acc.sum(dim=0)
            diffusion.append(per_channel.reshape(shape) if int(np.prod(shape, dtype=np.int64)) == d and len(shape) == 1
                             else per_channel.sum().reshape(shape))
        grad_y0 = a if ctx.needs_input_grad[9] else None
        return (None,) * 9 + (grad_y0, g_w1, g_b1, g_w2, g_b2, diffusion[0], diffusion[1])


    @staticmethod
    def _backward_with_graph(ctx, gys):
        from . import adjoint, adjoint_double
        if ctx.generic is None:
            raise NotImplementedError("torchsde_amd: no differentiable backward pass was prepared for this call.")
        sde, ts_host, dt, own = ctx.generic
        ys = ctx.saved_tensors[0]
        params = [p for p in own if p.requires_grad]
        with _native.on_device_of(ys):
            plan = adjoint._plan_backward(ts_host, dt, ctx.bm, ys.device)
            a_y, a_theta = adjoint_double.run(adjoint.AdjointSDE(sde, params), ctx.backward_kind, ctx.bm, plan, ys, gys)
        a_theta = iter(a_theta)
        grads = [next(a_theta) if p.requires_grad else None for p in own]
        return (None,) * 9 + (a_y if ctx.needs_input_grad[9] else None, *grads)


def route(sde, y0, ts, bm, method, adjoint_method, dt, adaptive, adjoint_adaptive, options, adjoint_options,
          adjoint_params, extra_solver_state, solver=None):
    """`ys` with a grad_fn towards y0 and the module's six parameters if this call can take the kernels above, else
    None (the caller then runs the stepwise stochastic adjoint)."""
    from .sde import ForwardSDE
    if (adaptive or adjoint_adaptive or extra_solver_state is not None or adjoint_method not in _BACKWARD_KINDS
            or adjoint_options.get(METHOD_OPTIONS.grad_free, False) or options.get(METHOD_OPTIONS.grad_free, False)
            or not options.get("trajectory_kernel", True) or not adjoint_options.get("trajectory_kernel", True)):
        return None
    base = getattr(sde, "_base_sde", None)
    if type(sde) is not ForwardSDE:
        return None
    published = hasattr(base, "closed_form") and closed_form.publishes_its_own_dynamics(base)
    code = _FORWARD_CODES.get((method, sde.sde_type))
    if (code is None or not isinstance(bm, BrownianInterval) or bm._rootW is not None or bm._rootH is not None
            or bm._snap or y0.dim() != 2 or tuple(bm.shape) != tuple(y0.shape) or y0.dtype != torch.float32
            or bm.dtype != torch.float32 or bm._elem0 % 4 != 0 or y0.numel() == 0
            or (code == _native.TRAJ_SRK and not bm._have_H)):
        return None
    if published:
        spec = base.closed_form(y0.shape[1], y0.dtype, y0.device)
        if spec is None or spec[0] != "mlp_diagonal":
            return None
        own = list(base.closed_form_parameters())
    else:
        # an UNCHANGED user module whose drift turns out to be lin2(act(lin1(y))) with an affine / sigmoid diagonal
        # diffusion (recognise.py), its forward solve through the sampling kernel verified against the stepwise one
        if solver is None or not isinstance(base, torch.nn.Module):
            return None
        found = solver.recognised_perceptron(y0, ts)
        if found is None:
            return None
        own = found.perceptron_parameters()
        if own is None:
            return None
        spec = found.perceptron_spec()
    hidden = own[1].numel()
    # gradients go to exactly the module's own trainable parameters: a narrower or wider `adjoint_params` is the
    # stepwise adjoint's business
    module_params = {id(p) for p in base.parameters()}
    if ({id(p) for p in adjoint_params} != {id(p) for p in own if p.requires_grad}
            or module_params != {id(p) for p in own if id(p) in module_params or published}
            or hidden % 4 != 0 or y0.shape[0] * max(y0.shape[1], hidden) >= 2 ** 30):
        return None

    # ---- grids: every output on a forward step boundary, every backward step exactly one forward cell ------------------
    ts_host = timegrid.ts_to_host(ts)
    grid = timegrid.build(ts_host, dt)
    if grid.n_steps == 0 or any(not (w0 == 0.0 and w1 == 1.0) for (_, _, w0, w1) in grid.outputs):
        return None
    t64 = grid.t_f64()
    bm.adopt_grid(t64)
    cells = bm.match_grid(t64)
    if cells is None:
        return None
    cells = np.asarray(cells, dtype=np.int64)
    out_steps = [kc for (_, kc, _, _) in grid.outputs]
    np_dtype = grid.t.dtype.type
    h = bm._edges[cells + 1] - bm._edges[cells]

    def rows_for(dts):
        rows = np.zeros((grid.n_steps, 8), dtype=np.float64)
        rows[:, 0] = dts
        rows[:, 1] = np_dtype(0.5) * dts
        rows[:, 2] = np_dtype(1) / dts
        rows[:, 3] = np.sqrt(dts)
        rows[:, 4] = np.sqrt(h)
        rows[:, 5] = np.sqrt(h / 12.0)
        rows[:, 6] = h
        return rows

    # the backward solver builds its own grid on every [-ts[i], -ts[i-1]] (adjoint.py:97-112): its steps must be the
    # forward cells walked backwards, and it is ITS step sizes the backward kernel is given
    backward_dt = np.array(grid.dt, dtype=grid.dt.dtype)
    boundaries = [0] + out_steps
    for i in range(len(ts_host) - 1, 0, -1):
        back = timegrid.build(np.array([-ts_host[i], -ts_host[i - 1]], dtype=ts_host.dtype), dt)
        k_lo, k_hi = boundaries[i - 1], boundaries[i]
        if back.n_steps != k_hi - k_lo:
            return None
        walked = bm.match_grid(-back.t_f64()[::-1])
        if walked is None or not np.array_equal(np.asarray(walked, dtype=np.int64), cells[k_lo:k_hi]):
            return None
        backward_dt[k_lo:k_hi] = back.dt[::-1]
    schedule = K.TrajectorySchedule.cached(rows_for(grid.dt), cells, out_steps, [(0.0, 1.0)] * len(out_steps), y0.device,
                                           y0.dtype)
    backward_schedule = K.TrajectorySchedule.cached(rows_for(backward_dt), cells, out_steps,
                                                    [(0.0, 1.0)] * len(out_steps), y0.device, y0.dtype)
    ys = _MlpAdjointFn.apply(spec[-2], tuple(spec[-1]), code, sde.sde_type == SDE_TYPES.ito,
                             _BACKWARD_KINDS[adjoint_method], schedule, backward_schedule,
                             tuple(int(k) for k in out_steps), bm, y0, *own)
    if ys.grad_fn is not None:           # what a second-order backward pass needs (`ys.grad_fn` is the Function's ctx)
        ys.grad_fn.generic = (sde, ts_host, dt, own)
    return ys

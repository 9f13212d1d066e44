"""``sdeint_adjoint``: O(1)-memory gradients via the stochastic adjoint, with a fused augmented-state update.

Same signature, defaults, warnings and errors as the reference's ``sdeint_adjoint``
(torchsde/_core/adjoint.py:130-296). The mathematics of the backward pass is the reference's
(adjoint.py:64-127 + adjoint_sde.py:23-377): integrate, backwards in time and on the SAME Brownian path, the
augmented state (y, a_y, a_theta) whose drift is (-f~, a^T df~/dy, a^T df~/dtheta) and whose
diffusion-vector product is (-g.v, a^T d(g.v)/dy, a^T d(g.v)/dtheta); for Ito SDEs f~ carries the
double-Stratonovich correction (adjoint_sde.py:130-216); at every output time reset y to the stored forward
value and add the incoming gradient to a_y (adjoint.py:114-116).

What differs is the program: the reference flattens the augmented state into one (1, 2Bd+P) tensor and
re-packs it (cat/split) around every VJP and every output time, and pushes it through the generic solver
(~10 elementwise kernels per step); here the three segments stay unpacked in persistent buffers, the VJPs are
taken directly on (y, theta) with ``torch.autograd.grad`` (user code stays user code), and ONE kernel launch
(``tsde_aug_update``) applies the solver update to y, a_y and every parameter segment. The Brownian increments
of the backward sweep are re-materialised from the counter RNG (no tree, no cache to thrash).
"""
import warnings

import numpy as np
import torch
from torch import nn

from . import contract
from . import kernels as K
from . import solvers
from . import timegrid
from .brownian import BrownianInterval, ReverseBrownian
from .kernels import NoiseSpec
from .sde import BaseSDE, jvp, vjp
from .settings import METHOD_OPTIONS, METHODS, NOISE_TYPES, SDE_TYPES

_ADJOINT_NOISE = {NOISE_TYPES.general: NOISE_TYPES.general, NOISE_TYPES.additive: NOISE_TYPES.general,
                  NOISE_TYPES.scalar: NOISE_TYPES.scalar, NOISE_TYPES.diagonal: NOISE_TYPES.diagonal}


class AdjointSDE(BaseSDE):
    """The augmented backward SDE of ``forward_sde`` on the unpacked state (y, a_y, a_theta...).

    Methods take the *forward* time ``t`` (= minus the backward solver's time, adjoint_sde.py:240,298) and
    return lists ``[part_y, part_a, *parts_theta]`` holding f~, g.v (un-negated; the update kernel applies the
    reference's minus sign to the y segment) and the VJPs.
    """

    def __init__(self, forward_sde, params, shapes=None):
        super().__init__(sde_type=forward_sde.sde_type, noise_type=_ADJOINT_NOISE[forward_sde.noise_type])
        self.forward_sde = forward_sde
        self.params = list(params)
        self._shapes = shapes
        ito = forward_sde.sde_type == SDE_TYPES.ito
        kind = forward_sde.noise_type
        if not ito or kind == NOISE_TYPES.additive:
            self._correction = None
        elif kind == NOISE_TYPES.diagonal:
            self._correction = "diagonal"
        else:
            self._correction = "columns"

    # -- pieces ---------------------------------------------------------------------------------------
    def _inputs(self, y):
        return [y] + self.params

    def _drift_parts(self, f, g, y, a):
        """(f~, vjp_y, vjp_theta...) -- adjoint_sde.py:111-216."""
        if self._correction is None:
            grads = vjp(f, self._inputs(y), grad_outputs=a, allow_unused=True, retain_graph=True)
            return [f.detach()] + grads
        if self._correction == "diagonal":
            g_dg, = vjp(g, y, grad_outputs=g, allow_unused=True, create_graph=True)
            f = f - g_dg                                   # double Stratonovich correction
            grads = vjp(f, self._inputs(y), grad_outputs=a, allow_unused=True, retain_graph=True)
            a_dg, = vjp(g, y, grad_outputs=a, allow_unused=True, retain_graph=True)
            extra = vjp(g, self._inputs(y), grad_outputs=a_dg, allow_unused=True, retain_graph=True)
            return [f.detach()] + [p + q for p, q in zip(grads, extra)]
        columns = [c.squeeze(dim=-1) for c in g.split(1, dim=-1)]
        dg_g = sum(jvp(c, y, grad_inputs=c, allow_unused=True, create_graph=True)[0] for c in columns)
        f = f - dg_g
        grads = vjp(f, self._inputs(y), grad_outputs=a, allow_unused=True, retain_graph=True)
        for c in columns:
            a_dg, = vjp(c, y, grad_outputs=a, allow_unused=True, retain_graph=True)
            extra = vjp(c, self._inputs(y), grad_outputs=a_dg, allow_unused=True, retain_graph=True)
            grads = [p + q for p, q in zip(grads, extra)]
        return [f.detach()] + grads

    def _diffusion_parts(self, g_prod, y, a):
        """(g.v, vjp_y, vjp_theta...) -- adjoint_sde.py:218-230."""
        grads = vjp(g_prod, self._inputs(y), grad_outputs=a, allow_unused=True, retain_graph=True)
        return [g_prod.detach()] + grads

    @staticmethod
    def _leaf(y):
        return y.detach().requires_grad_(True)

    # -- what the backward solvers call ------------------------------------------------------------------
    def f_and_g_prod(self, t, y, a, v):
        """Drift and diffusion-product parts at forward time t (adjoint_sde.py:296-323)."""
        fwd = self.forward_sde
        y = self._leaf(y)
        with torch.enable_grad():
            if self._correction is None:
                f, g_prod = fwd.f_and_g_prod(t, y, v)
                g = None
            else:
                f, g = fwd.f_and_g(t, y)
                g_prod = fwd.prod(g, v)
            return self._drift_parts(f, g, y, a), self._diffusion_parts(g_prod, y, a)

    def fused_terms(self, t, y, a, v, cF, cG):
        """(f~, g.v, cF * vjp(f~) + cG * vjp(g.v)) for the Euler / midpoint backward updates.

        Same quantities as `f_and_g_prod`, but the cotangent-weighted sum over the adjoint segments is taken
        in ONE reverse sweep: ``autograd.grad([f~, g.v, g], [y, theta], [cF*a, cG*a, cF*a_dg])`` instead of one
        sweep per term (3 sweeps instead of the reference's 5 for Ito-diagonal, 1 instead of 2 otherwise).
        """
        fwd = self.forward_sde
        y = self._leaf(y)
        inputs = self._inputs(y)
        with torch.enable_grad():
            if self._correction is None:
                f, g_prod = fwd.f_and_g_prod(t, y, v)
                outs, cots = [f, g_prod], [a * cF, a * cG]
            elif self._correction == "diagonal":
                f, g = fwd.f_and_g(t, y)
                g_prod = fwd.prod(g, v)
                g_dg, = vjp(g, y, grad_outputs=g, allow_unused=True, create_graph=True)
                f = f - g_dg
                a_dg, = vjp(g, y, grad_outputs=a, allow_unused=True, retain_graph=True)
                outs, cots = [f, g_prod, g], [a * cF, a * cG, a_dg * cF]
            else:
                parts_f, parts_g = self.f_and_g_prod(t, y, a, v)
                return parts_f[0], parts_g[0], [p * cF + q * cG for p, q in zip(parts_f[1:], parts_g[1:])]
            live = [(o, c) for o, c in zip(outs, cots) if o.requires_grad]
            if live:
                grads = torch.autograd.grad([o for o, _ in live], inputs, grad_outputs=[c for _, c in live],
                                            allow_unused=True)
            else:
                grads = [None] * len(inputs)
            grads = [torch.zeros_like(x) if gr is None else gr for gr, x in zip(grads, inputs)]
            return f.detach(), g_prod.detach(), grads

    def g_prod(self, t, y, a, v):
        """Diffusion-product parts only (adjoint_sde.py:283-287): what Euler-Heun's second stage asks for."""
        y = self._leaf(y)
        with torch.enable_grad():
            return self._diffusion_parts(self.forward_sde.g_prod(t, y, v), y, a)

    def f(self, t, y, a):
        """Drift parts only (adjoint_sde.py:236-252)."""
        fwd = self.forward_sde
        y = self._leaf(y)
        with torch.enable_grad():
            if self._correction is None:
                return self._drift_parts(fwd.f(t, y), None, y, a)
            f, g = fwd.f_and_g(t, y)
            return self._drift_parts(f, g, y, a)

    def g_prod_and_gdg_prod(self, t, y, a, v1, v2):
        """Milstein pieces of the adjoint of a diagonal-noise SDE (adjoint_sde.py:332-377)."""
        if self.forward_sde.noise_type != NOISE_TYPES.diagonal:
            raise NotImplementedError
        fwd = self.forward_sde
        y = self._leaf(y)
        inputs = self._inputs(y)
        with torch.enable_grad():
            g = fwd.g(t, y)
            g_prod = fwd.prod(g, v1)
            vg_dg, = vjp(g, y, grad_outputs=v2 * g, allow_unused=True, retain_graph=True)
            dgdy, = vjp(g.sum(), y, allow_unused=True, retain_graph=True)
            prod_partials = vjp(g, inputs, grad_outputs=a * v2 * dgdy, allow_unused=True, retain_graph=True)
            avg_dg, = vjp(g, y, grad_outputs=(a * v2 * g).detach(), allow_unused=True, create_graph=True)
            mixed_partials = vjp(avg_dg.sum(), inputs, allow_unused=True, retain_graph=True)
            gdg_parts = [vg_dg] + [p - q for p, q in zip(prod_partials, mixed_partials)]
            return self._diffusion_parts(g_prod, y, a), gdg_parts

    # the reference never evaluates these for an adjoint SDE (adjoint_sde.py:254-281)
    def g(self, t, y):
        raise RuntimeError("Adjoint `g` not defined. Please report a bug to torchsde.")

    def f_and_g(self, t, y):
        raise RuntimeError("Adjoint `f_and_g` not defined. Please report a bug to torchsde.")

    def prod(self, g, v):
        raise RuntimeError("Adjoint `prod` not defined. Please report a bug to torchsde.")


class _AugState:
    """The augmented state as separate persistent buffers: [y, a_y, a_theta_0, ...]."""

    def __init__(self, tensors):
        self.t = list(tensors)

    @classmethod
    def like(cls, other):
        return cls([torch.empty_like(x) for x in other.t])


_SEG_CACHE = {}


def _update(dst, src, F, G, D, cF, cG):
    """dst = src + F*cF + cG*G (+ D), with the reference's sign on the y segment; ONE kernel launch.

    The (dst, src) buffer sets are persistent, so their descriptor array is built once and only the term
    pointers are refreshed per step (this runs ~1000 times per backward pass)."""
    from . import _native
    key = (id(dst), id(src))
    entry = _SEG_CACHE.get(key)
    if entry is None or entry[1] is not dst or entry[2] is not src:
        arr = (_native.Seg * len(src.t))()
        for i, s in enumerate(src.t):
            arr[i].out, arr[i].s, arr[i].n = dst.t[i].data_ptr(), s.data_ptr(), s.numel()
            sign = -1.0 if i == 0 else 1.0
            arr[i].sF, arr[i].sG, arr[i].sD = sign, sign, 1.0
        if len(_SEG_CACHE) > 64:
            _SEG_CACHE.clear()
        entry = (arr, dst, src)
        _SEG_CACHE[key] = entry
    arr = entry[0]
    keep = []
    for i, s in enumerate(src.t):
        for name, terms in (("F", F), ("G", G), ("D", D)):
            t = None if terms is None else terms[i]
            if t is None:
                setattr(arr[i], name, None)
                continue
            if t.dtype != s.dtype:
                t = t.to(s.dtype)
            if t.shape != s.shape:
                t = t.reshape(s.shape) if t.numel() == s.numel() else t.expand(s.shape)
            if not t.is_contiguous():
                t = t.contiguous()
            keep.append(t)
            setattr(arr[i], name, t.data_ptr())
    lib, dt_code, stream = K._launch_env(src.t[0])
    code = lib.tsde_aug_update(arr, len(src.t), float(cF), float(cG), dt_code, stream)
    if code:
        _native.check(code, "tsde_aug_update")
    return keep


def _check_adjoint_method(adjoint_sde, adjoint_method, adjoint_options, bm):
    """The compatibility errors the reference raises when it builds the backward solver (adjoint.py:83-93)."""
    if adjoint_method == METHODS.adjoint_reversible_heun:
        if adjoint_sde.sde_type != SDE_TYPES.stratonovich:
            raise ValueError(f"SDE is of type {adjoint_sde.sde_type} but solver is for type {SDE_TYPES.stratonovich}")
        return _ReversibleHeunBackward
    cls = solvers.select(adjoint_method, adjoint_sde.sde_type)
    if cls is solvers.LogODEMidpoint:    # methods/log_ode.py:32-35
        raise ValueError("Log-ODE schemes cannot be used for adjoint SDEs, because they require "
                         "direct access to the diffusion, whilst adjoint SDEs rely on a more efficient "
                         "diffusion-vector product. Use a different method instead.")
    if cls is solvers.ReversibleHeun:
        # the reference builds the solver and fails inside backward() when it asks the adjoint SDE for `f_and_g`
        # (methods/reversible_heun.py:58-59 -> adjoint_sde.py:262-264); same error, raised at call time
        raise RuntimeError("Adjoint `f_and_g` not defined: `reversible_heun` cannot integrate an adjoint SDE; use "
                           f"adjoint_method={repr(METHODS.adjoint_reversible_heun)} (with method="
                           f"{repr(METHODS.reversible_heun)}) instead.")
    if cls is solvers.SRK:
        raise ValueError("Stochastic Runge–Kutta methods cannot be used for adjoint SDEs, because it requires "
                         "direct access to the diffusion, whilst adjoint SDEs rely on a more efficient "
                         "diffusion-vector product. Use a different method instead.")
    if issubclass(cls, solvers._Milstein) and adjoint_options.get(METHOD_OPTIONS.grad_free, False):
        raise ValueError(f"Derivative-free Milstein cannot be used for adjoint SDEs, because it requires "
                         f"direct access to the diffusion, whilst adjoint SDEs rely on a more efficient "
                         f"diffusion-vector product. Use derivative-using Milstein instead: "
                         f"`adjoint_options=dict({METHOD_OPTIONS.grad_free}=False)`")
    if adjoint_sde.sde_type != cls.sde_type:
        raise ValueError(f"SDE is of type {adjoint_sde.sde_type} but solver is for type {cls.sde_type}")
    if adjoint_sde.noise_type not in cls.noise_types:
        raise ValueError(f"SDE has noise type {adjoint_sde.noise_type} but solver only supports noise types "
                         f"{cls.noise_types}")
    if bm.levy_area_approximation not in cls.levy_area_approximations:
        raise ValueError(f"SDE solver requires one of {cls.levy_area_approximations} set as the "
                         f"`levy_area_approximation` on the Brownian motion.")
    return cls


class _ReversibleHeunBackward:
    """Marker returned by `_check_adjoint_method` for METHODS.adjoint_reversible_heun."""


def _plan_reversible_heun_backward(ts_host, dt, native, device):
    """Host-side preparation of the reversible-Heun backward sweep: per output interval the reversed grid, the
    forward times of its step boundaries on the device, and the Brownian cell of every step if they line up."""
    intervals = []
    misaligned = False
    for i in range(len(ts_host) - 1, 0, -1):
        grid = timegrid.build(np.array([-ts_host[i], -ts_host[i - 1]], dtype=ts_host.dtype), dt)
        tau64 = grid.t_f64()
        fwd_times = torch.from_numpy((-grid.t).astype(grid.t.dtype)).to(device).unbind(0)   # forward time = -tau
        tau_dev = None
        cells = None
        if native is not None:
            cells = native.match_grid(-tau64[::-1]) if native.frozen else None
            misaligned = misaligned or cells is None
        else:
            tau_dev = torch.from_numpy(grid.t.copy()).to(device).unbind(0)
        intervals.append((i, grid, tau64, fwd_times, cells, tau_dev))
    if misaligned:
        native.locate(float(ts_host[0]), float(ts_host[-1]))
        native._device_edges()
    return intervals


def _reversible_heun_backward_step(sde, params, diag, state, noise, step_dt, t0_fwd, t1_fwd):
    """One backward step of the reversible-Heun pair (reference: methods/reversible_heun.py:98-144): reconstruct
    (y, f, g, z) algebraically while propagating (a_y, a_f, a_g, a_z); one VJP through f_and_g. `state` is
    (y, f, g, z, a_y, a_f, a_g, a_z); returns the next state and the step's parameter cotangents. Diagonal noise runs
    entirely on fused HIP kernels (tsde_rheun_*); other noise types use the contraction kernel for the state and torch
    ops for the (B, d, m) outer products of the adjoint. Nothing is updated in place."""
    y, f, g, z, a_y, a_f, a_g, a_z = state
    half_dt = type(step_dt)(0.5) * step_dt
    if diag:
        a_f0, a_g0 = K.rheun_adj_a(a_y, a_f, a_g, half_dt, noise)
        z1 = K.rheun_z(y, z, f, g, step_dt, -1.0, noise)
    else:
        dW, _ = noise.materialise()
        half_dW = 0.5 * dW
        a_y_half_dW = a_y.unsqueeze(-1) * half_dW.unsqueeze(-2)
        a_f0 = a_f + a_y * half_dt
        a_g0 = a_g + a_y_half_dW
        z1 = K.step_general_weighted(K.lincomb2(y, z, 2.0, -1.0), f, g, -1.0, step_dt, -1.0, 0, 0.0, 0.0, 0.0, noise)
    z_leaf = z.detach().requires_grad_(True)
    with torch.enable_grad():
        re_f, re_g = sde.f_and_g(t0_fwd, z_leaf)
        vjp_z, *vjp_theta = vjp((re_f, re_g), [z_leaf] + params, grad_outputs=[a_f0, a_g0], allow_unused=True)
    f1, g1 = sde.f_and_g(t1_fwd, z1)
    if diag:
        y1 = K.rheun_y(y, f, f1, g, g1, half_dt, -1.0, noise)
        a_y, a_z, a_f, a_g = K.rheun_adj_b(a_y, a_z, vjp_z, step_dt, half_dt, noise)
    else:
        y1 = K.step_general_weighted(y, K.lincomb2(f, f1, 1.0, 1.0), K.lincomb2(g, g1, 1.0, 1.0), -1.0, half_dt, -0.5,
                                     0, 0.0, 0.0, 0.0, noise)
        zz = a_z + vjp_z
        a_f = a_y * half_dt + zz * step_dt
        a_g = a_y_half_dW + zz.unsqueeze(-1) * dW.unsqueeze(-2)
        a_y = a_y + 2 * zz
        a_z = -zz
    return (y1, f1, g1, z1, a_y, a_f, a_g, a_z), vjp_theta


def _run_reversible_heun_backward(sde, bm, params, plan, ys, grad_ys, f, g, z, a_f, a_g, a_z):
    """Exact-gradient backward pass of reversible Heun (reference: methods/reversible_heun.py:98-144 driven by
    adjoint.py:64-127), launch-only part. Returns [a_y, a_f, a_g, a_z, *a_theta]."""
    diag = sde.noise_type == NOISE_TYPES.diagonal
    native = bm if isinstance(bm, BrownianInterval) else None
    reverse_bm = None if native is not None else ReverseBrownian(bm)
    params = list(params)
    state = (ys[-1], f, g, z, grad_ys[-1].contiguous().clone(), a_f.contiguous().clone(), a_g.contiguous().clone(),
             a_z.contiguous().clone())
    a_theta = [torch.zeros_like(p) for p in params]

    for (i, grid, tau64, fwd_times, cells, tau_dev) in plan:
        n = grid.n_steps
        for k in range(n):
            if native is not None:
                if cells is not None:
                    c = int(cells[n - 1 - k])
                    noise = NoiseSpec.generated(native, c, native.cell_width(c))
                else:
                    W, _ = native.increment(-tau64[k + 1], -tau64[k])
                    noise = NoiseSpec.external(W)
            else:
                noise = NoiseSpec.external(reverse_bm(tau_dev[k], tau_dev[k + 1]))
            state, vjp_theta = _reversible_heun_backward_step(sde, params, diag, state, noise, grid.dt[k], fwd_times[k],
                                                              fwd_times[k + 1])
            for acc, v in zip(a_theta, vjp_theta):
                acc.add_(v)
        # adjoint.py:114-116
        state = (ys[i - 1],) + state[1:4] + (state[4] + grad_ys[i - 1],) + state[5:]
    return [state[4], state[5], state[6], state[7]] + a_theta


def _run_reversible_heun_backward_adaptive(sde, bm, params, ts_host, dt, rtol, atol, dt_min, ys, grad_ys, f, g, z, a_f,
                                           a_g, a_z):
    """`adjoint_adaptive=True` with `adjoint_reversible_heun`: the reference pushes this solver's step through its
    generic step-doubling loop like any other (base_solver.py:117-142): per attempt one full and two half steps from
    the same state, the error norm over the flat (y, a_y, a_f, a_g, a_z, a_theta), the carried (f, g, z) following the
    two half steps; every output interval starts again from `dt` (adjoint.py:97-112 calls integrate() per interval).
    (The gradients are then no longer the exact ones of the forward solve -- the backward steps do not retrace it.)"""
    diag = sde.noise_type == NOISE_TYPES.diagonal
    native = bm if isinstance(bm, BrownianInterval) else None
    reverse_bm = None if native is not None else ReverseBrownian(bm)
    params = list(params)
    device = ys.device
    np_dtype = ts_host.dtype.type
    state = (ys[-1], f, g, z, grad_ys[-1].contiguous().clone(), a_f.contiguous().clone(), a_g.contiguous().clone(),
             a_z.contiguous().clone())
    a_theta = [torch.zeros_like(p) for p in params]

    def noise_of(ta, tb):                       # the increment of the forward interval [-tb, -ta]
        if native is not None:
            return NoiseSpec.external(native.increment(-float(tb), -float(ta))[0])
        return NoiseSpec.external(reverse_bm(torch.tensor(ta, device=device), torch.tensor(tb, device=device)))

    def flat(st, theta):
        return torch.cat([st[0].reshape(-1)] + [x.reshape(-1) for x in st[4:]] + [x.reshape(-1) for x in theta])

    for i in range(len(ts_host) - 1, 0, -1):
        curr_t, t_end = -ts_host[i], -ts_host[i - 1]
        step_size = dt if not torch.is_tensor(dt) else float(dt)
        prev_error_ratio = None
        while curr_t < t_end:
            nxt = curr_t + np_dtype(step_size)
            next_t = nxt if nxt <= t_end else t_end
            mid_t = np_dtype(0.5) * (curr_t + next_t)
            t_dev = torch.from_numpy(np.asarray([-curr_t, -mid_t, -next_t], dtype=ts_host.dtype)).to(device).unbind(0)
            full, th_full = _reversible_heun_backward_step(sde, params, diag, state, noise_of(curr_t, next_t),
                                                           np_dtype(next_t - curr_t), t_dev[0], t_dev[2])
            half, th_a = _reversible_heun_backward_step(sde, params, diag, state, noise_of(curr_t, mid_t),
                                                        np_dtype(mid_t - curr_t), t_dev[0], t_dev[1])
            two, th_b = _reversible_heun_backward_step(sde, params, diag, half, noise_of(mid_t, next_t),
                                                       np_dtype(next_t - mid_t), t_dev[1], t_dev[2])
            theta_full = [acc + v for acc, v in zip(a_theta, th_full)]
            theta_two = [acc + va + vb for acc, va, vb in zip(a_theta, th_a, th_b)]
            error_estimate = solvers._error_estimate(flat(full, theta_full), flat(two, theta_two), rtol, atol)
            step_size, prev_error_ratio = solvers._update_step_size(error_estimate, step_size, prev_error_ratio)
            if step_size < dt_min:
                warnings.warn("Hitting minimum allowed step size in adaptive time-stepping.")
                step_size = dt_min
                prev_error_ratio = None
            if error_estimate <= 1 or step_size <= dt_min:
                curr_t, state, a_theta = next_t, two, theta_two
        # adjoint.py:114-116
        state = (ys[i - 1],) + state[1:4] + (state[4] + grad_ys[i - 1],) + state[5:]
    return [state[4], state[5], state[6], state[7]] + a_theta


class _SdeintAdjointMethod(torch.autograd.Function):

    @staticmethod
    def forward(ctx, sde, ts, dt, bm, solver, method, adjoint_method, adjoint_adaptive, adjoint_rtol, adjoint_atol,
                dt_min, adjoint_options, len_extras, y0, *extras_and_adjoint_params):
        ctx.sde, ctx.dt, ctx.bm = sde, dt, bm
        ctx.adjoint_method, ctx.adjoint_options = adjoint_method, adjoint_options
        ctx.adjoint_adaptive = (adjoint_adaptive, adjoint_rtol, adjoint_atol, dt_min)
        ctx.len_extras = len_extras
        extra_solver_state = extras_and_adjoint_params[:len_extras]
        adjoint_params = extras_and_adjoint_params[len_extras:]
        # detached on purpose: the backward pass differentiates f, g at leaf copies of y (adjoint.py:48-51)
        y0 = y0.detach()
        extra_solver_state = tuple(x.detach() for x in extra_solver_state)
        ys, extra_solver_state = solver.integrate(y0, ts, extra_solver_state)
        # only the reversible pair reuses the solver's final (f, g, z) in the backward pass (adjoint.py:54-60)
        ctx.saved_extras_for_backward = (method == METHODS.reversible_heun and
                                         adjoint_method == METHODS.adjoint_reversible_heun)
        extras_for_backward = tuple(extra_solver_state) if ctx.saved_extras_for_backward else ()
        ctx.save_for_backward(ys, ts, *extras_for_backward, *adjoint_params)
        ctx.captured_backward = None   # set by `sdeint_adjoint` when the backward sweep is to replay a HIP graph
        ctx.watch_backward = None      # ... or ("auto") when an eager sweep is to report whether it synchronised
        return (ys, *extra_solver_state)

    @staticmethod
    def backward(ctx, grad_ys, *grad_extra_solver_state):
        from . import _native
        with _native.on_device_of(grad_ys):
            if torch.is_grad_enabled():      # create_graph=True: a differentiable statement of the same sweep
                return _SdeintAdjointMethod._backward_with_graph(ctx, grad_ys)
            return _SdeintAdjointMethod._backward(ctx, grad_ys, *grad_extra_solver_state)

    @staticmethod
    def _backward_with_graph(ctx, grad_ys):
        """Second derivatives (reference: the nested apply of adjoint.py:97-112); see adjoint_double.py."""
        from . import adjoint_double
        if ctx.saved_extras_for_backward or ctx.adjoint_adaptive[0]:
            raise NotImplementedError("torchsde_amd: double backward is available for the fixed-step adjoint methods "
                                      f"{adjoint_double.SUPPORTED} only.")
        ys, ts, *adjoint_params = ctx.saved_tensors
        kind = _backward_kind(ctx.sde, ctx.bm, ctx.adjoint_method, ctx.adjoint_options, adjoint_params)
        native = ctx.bm if isinstance(ctx.bm, BrownianInterval) else None
        plan = _plan_backward(timegrid.ts_to_host(ts), ctx.dt, native, ys.device)
        a_y, a_theta = adjoint_double.run(AdjointSDE(ctx.sde, adjoint_params), kind, ctx.bm, plan, ys, grad_ys)
        return (None,) * 13 + (a_y,) + (None,) * ctx.len_extras + tuple(a_theta)

    @staticmethod
    def _backward(ctx, grad_ys, *grad_extra_solver_state):
        ys, ts, *rest = ctx.saved_tensors
        reversible = ctx.saved_extras_for_backward
        forward_extras = rest[:ctx.len_extras] if reversible else []
        adjoint_params = rest[ctx.len_extras:] if reversible else rest
        grad_extras = [torch.zeros_like(x) if gr is None else gr
                       for gr, x in zip(grad_extra_solver_state, forward_extras)]
        inputs = [ys, grad_ys] + list(forward_extras) + grad_extras
        captured = ctx.captured_backward
        if ctx.adjoint_adaptive[0]:
            _, rtol, atol, dt_min = ctx.adjoint_adaptive
            kind = _backward_kind(ctx.sde, ctx.bm, ctx.adjoint_method, ctx.adjoint_options, adjoint_params)
            if kind == "reversible_heun":
                out = _run_reversible_heun_backward_adaptive(ctx.sde, ctx.bm, adjoint_params, timegrid.ts_to_host(ts),
                                                             ctx.dt, rtol, atol, dt_min, *inputs)
            else:
                a_y, a_theta = _run_backward_adaptive(AdjointSDE(ctx.sde, adjoint_params), kind, ctx.bm,
                                                      timegrid.ts_to_host(ts), ctx.dt, rtol, atol, dt_min, ys, grad_ys)
                out = [a_y] + list(a_theta)
        elif captured is not None and not getattr(captured, "broken", False):
            out = captured.replay(ctx.bm, inputs)
            captured.replays += 1
            from . import graph
            if captured.replays <= 2 or (getattr(ctx, "backward_graph_is_auto", True)
                                         and graph.due_for_a_check(captured.replays)):
                # A recorded sweep has passed its checks at recording time (graph.replays_are_stable); its first replays
                # in real use -- and then every 8th, 64th, 512th ... -- are still compared with the eager sweep: the fault
                # those checks exist for shows only after other work has run on the device, and state the cache key
                # cannot see is caught late rather than never. A graph that fails is never replayed again.
                kind = _backward_kind(ctx.sde, ctx.bm, ctx.adjoint_method, ctx.adjoint_options, adjoint_params)
                run = _backward_runner(ctx.sde, ctx.bm, ctx.dt, kind, adjoint_params, timegrid.ts_to_host(ts), ys.device)
                eager = run(*inputs)
                if not graph._same_tensors(out, eager, exact=False):
                    captured.broken = True
                    out = eager
        else:
            kind = _backward_kind(ctx.sde, ctx.bm, ctx.adjoint_method, ctx.adjoint_options, adjoint_params)
            run = _backward_runner(ctx.sde, ctx.bm, ctx.dt, kind, adjoint_params, timegrid.ts_to_host(ts), ys.device)
            if ctx.watch_backward is not None and captured is None:
                from . import graph
                verdict = {}
                try:
                    out, reason = graph.run_screened(lambda: run(*inputs), verdict)
                except Exception as e:
                    # the user's own error, or code that does not run under the screen: this structure stays eager, and
                    # the plain sweep reports the error if it is real
                    ctx.watch_backward(f"the screened sweep raised {type(e).__name__}")
                    out = run(*inputs)
                else:
                    ctx.watch_backward(reason, verdict["independent"])
            else:
                out = run(*inputs)
        if reversible:      # a_y, (a_f, a_g, a_z), a_theta...
            return (None,) * 13 + tuple(out)
        return (None,) * 13 + tuple([out[0]] + ([None] * ctx.len_extras) + list(out[1:]))


def _capture_backward(sde, bm, dt, adjoint_method, adjoint_options, adjoint_params, ts, ys, forward_extras, auto=False):
    """HIP graph of the backward sweep (cached per structure on the SDE object), or None -> eager backward.

    The VJPs of the sweep are taken w.r.t. the parameters. The real parameters already carry AccumulateGrad nodes
    created on the default stream by the forward call (the autograd graph of `ys` keeps them alive), and autograd
    would synchronise the capture stream with that stream -- which invalidates the capture. So, while the sweep is
    recorded, the module runs on leaf ALIASES of its parameters (same storage, fresh autograd identity): the
    recorded kernels read the real parameter memory, so optimiser steps are seen by every replay."""
    from torch.nn.utils.stateless import _reparametrize_module
    from . import graph
    if not isinstance(sde, nn.Module):
        return (None, None) if auto else None
    ts_host = timegrid.ts_to_host(ts)
    kind = _backward_kind(sde, bm, adjoint_method, adjoint_options, adjoint_params)
    signature = ("adjoint-backward", kind, sde.sde_type, sde.noise_type, tuple(ys.shape), ys.dtype, str(ys.device),
                 tuple(ts_host.tolist()), float(dt), tuple((p.data_ptr(), tuple(p.shape)) for p in adjoint_params))

    def capture():
        alias_of = {id(p): p.detach().requires_grad_(True) for p in adjoint_params}
        swapped = {name: alias_of[id(p)] for name, p in sde.named_parameters(remove_duplicate=False)
                   if id(p) in alias_of}
        if len({id(a) for a in swapped.values()}) != len(alias_of):
            if not auto:
                warnings.warn("adjoint_options['hip_graph'] needs every adjoint parameter to be a parameter of the SDE "
                              "module; running the backward pass eagerly.")
            return None
        run = _backward_runner(sde, bm, dt, kind, [alias_of[id(p)] for p in adjoint_params], ts_host, ys.device)
        # The capture only records; `backward` copies the real cotangents in before each replay. All-ones cotangents:
        # the recorded graph is replayed a few times right away and must keep giving the same (and for "auto": the
        # eager sweep's) gradients, which says nothing if they are identically zero.
        fill = torch.ones_like
        inputs = [ys, fill(ys)] + list(forward_extras) + [fill(x) for x in forward_extras]
        with torch.no_grad(), _reparametrize_module(sde, swapped):
            return graph._CapturedBackward(run, bm, inputs, keepalive=(run.plan,), verify=auto)

    def tuned_capture():       # drift and diffusion in sequence or as parallel graph branches: whichever replays faster
        from .sde import ForwardSDE
        return graph.faster_of_sequential_and_parallel(sde if isinstance(sde, ForwardSDE) else None, capture, ys.device)

    return graph.cached_backward(sde, bm, signature, capture if auto else tuned_capture, auto=auto,
                                 tuned_capture=tuned_capture)


def _backward_kind(sde, bm, adjoint_method, adjoint_options, adjoint_params):
    method_cls = _check_adjoint_method(AdjointSDE(sde, adjoint_params), adjoint_method, adjoint_options, bm)
    if method_cls is _ReversibleHeunBackward:
        return "reversible_heun"
    for cls, kind in ((solvers.Euler, "euler"), (solvers.Midpoint, "midpoint"), (solvers._Milstein, "milstein"),
                      (solvers.Heun, "heun"), (solvers.EulerHeun, "euler_heun")):
        if issubclass(method_cls, cls):
            return kind
    raise NotImplementedError(f"torchsde_amd: no backward sweep for adjoint_method={repr(adjoint_method)}.")


def _backward_runner(sde, bm, dt, kind, adjoint_params, ts_host, device):
    """`run(ys, grad_ys, *extras) -> [a_y0, ..., *a_theta]`: the launch-only backward sweep over a plan prepared here.
    For the reversible-Heun pair the extras are (f, g, z, a_f, a_g, a_z) and the result carries (a_f, a_g, a_z)
    after a_y."""
    native = bm if isinstance(bm, BrownianInterval) else None
    if kind == "reversible_heun":
        plan = _plan_reversible_heun_backward(ts_host, dt, native, device)

        def run(ys_, grad_ys_, f, g, z, a_f, a_g, a_z):
            return _run_reversible_heun_backward(sde, bm, adjoint_params, plan, ys_, grad_ys_, f, g, z, a_f, a_g, a_z)
        run.plan = plan
        return run
    adjoint_sde = AdjointSDE(sde, adjoint_params)
    plan = _plan_backward(ts_host, dt, native, device)

    def run(ys_, grad_ys_):
        a_y, a_theta = _run_backward(adjoint_sde, kind, bm, plan, ys_, grad_ys_)
        return [a_y] + list(a_theta)
    run.plan = plan
    return run


def _plan_backward(ts_host, dt, native, device):
    """Host-side preparation of the backward sweep (everything that copies from the host): per output interval the
    reversed time grid, its stage times on the device, and the Brownian cell of every step when the reversed steps
    line up with the generator's cells."""
    intervals = []
    misaligned = False
    for i in range(len(ts_host) - 1, 0, -1):
        grid = timegrid.build(np.array([-ts_host[i], -ts_host[i - 1]], dtype=ts_host.dtype), dt)
        n = grid.n_steps
        np_dtype = grid.t.dtype.type
        tau64 = grid.t_f64()
        # forward-time stage tensors: -(tau0), and for midpoint -(tau0 + dt/2)   (adjoint_sde.py passes -t)
        # ... and -(tau1) for the second stage of Heun / Euler-Heun
        stage = np.empty((max(n, 1), 3), dtype=grid.t.dtype)
        stage[:n, 0] = -grid.t[:-1]
        stage[:n, 1] = -(grid.t[:-1] + np_dtype(0.5) * grid.dt)
        stage[:n, 2] = -grid.t[1:]
        stage_dev = torch.from_numpy(stage).to(device)
        stage_rows = [r.unbind(0) for r in stage_dev.unbind(0)]
        tau_dev = None
        cells = None
        if native is not None:
            cells = native.match_grid(-tau64[::-1]) if native.frozen else None
            misaligned = misaligned or cells is None
        else:
            tau_dev = torch.from_numpy(grid.t.copy()).to(device).unbind(0)
        intervals.append((i, grid, tau64, stage_rows, cells, tau_dev))
    if misaligned:
        native.locate(float(ts_host[0]), float(ts_host[-1]))   # freezes a generator that never saw a grid
        native._device_edges()    # upload the cell edges now: the sweep itself must not copy from the host
    return intervals


def _aug_step(adjoint_sde, kind, ito, src, dst, mid, t_fwd, t_fwd_half, t_fwd_end, step_dt, v):
    """One backward step of the augmented state `src` -> `dst` over a step of size `step_dt` that starts at forward
    time `t_fwd` (0-d device tensor; `t_fwd_half`: the midpoint stage's, `t_fwd_end`: the step's end, for the second
    stage of Heun / Euler-Heun), with the reversed increment `v`."""
    y, a = src.t[0], src.t[1]
    none_tail = [None] * (len(src.t) - 1)
    if kind == "euler":
        # y' = y - f~ dt - g.v ;  (a, theta)' += dt*vjp(f~) + vjp(g.v)  (one reverse sweep, one launch)
        ft, gp, tot = adjoint_sde.fused_terms(t_fwd, y, a, v, float(step_dt), 1.0)
        _update(dst, src, [ft] + none_tail, [gp] + none_tail, [None] + tot, step_dt, 1.0)
    elif kind == "midpoint":
        half_dt = type(step_dt)(0.5) * step_dt
        ft, gp, tot = adjoint_sde.fused_terms(t_fwd, y, a, v, float(half_dt), 0.5)
        _update(mid, src, [ft] + none_tail, [gp] + none_tail, [None] + tot, half_dt, 0.5)
        ft, gp, tot = adjoint_sde.fused_terms(t_fwd_half, mid.t[0], mid.t[1], v, float(step_dt), 1.0)
        _update(dst, src, [ft] + none_tail, [gp] + none_tail, [None] + tot, step_dt, 1.0)
    elif kind == "heun":
        # heun.py:35-48 on the augmented state: prime = s + dt F + G ; s1 = s + (dt (F + F') + G + G') / 2
        half_dt = type(step_dt)(0.5) * step_dt
        ft, gp, tot = adjoint_sde.fused_terms(t_fwd, y, a, v, float(step_dt), 1.0)
        _update(mid, src, [ft] + none_tail, [gp] + none_tail, [None] + tot, step_dt, 1.0)
        ft2, gp2, tot2 = adjoint_sde.fused_terms(t_fwd_end, mid.t[0], mid.t[1], v, float(step_dt), 1.0)
        both = [K.lincomb2(p, q, 0.5, 0.5) for p, q in zip(tot, tot2)]
        _update(dst, src, [K.lincomb2(ft, ft2, 1.0, 1.0)] + none_tail, [K.lincomb2(gp, gp2, 1.0, 1.0)] + none_tail,
                [None] + both, half_dt, 0.5)
    elif kind == "euler_heun":
        # euler_heun.py:29-42: prime = s + G ; G' = g_prod(t1, prime) ; s1 = s + dt F + (G + G') / 2
        F, G = adjoint_sde.f_and_g_prod(t_fwd, y, a, v)
        _update(mid, src, None, G, None, 0.0, 1.0)
        G2 = adjoint_sde.g_prod(t_fwd_end, mid.t[0], mid.t[1], v)
        _update(dst, src, F, [K.lincomb2(p, q, 1.0, 1.0) for p, q in zip(G, G2)], None, step_dt, 0.5)
    else:  # milstein (diagonal noise): v_term = I^2 - dt (Ito) or I^2, halved (milstein.py:56,70)
        v2, _ = K.milstein_v(NoiseSpec.external(v), step_dt, ito, 0.5, like=y)
        F = adjoint_sde.f(t_fwd, y, a)
        G, D = adjoint_sde.g_prod_and_gdg_prod(t_fwd, y, a, v, v2)
        _update(dst, src, F, G, D, step_dt, 1.0)


def _run_backward(adjoint_sde, kind, bm, plan, ys, grad_ys):
    """Launch-only part of the backward sweep (capturable in a HIP graph): returns (a_y0, [a_theta...])."""
    sde = adjoint_sde.forward_sde
    adjoint_params = adjoint_sde.params
    ito = sde.sde_type == SDE_TYPES.ito
    native = bm if isinstance(bm, BrownianInterval) else None
    reverse_bm = None if native is not None else ReverseBrownian(bm)
    state = _AugState([ys[-1].clone(), grad_ys[-1].contiguous().clone()] +
                      [torch.zeros_like(p) for p in adjoint_params])
    other = _AugState.like(state)
    mid = _AugState.like(state) if kind in ("midpoint", "heun", "euler_heun") else None

    for (i, grid, tau64, stage_rows, cells, tau_dev) in plan:
        n = grid.n_steps
        for k in range(n):
            if native is not None:
                if cells is not None:
                    c = int(cells[n - 1 - k])
                    noise = NoiseSpec.generated(native, c, native.cell_width(c))
                    v, _ = noise.materialise()
                else:
                    v, _ = native.increment(-tau64[k + 1], -tau64[k])
            else:
                v = reverse_bm(tau_dev[k], tau_dev[k + 1])
            _aug_step(adjoint_sde, kind, ito, state, other, mid, *stage_rows[k], grid.dt[k], v)
            state, other = other, state
        # adjoint.py:114-116
        state.t[0].copy_(ys[i - 1])
        state.t[1].add_(grad_ys[i - 1])

    _SEG_CACHE.clear()   # drop the references to this pass's buffers
    return state.t[1], state.t[2:]


def _run_backward_adaptive(adjoint_sde, kind, bm, ts_host, dt, rtol, atol, dt_min, ys, grad_ys):
    """The backward sweep with step doubling on the augmented state (`adjoint_adaptive=True`; reference: the generic
    adaptive loop of base_solver.py:117-142 applied to the flat augmented state, adjoint.py:97-112). The error norm
    is taken over all segments together, like the reference's single flat tensor. One host sync per attempt."""
    sde = adjoint_sde.forward_sde
    ito = sde.sde_type == SDE_TYPES.ito
    native = bm if isinstance(bm, BrownianInterval) else None
    reverse_bm = None if native is not None else ReverseBrownian(bm)
    device = ys.device
    np_dtype = ts_host.dtype.type
    state = _AugState([ys[-1].clone(), grad_ys[-1].contiguous().clone()] +
                      [torch.zeros_like(p) for p in adjoint_sde.params])
    full, half, nxt_state = _AugState.like(state), _AugState.like(state), _AugState.like(state)
    mid = _AugState.like(state) if kind in ("midpoint", "heun", "euler_heun") else None

    def increment(ta, tb):                       # reversed increment of the backward-time interval [ta, tb]
        if native is not None:
            return native.increment(-float(tb), -float(ta))[0]
        return reverse_bm(torch.tensor(ta, device=device), torch.tensor(tb, device=device))

    def flat(x):
        return torch.cat([t.reshape(-1) for t in x.t])

    for i in range(len(ts_host) - 1, 0, -1):
        curr_t, t_end = -ts_host[i], -ts_host[i - 1]
        # every output interval is its own integrate() call in the reference (adjoint.py:97-112): back to `dt`
        step_size = dt if not torch.is_tensor(dt) else float(dt)
        prev_error_ratio = None
        while curr_t < t_end:
            nxt = curr_t + np_dtype(step_size)
            next_t = nxt if nxt <= t_end else t_end
            mid_t = np_dtype(0.5) * (curr_t + next_t)
            v_a, v_b = increment(curr_t, mid_t), increment(mid_t, next_t)
            v_full = v_a + v_b if native is not None else increment(curr_t, next_t)
            h_full, h_a, h_b = np_dtype(next_t - curr_t), np_dtype(mid_t - curr_t), np_dtype(next_t - mid_t)
            stage = np.asarray([-curr_t, -(curr_t + np_dtype(0.5) * h_full), -(curr_t + np_dtype(0.5) * h_a),
                                -mid_t, -(mid_t + np_dtype(0.5) * h_b), -next_t], dtype=ts_host.dtype)
            t_dev = torch.from_numpy(stage).to(device).unbind(0)
            _aug_step(adjoint_sde, kind, ito, state, full, mid, t_dev[0], t_dev[1], t_dev[5], h_full, v_full)
            _aug_step(adjoint_sde, kind, ito, state, half, mid, t_dev[0], t_dev[2], t_dev[3], h_a, v_a)
            _aug_step(adjoint_sde, kind, ito, half, nxt_state, mid, t_dev[3], t_dev[4], t_dev[5], h_b, v_b)
            error_estimate = solvers._error_estimate(flat(full), flat(nxt_state), rtol, atol)
            step_size, prev_error_ratio = solvers._update_step_size(error_estimate, step_size, prev_error_ratio)
            if step_size < dt_min:
                warnings.warn("Hitting minimum allowed step size in adaptive time-stepping.")
                step_size = dt_min
                prev_error_ratio = None
            if error_estimate <= 1 or step_size <= dt_min:
                curr_t = next_t
                state, nxt_state = nxt_state, state
        # adjoint.py:114-116
        state.t[0].copy_(ys[i - 1])
        state.t[1].add_(grad_ys[i - 1])

    _SEG_CACHE.clear()
    return state.t[1], state.t[2:]


def sdeint_adjoint(sde, y0, ts, bm=None, method=None, adjoint_method=None, dt=1e-3, adaptive=False,
                   adjoint_adaptive=False, rtol=1e-5, adjoint_rtol=1e-5, atol=1e-4, adjoint_atol=1e-4, dt_min=1e-5,
                   options=None, adjoint_options=None, adjoint_params=None, names=None, logqp=False, extra=False,
                   extra_solver_state=None, **unused_kwargs):
    """Numerically integrate an SDE with stochastic-adjoint gradients (see ``torchsde.sdeint_adjoint``)."""
    contract.handle_unused_kwargs(unused_kwargs, msg="`sdeint_adjoint`")
    del unused_kwargs
    from . import _native
    with _native.on_device_of(y0 if torch.is_tensor(y0) else "cpu"):
        return _sdeint_adjoint(sde, y0, ts, bm, method, adjoint_method, dt, adaptive, adjoint_adaptive, rtol,
                               adjoint_rtol, atol, adjoint_atol, dt_min, options, adjoint_options, adjoint_params, names,
                               logqp, extra, extra_solver_state)


def _sdeint_adjoint(sde, y0, ts, bm, method, adjoint_method, dt, adaptive, adjoint_adaptive, rtol, adjoint_rtol, atol,
                    adjoint_atol, dt_min, options, adjoint_options, adjoint_params, names, logqp, extra,
                    extra_solver_state):
    if adjoint_params is None and not isinstance(sde, nn.Module):
        raise ValueError("`sde` must be an instance of nn.Module to specify the adjoint parameters; alternatively they "
                         "can be specified explicitly via the `adjoint_params` argument. If there are no parameters "
                         "then it is allowable to set `adjoint_params=()`.")

    sde, y0, ts, bm, method, options = contract.check_contract(sde, y0, ts, bm, method, adaptive, options, names,
                                                               logqp)
    contract.assert_no_grad(["ts", "dt", "rtol", "adjoint_rtol", "atol", "adjoint_atol", "dt_min"],
                            [ts, dt, rtol, adjoint_rtol, atol, adjoint_atol, dt_min])
    adjoint_params = tuple(sde.parameters()) if adjoint_params is None else tuple(adjoint_params)
    adjoint_params = tuple(p for p in adjoint_params if p.requires_grad)
    adjoint_method = _select_default_adjoint_method(sde, method, adjoint_method)
    adjoint_options = {} if adjoint_options is None else adjoint_options.copy()
    if method == METHODS.reversible_heun:   # adjoint.py:243-257
        if adjoint_method != METHODS.adjoint_reversible_heun:
            warnings.warn(f"method={repr(method)}, but adjoint_method!={repr(METHODS.adjoint_reversible_heun)}.")
        num_steps = (ts - ts[0]) / dt
        if not torch.allclose(num_steps, num_steps.round()):
            warnings.warn(f"The spacing between time points `ts` is not an integer multiple of the time step `dt`. "
                          f"This means that the backward pass (which is forced to step to each of `ts` to get "
                          f"dL/dy(t) for t in ts) will not perfectly mimick the forward pass (which does not step "
                          f"to each `ts`, and instead interpolates to them). This means that "
                          f"method={repr(method)} may not be perfectly accurate.")

    solver_cls = solvers.select(method=method, sde_type=sde.sde_type)
    solver = solver_cls(sde=sde, bm=bm, dt=dt, adaptive=adaptive, rtol=rtol, atol=atol, dt_min=dt_min,
                        options=options)
    # An unusable adjoint method fails in the reference when backward() constructs the backward solver
    # (adjoint.py:64-96), so forward-only and no-grad calls go through there; here it is reported at call time, but
    # only when a backward pass can follow at all.
    if torch.is_grad_enabled() and (y0.requires_grad or adjoint_params):
        _check_adjoint_method(AdjointSDE(sde, adjoint_params), adjoint_method, adjoint_options, bm)
    # the perceptron-drift module with adjoint_method="euler": forward and backward solves on the matrix cores
    if torch.is_grad_enabled() and (y0.requires_grad or adjoint_params):
        from . import mlp_adjoint
        ys = mlp_adjoint.route(sde, y0, ts, bm, method, adjoint_method, dt, adaptive, adjoint_adaptive, options,
                               adjoint_options, adjoint_params, extra_solver_state, solver=solver)
        if ys is not None:
            return contract.parse_return(y0, ys, (), extra, logqp)
    # the reversible pair on networks of (t, y): forward one launch, backward the exact-gradient sweep on the matrix cores
    # (neural_rheun.py). The first solve of a form runs both routes and compares values AND gradients (below).
    kernels_route = None
    if (method == METHODS.reversible_heun and adjoint_method == METHODS.adjoint_reversible_heun and torch.is_grad_enabled()
            and (y0.requires_grad or adjoint_params) and not extra and not logqp and extra_solver_state is None
            and not adaptive and not adjoint_adaptive and adjoint_options.get("trajectory_kernel", True)
            and y0.numel() > 0):
        from . import neural_rheun_route
        kernels_route = neural_rheun_route.plan_adjoint(solver, sde, y0, ts, bm, dt, adjoint_params)
        if kernels_route is not None and kernels_route.trusted:
            return contract.parse_return(y0, kernels_route.solve(y0), (), extra, logqp)
    if hasattr(solver, "wants_extra"):
        solver.wants_extra = True        # (the autograd function below carries the solver's final (f, g, z) to its backward)
    if extra_solver_state is None:
        extra_solver_state = solver.init_extra_solver_state(ts[0], y0)
    if y0.numel() == 0:      # an empty batch: nothing to launch, and no trajectory for a gradient to come from
        ys = y0.unsqueeze(0).repeat(len(ts), *([1] * y0.dim()))
        return contract.parse_return(y0, ys, tuple(extra_solver_state), extra, logqp)

    ys, *extra_solver_state = _SdeintAdjointMethod.apply(
        sde, ts, dt, bm, solver, method, adjoint_method, adjoint_adaptive, adjoint_rtol, adjoint_atol, dt_min,
        adjoint_options, len(extra_solver_state), y0, *extra_solver_state, *adjoint_params)
    if kernels_route is not None:
        # the verifying solve: the stepwise pair's result is what is returned; the kernels must reproduce its values and, for
        # one random cotangent, its gradients with respect to y0 and every adjoint parameter
        kernels_route.record(kernels_route.solve(y0), ys, y0, extra_inputs=adjoint_params)
    from . import graph
    graph_mode = graph.mode_of(adjoint_options)
    # (the parameter gradients of a sweep are sums over the batch; above ~1000 rows torch computes them with multi-block
    #  reductions, whose semaphore memsets are the graph nodes that go wrong on this runtime: graph._capturing rewrites
    #  them as kernels, and the recorded sweep still has to pass graph.replays_are_stable and its probation)
    if graph_mode == "auto" and not (isinstance(sde, nn.Module) and graph._auto_eligible(bm, y0, len(ts))):
        graph_mode = False
    if graph_mode and not adjoint_adaptive and ys.grad_fn is not None and isinstance(bm, BrownianInterval):
        # The backward sweep replays ONE HIP graph, captured HERE (on the caller's thread, outside the autograd
        # Function); `backward` only copies ys / grad_ys into the graph's static inputs and replays. `ys.grad_fn` is
        # the Function's ctx. "auto" (the default): the first sweep of a structure runs eagerly and is watched for
        # host synchronisation, the second is captured and checked against an eager sweep, later ones replay.
        reversible = method == METHODS.reversible_heun and adjoint_method == METHODS.adjoint_reversible_heun
        captured = _capture_backward(
            sde, bm, dt, adjoint_method, adjoint_options, adjoint_params, ts, ys.detach(),
            [x.detach() for x in extra_solver_state] if reversible else [], auto=graph_mode == "auto")
        ys.grad_fn.backward_graph_is_auto = graph_mode == "auto"
        if graph_mode == "auto":
            ys.grad_fn.captured_backward, ys.grad_fn.watch_backward = captured
        else:
            ys.grad_fn.captured_backward = captured
    return contract.parse_return(y0, ys, tuple(extra_solver_state), extra, logqp)


def _select_default_adjoint_method(sde, method, adjoint_method):
    """adjoint.py:281-296."""
    if adjoint_method is not None:
        return adjoint_method
    if method == METHODS.reversible_heun:
        return METHODS.adjoint_reversible_heun
    if sde.sde_type == SDE_TYPES.stratonovich:
        return METHODS.midpoint
    return METHODS.milstein if sde.noise_type == NOISE_TYPES.diagonal else METHODS.euler

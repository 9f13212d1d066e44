"""Fixed-step SDE solvers whose state updates are HIP kernels.

Solver protocol as in the reference (torchsde/_core/base_solver.py:29-149): class attributes
``strong_order, weak_order, sde_type, noise_types, levy_area_approximations``; constructor
``(sde, bm, dt, adaptive, rtol, atol, dt_min, options)`` with the same compatibility errors (:49-58);
``init_extra_solver_state``; ``step(t0, t1, y0, extra0) -> (y1, extra1)``; ``integrate(y0, ts, extra0)``.

``integrate`` is a different program from the reference's loop (:114-149). Closed-form SDEs (closed_form.py) run all
their steps in one launch of a trajectory kernel (`_integrate_trajectory`). For everything else the time grid,
per-step ``dt``, stage times and interpolation weights are computed once on the host (timegrid.py); each step is the user's
``f``/``g`` torch ops plus ONE fused kernel per solver stage that reads ``y, f, g``, generates the Brownian
increment of the step's grid cell in registers and writes the new state straight into its destination
(the next ``ys[i]`` slot when the step lands on an output time). No host sync, no per-step allocation of
increments, no ``torch.stack`` copy.
"""
import os

import numpy as np
import torch

from . import _native
from . import closed_form
from . import kernels as K
from . import timegrid
from .brownian import BrownianInterval
from .kernels import NoiseSpec
from .settings import LEVY_AREA_APPROXIMATIONS, METHOD_OPTIONS, METHODS, NOISE_TYPES, SDE_TYPES


# TSDE_VERIFY_EVERY=N: every N-th solve of a form that has earned trust on a kernel route runs both routes again and compares
# values (and gradients where autograd records); a mismatch raises. 0 = only the first solve verifies.
try:
    VERIFY_EVERY = max(0, int(os.environ.get("TSDE_VERIFY_EVERY", "0").strip() or 0))
except ValueError:
    VERIFY_EVERY = 0


class _Step:
    """Everything one solver step needs besides the state.

    `dt` and the scalars derived from it are numpy scalars of ts.dtype (rounded like the reference's 0-d tensor
    arithmetic) -- or, when an adaptive solve is controlled on the device (adaptive.py), TSDE_DEV_SCALAR floats: the
    kernels then read the values from the controller's table, and `times` are 0-d views of the same table."""
    __slots__ = ("times", "dt", "noise", "h64", "t0_64", "_half_dt", "_sqrt_dt", "_rdt")

    def __init__(self, times, dt, noise, h64, t0_64=None, half_dt=None, sqrt_dt=None, rdt=None):
        self.t0_64 = t0_64   # float(t0) on the host
        self.times = times   # tuple of 0-d device tensors: stage times (times[0] = t0)
        self.dt = dt         # t1 - t0
        self.noise = noise   # NoiseSpec
        self.h64 = h64       # float(t1) - float(t0) in double (what the Brownian motion sees)
        self._half_dt, self._sqrt_dt, self._rdt = half_dt, sqrt_dt, rdt

    @property
    def half_dt(self):       # `0.5 * dt` (midpoint.py:35, reversible_heun.py:71)
        if self._half_dt is None:
            self._half_dt = type(self.dt)(0.5) * self.dt
        return self._half_dt

    @property
    def sqrt_dt(self):       # `dt.sqrt()` (milstein.py:60, srk.py:62)
        if self._sqrt_dt is None:
            self._sqrt_dt = np.sqrt(self.dt)
        return self._sqrt_dt

    @property
    def rdt(self):           # `1 / dt` (srk.py:59)
        if self._rdt is None:
            self._rdt = type(self.dt)(1) / self.dt
        return self._rdt


def _error_estimate(y_full, y_half, rtol, atol, eps=1e-7):
    """Scaled RMS difference between one full step and two half steps (adaptive_stepping.py:42-76)."""
    value = K.error_norm(y_full, y_half, rtol, atol, eps).item()   # the one host sync of an adaptive step
    if value != value:
        raise AssertionError("Found nans in the error estimate. Try increasing the tolerance or regularizing the "
                             "dynamics.")
    return value


def _update_step_size(error_estimate, prev_step_size, prev_error_ratio, safety=0.9, facmin=0.2, facmax=1.4):
    """PI controller proposing the next step size (adaptive_stepping.py:21-39)."""
    if error_estimate > 1:
        pfactor, ifactor = 0, 1 / 1.5
    else:
        pfactor, ifactor = 0.13, 1 / 4.5
    error_ratio = safety / error_estimate
    if prev_error_ratio is None:
        prev_error_ratio = error_ratio
    factor = error_ratio ** ifactor * (error_ratio / prev_error_ratio) ** pfactor
    if error_estimate <= 1:
        prev_error_ratio = error_ratio
        facmin = 1.0
    factor = min(facmax, max(facmin, factor))
    return prev_step_size * factor, prev_error_ratio


class _Found(Exception):
    """Control flow of `_integrate_recognised`: the additive route found its form, skip the other interpreters."""


class BaseSDESolver:
    strong_order = None
    weak_order = None
    sde_type = None
    noise_types = ()
    levy_area_approximations = ()
    needs_U = False
    stateful = False   # True for solvers that carry extra state between steps (reversible Heun: f, g, z)
    # host-side stage-time offsets as multiples of dt (times[j] = t0 + stage_fracs[j]*dt), t0 first
    stage_fracs = (0,)

    def __init__(self, sde, bm, dt, adaptive, rtol, atol, dt_min, options, **kwargs):
        super().__init__(**kwargs)
        for attr in ("strong_order", "weak_order", "sde_type"):
            if getattr(self, attr) is None:
                raise NotImplementedError(f"{type(self).__name__} must define `{attr}`.")
        if sde.sde_type != self.sde_type:
            raise ValueError(f"SDE is of type {sde.sde_type} but solver is for type {self.sde_type}")
        if sde.noise_type not in self.noise_types:
            raise ValueError(f"SDE has noise type {sde.noise_type} but solver only supports noise types "
                             f"{self.noise_types}")
        if bm.levy_area_approximation not in self.levy_area_approximations:
            raise ValueError(f"SDE solver requires one of {self.levy_area_approximations} set as the "
                             f"`levy_area_approximation` on the Brownian motion.")
        if sde.noise_type == NOISE_TYPES.scalar and torch.Size(bm.shape[1:]).numel() != 1:
            raise ValueError("The Brownian motion for scalar SDEs must of dimension 1.")
        self.sde = sde
        # BrownianPath / BrownianTree (derived.py:52-191) answer interval queries with their BrownianInterval's increments
        # (w0 only shifts point values): the solver talks to that interval, so these objects reach every route it does
        from .brownian import _IntervalWrapper
        stock = isinstance(bm, _IntervalWrapper) and type(bm).__call__ is _IntervalWrapper.__call__
        self.bm = bm._interval if stock else bm            # (a subclass with its own __call__ is a foreign Brownian motion)
        self.dt = dt
        self.adaptive = adaptive
        self.rtol = rtol
        self.atol = atol
        self.dt_min = dt_min
        self.options = options

    def __repr__(self):
        return f"{self.__class__.__name__} of strong order: {self.strong_order}, and weak order: {self.weak_order}"

    def init_extra_solver_state(self, t0, y0):
        return ()

    # ---- what subclasses implement ---------------------------------------------------------------
    def _advance(self, y0, st, out):
        """One step from y0 using the prepared `_Step`; writes into `out` if given. Returns y1."""
        raise NotImplementedError

    # ---- noise plumbing ---------------------------------------------------------------------------
    def _native_bm(self):
        bm = self.bm
        return bm if isinstance(bm, BrownianInterval) else None

    def _noise_for(self, ta, tb, t0_tensor, t1_tensor, cell=None):
        """NoiseSpec of [ta, tb]: generated cell, native query, or a call into a foreign Brownian object."""
        bm = self._native_bm()
        if bm is not None:
            if cell is not None:
                return NoiseSpec.generated(bm, cell, bm.cell_width(cell))
            W, U = bm.increment(ta, tb, want_U=self.needs_U)
            return NoiseSpec.external(self._as_state_dtype(W), self._as_state_dtype(U))
        if self.needs_U:
            W, U = self.bm(t0_tensor, t1_tensor, return_U=True)
            return NoiseSpec.external(self._as_state_dtype(W), self._as_state_dtype(U))
        return NoiseSpec.external(self._as_state_dtype(self.bm(t0_tensor, t1_tensor)))

    def _as_state_dtype(self, x):
        """Materialised increments are read by the kernels in the state's dtype."""
        dtype = getattr(self, "_state_dtype", None)
        if x is None or dtype is None or x.dtype == dtype:
            return x
        return x.to(dtype)

    # ---- public single-step API (the reference's solver seam) --------------------------------------
    def _stage_times_host(self, t0n, t1n):
        """Stage times of the step [t0n, t1n] (numpy scalars in ts.dtype): t0 + frac*dt per stage, then t1."""
        dt = type(t0n)(t1n - t0n)
        return [t0n if frac == 0 else t0n + type(t0n)(frac) * dt for frac in self.stage_fracs] + [t1n]

    def step(self, t0, t1, y0, extra0, noise=None, times=None):
        """One step (the reference's solver seam). `noise`: a prepared NoiseSpec of [t0, t1] (step doubling hands in
        the whole step's increment merged from its halves); by default the Brownian motion is queried here.
        `times`: the step's stage times already on the device (one upload per attempt of step doubling)."""
        self._extra = tuple(extra0) if extra0 is not None else ()
        self._state_dtype = y0.dtype
        np_dtype = timegrid._NP.get(y0.dtype if not torch.is_tensor(t0) else t0.dtype, np.float64)
        ta, tb = float(t0), float(t1)
        t0n, t1n = np_dtype(ta), np_dtype(tb)
        dt = np_dtype(t1n - t0n)
        dev = y0.device
        if times is None:
            t0_t = t0 if torch.is_tensor(t0) else torch.tensor(ta, dtype=y0.dtype, device=dev)
            t1_t = t1 if torch.is_tensor(t1) else torch.tensor(tb, dtype=y0.dtype, device=dev)
            times = tuple(t0_t if frac == 0 else
                          torch.tensor(t0n + np_dtype(frac) * dt, dtype=t0_t.dtype, device=dev)
                          for frac in self.stage_fracs) + (t1_t,)
        if noise is None:
            cell = None
            bm = self._native_bm()
            if bm is not None and bm.frozen:
                cells = bm.match_grid(np.array([ta, tb]))
                cell = None if cells is None else int(cells[0])
            noise = self._noise_for(ta, tb, times[0], times[-1], cell)
        st = _Step(times, dt, noise, tb - ta, ta)
        y1 = self._advance(y0, st, None)
        return y1, self._extra

    # ---- the fixed-step driver -------------------------------------------------------------------------
    def integrate(self, y0, ts, extra0):
        if self.adaptive:
            return self._integrate_adaptive(y0, ts, extra0)
        self._extra = tuple(extra0) if extra0 is not None else ()
        self._state_dtype = y0.dtype
        coefficients = self._closed_form_coefficients(y0)
        if coefficients is not None:
            ys = self._integrate_trajectory(coefficients, y0, ts)
            if ys is not None:
                return ys, self._extra
        else:
            self._counter_start = None
            ys = self._integrate_recognised(y0, ts)
            if ys is not None:
                return ys, self._extra
            if self._counter_start is not None:       # (the probe calls of the interpretation are not steps of the solve)
                owner, start = self._counter_start
                for name, value in start.items():
                    setattr(owner, name, value)
        from . import graph
        mode = graph.mode_of(self.options)
        if mode is True:
            if not self._tracks_grad(y0):
                return graph.replay_or_capture(self, y0, ts, self._extra)
            graphed = graph.replay_or_capture_training(self, y0, ts, self._extra, self._params())
            if graphed is not None:
                return graphed
        elif mode == "auto" and not torch.is_grad_enabled():
            # the drop-in default: replay a HIP graph of this solve from its third occurrence on, when that is legal
            # and provably the same computation (graph.py); autograd THROUGH the solver stays eager unless asked for
            # (a replayed backward pass constrains the order of the caller's forward and backward calls)
            graphed = graph.auto_solve(self, y0, ts, self._extra)
            if graphed is not None:
                return graphed
        ys = self._run(self._plan(y0, ts), y0)
        return ys, self._extra

    def _integrate_adaptive(self, y0, ts, extra0):
        """Step-doubling adaptive stepping (reference: base_solver.py:114-149 adaptive branch +
        adaptive_stepping.py:21-76): one full step vs two half steps on the SAME Brownian path (the virtual bridge
        tree serves the half-interval queries), PI step-size controller, accept when the scaled RMS error <= 1.
        With this package's BrownianInterval (and autograd off) the accept / reject decision and the step-size
        controller run on the device between the attempts and the host synchronises once per output time
        (adaptive.py, csrc/adaptive.hip). The loop below is the host-driven form of the same algorithm, for foreign
        Brownian motions (their `bm(ta, tb)` needs the times on the host), stateful solvers and solves that track
        gradients: it synchronises once per attempted step. Either way the state updates are the HIP kernels of the
        fixed-step path."""
        import warnings
        from . import adaptive
        if adaptive.usable(self, y0, ts):
            # the same loop with accept / reject decided ON THE DEVICE: no sync per attempt (adaptive.py)
            return adaptive.integrate(self, y0, ts, extra0, _Step)
        if self.options.get("adaptive_replay", False) and adaptive.controllable(self, y0, ts) and torch.is_grad_enabled():
            # gradients flow, opt-in: the device-controlled loop finds the accepted steps, autograd records only those -- a
            # third of the autograd graph and one synchronisation, but the steps are computed twice: measured SLOWER than the
            # loop below unless the first pass replays a recorded attempt on a small state (profiles/r5_adaptive_one_sync.txt)
            done = adaptive.integrate_with_grad(self, y0, ts, extra0, _Step)
            if done is not None:
                return done
        np_dtype = timegrid._NP[ts.dtype]
        ts_host = timegrid.ts_to_host(ts)
        t_end = ts_host[-1]
        step_size = self.dt if not torch.is_tensor(self.dt) else float(self.dt)
        prev_t = curr_t = ts_host[0]
        prev_y = curr_y = y0
        curr_extra = tuple(extra0) if extra0 is not None else ()
        ys = [y0]
        prev_error_ratio = None
        for out_t in ts_host[1:]:
            while curr_t < out_t:
                nxt = curr_t + np_dtype(step_size)
                next_t = nxt if nxt <= t_end else t_end
                midpoint_t = np_dtype(0.5) * (curr_t + next_t)
                n_full, n_a, n_b = self._step_doubling_noise(float(curr_t), float(midpoint_t), float(next_t))
                # the stage times of the three steps of this attempt, in ONE host->device copy
                host = (self._stage_times_host(curr_t, next_t) + self._stage_times_host(curr_t, midpoint_t) +
                        self._stage_times_host(midpoint_t, next_t))
                dev_times = torch.tensor(np.asarray(host, dtype=ts_host.dtype), device=y0.device).to(ts.dtype).unbind(0)
                k = len(host) // 3
                y_full, _ = self.step(curr_t, next_t, curr_y, curr_extra, noise=n_full, times=dev_times[:k])
                y_mid, mid_extra = self.step(curr_t, midpoint_t, curr_y, curr_extra, noise=n_a,
                                             times=dev_times[k:2 * k])
                y_next, next_extra = self.step(midpoint_t, next_t, y_mid, mid_extra, noise=n_b,
                                               times=dev_times[2 * k:])
                with torch.no_grad():
                    error_estimate = _error_estimate(y_full, y_next, self.rtol, self.atol)
                    step_size, prev_error_ratio = _update_step_size(error_estimate, step_size, prev_error_ratio)
                if step_size < self.dt_min:
                    warnings.warn("Hitting minimum allowed step size in adaptive time-stepping.")
                    step_size = self.dt_min
                    prev_error_ratio = None
                if error_estimate <= 1 or step_size <= self.dt_min:
                    prev_t, prev_y = curr_t, curr_y
                    curr_t, curr_y, curr_extra = next_t, y_next, next_extra
            w0 = (curr_t - out_t) / (curr_t - prev_t)
            w1 = (out_t - prev_t) / (curr_t - prev_t)
            ys.append(K.linear_interp(prev_y, curr_y, float(w0), float(w1)))
        return torch.stack(ys, dim=0), curr_extra

    merges_half_steps = True   # False for solvers that draw more than (W, U) per step (Levy area)

    def _step_doubling_noise(self, ta, tm, tb):
        """(whole, first half, second half) increments of one attempt of step doubling from TWO generator queries:
        the whole step's (W, H) is the concatenation of its halves, by the formula the generator itself uses to
        join pieces (brownian_interval.py:647-672) -- the third bridge-tree walk of the reference's loop
        (base_solver.py:120-123) is saved. Foreign Brownian motions keep the reference's three queries."""
        bm = self._native_bm()
        if bm is None or not self.merges_half_steps or self.options.get("general_noise", False):
            return None, None, None
        want_U = self.needs_U and bm._have_H
        cast = self._as_state_dtype
        ha, hb = bm._round(tm) - bm._round(ta), bm._round(tb) - bm._round(tm)
        if not want_U:
            Wa, _ = bm.increment(ta, tm)
            Wb, _ = bm.increment(tm, tb)
            W, _ = K.merge_halves(torch.empty_like(Wa), None, Wa, None, Wb, None, ha, hb)
            return NoiseSpec.external(cast(W)), NoiseSpec.external(cast(Wa)), NoiseSpec.external(cast(Wb))
        Ha = torch.empty(bm.shape, dtype=bm.dtype, device=bm.device)
        Hb = torch.empty_like(Ha)
        Wa, Ua = bm.increment(ta, tm, want_U=True, out_H=Ha)
        Wb, Ub = bm.increment(tm, tb, want_U=True, out_H=Hb)
        # one kernel, the generator's own arithmetic (tsde_bridge.h: interval_merge) -- and the same kernel the
        # device-controlled loop uses, so both forms of the loop see the same increments bit for bit
        W, U = K.merge_halves(torch.empty_like(Wa), torch.empty_like(Wa), Wa, Ha, Wb, Hb, ha, hb)
        return (NoiseSpec.external(cast(W), cast(U)), NoiseSpec.external(cast(Wa), cast(Ua)),
                NoiseSpec.external(cast(Wb), cast(Ub)))

    def _tracks_grad(self, y0):
        return torch.is_grad_enabled() and (y0.requires_grad or any(p.requires_grad for p in self._params()))

    # ---- whole-trajectory kernel (closed-form SDEs) ----------------------------------------------------------
    def _trajectory_code(self):
        """TSDE_TRAJ_* code of this scheme's in-register form, or None if it has none."""
        return None

    def _neural_code(self):
        """TSDE_TRAJ_* code of this scheme in the neural-SDE kernel (`tsde_trajectory_mlp_general`: diagonal, scalar or general
        noise; additive noise: `_additive_code`), or None."""
        return None

    def _deep_code(self):
        """TSDE_TRAJ_* code of this scheme in the deep-network kernel (`tsde_deep_mlp_forward`: nets of up to four Linear
        layers, LipSwish, a closing tanh; Euler, midpoint, Heun, Euler-Heun -- reversible Heun has its own route), or None."""
        return None

    def _program_code(self):
        """TSDE_TRAJ_* code of this scheme in the expression-program kernel (`tsde_trajectory_prog_diag`: diagonal or scalar
        noise), or None."""
        return None

    def _additive_code(self):
        """TSDE_TRAJ_* code of this scheme in the additive-noise kernel (`tsde_trajectory_prog_additive`: Euler, midpoint,
        SRK = SRA1), or None."""
        return None

    # stage times at which a scheme evaluates an additive diffusion, as multiples of dt from t0 (sra1.py: C1 = (1, 0))
    _ADDITIVE_SLOTS = {_native.TRAJ_EULER: (0,), _native.TRAJ_MIDPOINT: (0, 0.5), _native.TRAJ_SRK: (1, 0)}

    def _closed_form_coefficients(self, y0):
        """What `_integrate_trajectory` needs if the whole solve can run as ONE launch of a trajectory kernel, else
        None: a closed-form SDE handed to `sdeint` as is (closed_form.py) and this package's BrownianInterval
        generating the increments. Affine SDEs: the coefficient tensors, or ("differentiable", parameters...) when
        autograd is on (sensitivity kernel); perceptron drift: the ("mlp_diagonal", ...) spec, or
        ("mlp_differentiable", activation, parameters...) when autograd is on (Euler, Milstein: reverse-sweep kernel).
        `options={"trajectory_kernel": False}` keeps the stepwise path."""
        from .sde import ForwardSDE
        if not self.options.get("trajectory_kernel", True) or self.adaptive or self.stateful:
            return None
        base = getattr(self.sde, "_base_sde", None)
        if (type(self.sde) is not ForwardSDE or not hasattr(base, "closed_form")
                or not closed_form.publishes_its_own_dynamics(base)):
            return None
        bm = self._native_bm()
        if (self._trajectory_code() is None or bm is None or y0.dim() != 2 or tuple(bm.shape) != tuple(y0.shape)
                or y0.dtype not in (torch.float32, torch.float64)):
            return None
        spec = base.closed_form(y0.shape[1], y0.dtype, y0.device)
        if spec is None:
            return None
        if spec[0] == "mlp_diagonal":
            # perceptron drift on the matrix cores: sampling kernel (Euler, Milstein, midpoint, SRK); with autograd on,
            # sampling kernel + reverse sweep (Euler, Milstein)
            code = self._trajectory_code()
            if code not in (_native.TRAJ_EULER, _native.TRAJ_MILSTEIN_ITO, _native.TRAJ_MILSTEIN_STRAT,
                            _native.TRAJ_MIDPOINT, _native.TRAJ_SRK) or bm._elem0 % 4 != 0 or y0.numel() >= 2 ** 30:
                return None
            if code == _native.TRAJ_SRK and (not bm._have_H or y0.dtype != torch.float32):
                return None
            if self._tracks_grad(y0):
                # training: Euler and Milstein, through the reverse-sweep kernel; gradients reach y0 and the module's own
                # six parameters, so a subclass with more of them (or shapes the sweep does not take) goes stepwise
                own = list(base.closed_form_parameters())
                hidden = own[1].numel()
                sigmoid = spec[-1][0] == _native.DIFF_SIGMOID         # its reverse sweep exists for Euler only
                too_large = y0.shape[0] * max(y0.shape[1], hidden) >= 2 ** 30        # 32-bit lane offsets in the sweep
                if (code in (_native.TRAJ_MIDPOINT, _native.TRAJ_SRK) or (sigmoid and code != _native.TRAJ_EULER)
                        or hidden % 4 != 0
                        or too_large or {id(p) for p in base.parameters()} != {id(p) for p in own}):
                    return None
                return ("mlp_differentiable", spec[-2], spec[-1]) + tuple(own)
            return spec
        if spec[0] == "elementwise_diagonal":
            # values only: with autograd on, torch differentiates the module's own f, g on the stepwise path
            return None if self._tracks_grad(y0) else spec
        if spec[0] != "affine_diagonal":
            return None
        if self._tracks_grad(y0):
            # gradients flow to y0 and to the module's own four coefficients through the sensitivity kernel; any
            # other parameter on the module (a subclass adding some) would be lost, so such solves go stepwise
            own = list(base.closed_form_parameters())
            if {id(p) for p in base.parameters()} != {id(p) for p in own}:
                return None
            return ("differentiable",) + tuple(own)
        return spec[1:]

    # ---- unchanged user modules whose f and g are per-channel expressions (recognise.py) ----------------------
    _RECOGNISED_ATTR = "_tsde_recognised"

    def _integrate_recognised(self, y0, ts):
        """The solve as ONE trajectory-kernel launch when the user's own, unchanged drift and diffusion turn out to be
        per-channel expressions ``scale * phi(rate * y + shift) + offset`` (recognise.py interprets the code on a
        probe of a few rows at every solve, so the coefficients are this solve's live parameter values); None when that
        does not apply, and the caller goes on to the stepwise path.

        Trust is earned once per (form, scheme, state width, dtype) on each SDE object: the first such solve also runs
        stepwise -- the reference's arithmetic, base_solver.py:143-149 -- and the two must agree to the float32 tolerance
        the closed-form routes are tested to (rtol 1e-4, atol 1e-5; float64: 1e-9, 1e-11); the interpretation must be
        repeatable (same coefficients when run twice) and must leave the object's Python-side state alone (a call
        counter that feeds the coefficients would make one interpretation per solve mean something else than one call
        per step). That first solve returns the stepwise result. A refusal is remembered per Python-side state of the
        object, so code that does not fit costs one interpretation, not one per solve.
        `options={"trajectory_kernel": False}` opts out."""
        from . import graph, recognise
        from .sde import ForwardSDE
        sde = self.sde
        # diagonal noise: every scheme with an in-register form; drift AND diffusion networks (recognise.Recognised.neural:
        # the reference's Neural* problems, any of diagonal / scalar / general noise): Euler and midpoint
        elementwise = sde.noise_type == NOISE_TYPES.diagonal and self._trajectory_code() is not None
        networks = (sde.noise_type in (NOISE_TYPES.diagonal, NOISE_TYPES.scalar, NOISE_TYPES.general)
                    and (self._neural_code() is not None or self._deep_code() is not None))
        # ... and any other elementwise code (several functions of the state summed / multiplied, scalar noise): expression
        # programs (recognise.RecognisedProgram), every scheme with an in-register form
        programs = sde.noise_type in (NOISE_TYPES.diagonal, NOISE_TYPES.scalar) and self._program_code() is not None
        # ... and additive noise: the drift a program, the diffusion tabulated at the scheme's stage times
        additive = (sde.noise_type == NOISE_TYPES.additive and self._additive_code() is not None
                    and not getattr(sde, "user_g_prod", False))
        precision = self.options.get("matrix_precision", "f32")
        if precision not in ("f32", "bf16x3"):
            raise ValueError(f"Expected options['matrix_precision'] in ('f32', 'bf16x3'), got {precision!r}.")
        if (not recognise.ENABLED or not self.options.get("trajectory_kernel", True) or self.adaptive or self.stateful
                or type(sde) is not ForwardSDE or sde.user_product or not (elementwise or networks or programs or additive)):
            return None
        if self._tracks_grad(y0):
            return self._integrate_recognised_with_grad(y0, ts) if (elementwise or programs) else None
        bm = self._native_bm()
        if (bm is None or y0.dim() != 2 or len(bm.shape) != 2 or bm.shape[0] != y0.shape[0] or not y0.is_cuda
                or y0.shape[0] < 8
                or y0.dtype not in (torch.float32, torch.float64) or ts.dtype != y0.dtype or bm.dtype != y0.dtype
                or bm._rootW is not None or bm._rootH is not None or torch.cuda.is_current_stream_capturing()
                or (self._program_code() == _native.TRAJ_SRK and not bm._have_H)):
            return None
        if additive and (not hasattr(sde._base_sde, "f") or not hasattr(sde._base_sde, "g")):
            return None
        if sde.noise_type == NOISE_TYPES.diagonal and tuple(bm.shape) != tuple(y0.shape):
            return None
        if sde.noise_type == NOISE_TYPES.scalar and tuple(bm.shape) != (y0.shape[0], 1):
            return None
        chain, base = graph._wrapper_chain(sde)
        if not self._may_be_interpreted(base):
            return None
        try:
            book = base.__dict__.setdefault(self._RECOGNISED_ATTR, {"refused": {}, "trusted": {}})
        except AttributeError:
            return None
        # Pure call counters (`self._nfe += 1` in the reference's Ex* test problems): not state of the dynamics (graph.
        # call_counters decides that on the bytecode), so they neither refuse the form nor lose their meaning -- the verifying
        # solve learns by how much the stepwise loop advances them per step, and a kernel solve leaves them at that value.
        # `options={"assume_pure": True}` (or the attribute `tsde_assume_pure = True` on the SDE object): the USER vouches that
        # whatever Python-side state their f and g touch (counters, logs, caches) does not reach the dynamics -- the documented
        # switch for modules the checks below would keep stepwise. Nothing is fingerprinted then, counters run once per solve
        # instead of once per step; the both-routes comparison of the first solve (and TSDE_VERIFY_EVERY) still applies.
        assume_pure = self._assume_pure(base)
        counters = {} if assume_pure else graph.call_counters(base)
        ignore = frozenset(counters)
        counter_start = {name: getattr(base, name) for name in counters}
        self._counter_start = (base, counter_start) if counters else None

        def state_of():
            return ("assumed pure",) if assume_pure else graph.python_state(base, ignore=ignore)
        state = None
        if book["refused"]:
            state = state_of()
            if state is None or (state, chain, type(self).__name__) in book["refused"]:
                return None

        def refuse(reason):
            key = (state if state is not None else state_of(), chain, type(self).__name__)
            if key[0] is not None:
                if len(book["refused"]) >= 16:
                    book["refused"].clear()
                book["refused"][key] = reason
            return None

        times = None
        milstein = self._program_code() in (_native.TRAJ_MILSTEIN_ITO, _native.TRAJ_MILSTEIN_STRAT)

        def as_program(first_reason):
            """The second chance of code the single-function forms cannot hold: its expression trees as programs -- and the
            third: channels that read each other (`split` / `cat` of the state's columns), a small ROW-COUPLED system
            (recognise_rows.py: one lane per row, the model generated and compiled at run time)."""
            if not programs:
                raise recognise.NotElementwise(first_reason)
            try:
                found = recognise.recognise_program(sde, ts[0], y0, sde.noise_type)
                spec = found.spec(milstein)
            except recognise.NotElementwise as e:
                from . import recognise_rows
                reason = f"{first_reason}; as an expression program: {e}"
                if (sde.noise_type != NOISE_TYPES.diagonal or milstein or y0.shape[1] > recognise_rows.MAX_D
                        or self._program_code() is None):
                    raise recognise.NotElementwise(reason) from None
                try:
                    found = recognise_rows.recognise_rows(sde, ts[0], y0)
                    return found, found.spec()
                except recognise.NotElementwise as e2:
                    raise recognise.NotElementwise(f"{reason}; as a row-coupled system: {e2}") from None
            book["program"] = (chain, type(self).__name__)
            return found, spec

        try:
            try:
                if additive:
                    times = self._stage_times(ts, y0.device, slots=self._ADDITIVE_SLOTS[self._additive_code()])
                    found = recognise.recognise_additive(sde, ts[0], y0, times)
                    spec = found.spec()
                    if tuple(bm.shape) != (y0.shape[0], found.m):
                        return None
                    raise _Found()
                if book.get("program") == (chain, type(self).__name__):
                    raise recognise.NotElementwise("(remembered: an expression program)")
                if book.get("uses_t") == (chain, type(self).__name__):
                    raise recognise.DependsOnTime("(remembered)")         # skip the pass that is known to end at t
                found = recognise.recognise(sde, ts[0], y0)
            except recognise.DependsOnTime:
                book["uses_t"] = (chain, type(self).__name__)
                # f, g use t in their arithmetic: interpret once more with ALL the times at which this scheme evaluates them
                # (its stage times of every step); the kernels then read one coefficient row per stage time
                times = self._stage_times(ts, y0.device) if elementwise else None
                try:
                    if times is None:
                        raise recognise.NotElementwise("no coefficient tables for this scheme")
                    found = recognise.recognise(sde, ts[0], y0, times=times)
                except recognise.NotElementwise as e:
                    # ... or t takes part in arithmetic that is not affine in the state: t as an operand of a program
                    times = None
                    found, spec = as_program("drift or diffusion depends on t, and not only through arithmetic that "
                                             f"broadcasts ({e})")
            except _Found:
                pass
            except recognise.NotElementwise as e:
                if additive:
                    raise
                found, spec = as_program(str(e))
            if isinstance(found, (recognise.RecognisedProgram, recognise.RecognisedAdditive)) \
                    or getattr(found, "statements", None) is not None:       # (programs, additive tables, row-coupled systems)
                pass
            elif found.neural:
                if not networks or times is not None:
                    raise recognise.NotElementwise("drift and diffusion networks, but no neural-SDE kernel for this scheme")
                try:
                    if self._neural_code() is None:
                        raise recognise.NotElementwise("no two-layer neural-SDE kernel for this scheme")
                    spec = found.neural_spec(sde.noise_type)
                except recognise.NotElementwise as e:
                    # deeper nets, LipSwish, a closing tanh -- or a scheme only the deep kernel has (Heun, Euler-Heun)
                    if self._deep_code() is None or y0.dtype != torch.float32:
                        raise
                    try:
                        spec = found.deep_spec(sde.noise_type)
                    except recognise.NotElementwise as e2:
                        raise recognise.NotElementwise(f"{e}; as deeper networks: {e2}") from None
                if tuple(bm.shape) != (y0.shape[0], spec[4]) or y0.numel() >= 2 ** 30:
                    return None
                if precision == "bf16x3" and sde.noise_type == NOISE_TYPES.general and spec[0] == "neural":
                    # opt-in: the diffusion net's second layer on split-bf16 products (csrc/mlp_general.hip SPLIT); NOT the
                    # reference's arithmetic -- the default, and everything benchmarked as such, stays exact f32
                    spec[2].precision = _native.PRECISION_BF16X3
            elif not elementwise:
                raise recognise.NotElementwise(f"{sde.noise_type} noise whose drift and diffusion are not both networks")
            else:
                try:
                    spec = found.spec()
                except recognise.NotElementwise as e:
                    found, spec = as_program(str(e))
        except recognise.NotElementwise as e:
            return refuse(str(e))
        if spec[0] == "mlp_diagonal":
            # perceptron drift: the sampling kernel's own limits (cf. `_closed_form_coefficients`)
            code = self._trajectory_code()
            if (code not in (_native.TRAJ_EULER, _native.TRAJ_MILSTEIN_ITO, _native.TRAJ_MILSTEIN_STRAT,
                             _native.TRAJ_MIDPOINT, _native.TRAJ_SRK) or bm._elem0 % 4 != 0 or y0.numel() >= 2 ** 30):
                return None
        key = self._recognised_key(found, chain, y0)
        if spec[0] == "neural" and spec[2].precision != _native.PRECISION_F32:
            key = key + ("bf16x3",)
        verdict = book["trusted"].get(key)
        launch = spec[1:] if spec[0] == "affine_diagonal" else spec       # (what `_integrate_trajectory` takes)
        reverify = verdict is True and self._due_for_reverification(book, key)
        if reverify:
            verdict = None
        if verdict is True:
            rate = book.get("counter_rate", {}).get(key) if counters else {}
            if rate is None or set(rate) != set(counters):
                return None
            ys = self._integrate_trajectory(launch, y0, ts)
            if ys is not None and counters:
                n_steps = timegrid.build(timegrid.ts_to_host(ts), self.dt).n_steps
                for name in counters:
                    setattr(base, name, counter_start[name] + rate[name] * n_steps)
            return ys
        if verdict is not None:
            return None
        # first solve of this form: is the interpretation repeatable and free of side effects, and does the kernel
        # reproduce the stepwise solve?
        # (the second interpretation runs on a probe of another height: a coefficient computed from the number of rows
        #  -- `y / y.shape[0]` -- comes out different and the form is refused for what it is, not by a numeric accident)
        before = state_of()
        rng_before = self._rng_states(y0.device)
        try:
            if spec[0] == "program_rows":
                from . import recognise_rows
                again = recognise_rows.recognise_rows(sde, ts[0], y0, rows=5).spec()
            elif spec[0] == "program_diagonal":
                again = recognise.recognise_program(sde, ts[0], y0, sde.noise_type, rows=5).spec(milstein)
            elif spec[0] in ("program_additive", "neural_additive"):
                again = recognise.recognise_additive(sde, ts[0], y0, times, rows=5, check_rows=True).spec()
                # (a diffusion that is a network of t comes out of another matrix-product kernel on the taller probe: its
                #  table is compared to rounding; everything else bit for bit, below)
                tight = dict(rtol=1e-5, atol=1e-7) if y0.dtype == torch.float32 else dict(rtol=1e-12, atol=1e-14)
                if again[3].shape == spec[3].shape and torch.allclose(again[3], spec[3], **tight):
                    again = again[:3] + (spec[3],) + again[4:]
            else:
                again = recognise.recognise(sde, ts[0], y0, times=times, rows=5)
                if spec[0] == "neural_rheun":
                    again = again.deep_spec(sde.noise_type)
                    same_nets = all(a.structure() == b.structure() and all(x is y for x, y in zip(a.parameters(), b.parameters()))
                                    for a, b in zip(again[1:3], spec[1:3]))
                    again = spec if (same_nets and again[3:] == spec[3:]) else again
                else:
                    again = again.neural_spec(sde.noise_type) if spec[0] == "neural" else again.spec()
                if spec[0] == "neural":
                    again[2].precision = spec[2].precision
        except recognise.NotElementwise as e:
            return refuse(str(e))
        if before is None or state_of() != before:
            return refuse("calling f and g changes the object's Python-side state")
        if any(not torch.equal(a, b) for a, b in zip(rng_before, self._rng_states(y0.device))):
            return refuse("calling f and g advances a random number generator")
        same = len(again) == len(spec) and all(
            (torch.equal(a, b) if torch.is_tensor(a) else a == b) for a, b in zip(again, spec))
        if not same:
            self._record_verdict(book, key, "two interpretations of the same code (probes of 2 and 5 rows) gave different "
                                 "coefficients: they depend on the batch size or on how often the code has run", reverify)
            return None
        fast = self._integrate_trajectory(launch, y0, ts)
        if fast is None:
            return None                  # (grid and Brownian cells do not line up: nothing learnt about the form)
        self._extra = ()
        for name, value in counter_start.items():       # (the interpretations' calls are not steps: count the real loop only)
            setattr(base, name, value)
        stepwise = self._run(self._plan(y0, ts), y0)
        if counters:
            n_steps = timegrid.build(timegrid.ts_to_host(ts), self.dt).n_steps
            advanced = {name: getattr(base, name) - counter_start[name] for name in counters}
            if any(v % n_steps for v in advanced.values()):
                self._record_verdict(book, key, "a call counter does not advance by a fixed amount per step", reverify)
                return stepwise
            counter_rate = {name: v // n_steps for name, v in advanced.items()}
        rtol, atol = (1e-4, 1e-5) if y0.dtype == torch.float32 else (1e-9, 1e-11)
        if spec[0] in ("mlp_diagonal", "neural", "neural_additive", "neural_rheun"):      # the matrix cores sum the layers' products in another order than the library
            rtol, atol = 1e-3, 1e-4
        both_nan = fast.isnan() & stepwise.isnan()
        close = ((fast - stepwise).abs() <= atol + rtol * stepwise.abs()) | both_nan | (fast == stepwise)
        self._record_verdict(book, key, True if bool(close.all()) else
                             "the trajectory kernel did not reproduce the stepwise solve", reverify)
        if counters:                        # (after the verdict: recording it prunes the book, rates included, when it is full)
            book.setdefault("counter_rate", {})[key] = counter_rate
        return stepwise

    def _assume_pure(self, base):
        return bool(self.options.get("assume_pure", False) or getattr(base, "tsde_assume_pure", False))

    @staticmethod
    def _rng_states(device):
        """Host-side snapshots of the default CPU and device generators (seed + offset; no device synchronisation)."""
        return torch.get_rng_state(), torch.cuda.get_rng_state(device)

    @staticmethod
    def _may_be_interpreted(base):
        """The interpretation CALLS the user's f and g on a two-row probe. Modules for which one extra call is not
        harmless are left alone: compiled modules (a dispatch mode under torch.compile recompiles or fails), and modules
        with normalisation layers in training mode (a call would feed the probe into their running statistics)."""
        if type(base).__name__ == "OptimizedModule":
            return False
        if isinstance(base, torch.nn.Module):
            for m in base.modules():
                if m.training and isinstance(m, torch.nn.modules.batchnorm._NormBase) \
                        and getattr(m, "track_running_stats", False):
                    return False
        return True

    def _integrate_recognised_with_grad(self, y0, ts):
        """Autograd is recording the solve (`sdeint` with trainable parameters or y0). A recognised module whose drift and
        diffusion are plain `rate * y + shift` with the user's own tensors as coefficients takes the sensitivity kernel
        (`tsde_trajectory_affine_diag_sens`: forward-mode tangents in registers, `backward()` a few reductions) and the
        gradients land on those tensors -- through whatever graph the user's code built on the way to them. Trust as for
        the forward route; the verifying solve compares the kernel's VALUES with the stepwise solve, which is the one
        that is returned (with its graph). None: the stepwise path."""
        from . import graph, recognise
        sde, bm = self.sde, self._native_bm()
        scalar = sde.noise_type == NOISE_TYPES.scalar
        if (bm is None or y0.dim() != 2 or not y0.is_cuda or y0.shape[0] < 8
                or tuple(bm.shape) != ((y0.shape[0], 1) if scalar else tuple(y0.shape))
                or y0.dtype not in (torch.float32, torch.float64) or ts.dtype != y0.dtype or bm.dtype != y0.dtype
                or bm._rootW is not None or bm._rootH is not None or torch.cuda.is_current_stream_capturing()
                or (self._program_code() == _native.TRAJ_SRK and not bm._have_H)):
            return None
        chain, base = graph._wrapper_chain(sde)
        assume_pure = self._assume_pure(base)
        if not self._may_be_interpreted(base) or (not assume_pure and graph.call_counters(base)):
            # (call counters: only the forward route above keeps them at the stepwise loop's value; with autograd recording such
            #  modules stay stepwise, where the counters are right by construction)
            return None
        try:
            book = base.__dict__.setdefault(self._RECOGNISED_ATTR, {"refused": {}, "trusted": {}})
        except AttributeError:
            return None
        if book["refused"]:
            state = ("assumed pure",) if assume_pure else graph.python_state(base)
            if state is None or (state, chain, type(self).__name__) in book["refused"]:
                return None
        leaves = None
        if not scalar and self._trajectory_code() is not None:
            try:
                found = recognise.recognise(sde, ts[0], y0, differentiable=True)
                leaves = found.affine_leaves()
            except recognise.NotElementwise:
                pass    # (the forward route records refusals; a training loop reaches it under no_grad or not at all)
        if leaves is None:
            # anything else that is elementwise: expression programs on dual numbers (tsde_trajectory_prog_diag_sens)
            return self._integrate_program_with_grad(y0, ts, book, chain)
        key = self._recognised_key(found, chain, y0) + ("autograd",)
        verdict = book["trusted"].get(key)
        reverify = verdict is True and self._due_for_reverification(book, key)
        if verdict is True and not reverify:
            return self._integrate_trajectory(("differentiable",) + tuple(leaves), y0, ts)
        if verdict is not None and not reverify:
            return None
        # (as in `_integrate_recognised`: a second interpretation on a probe of another height must give the same values)
        try:
            again = recognise.recognise(sde, ts[0], y0, differentiable=True, rows=5).affine_leaves()
        except recognise.NotElementwise:
            return None
        if again is None or any(a.shape != b.shape or not torch.equal(a.detach(), b.detach()) for a, b in zip(again, leaves)):
            self._record_verdict(book, key, "two interpretations of the same code (probes of 2 and 5 rows) gave different "
                                 "coefficients", reverify)
            return None
        fast = self._integrate_trajectory(("differentiable",) + tuple(leaves), y0, ts)     # values AND a grad_fn
        if fast is None:
            return None
        self._extra = ()
        stepwise = self._run(self._plan(y0, ts), y0)           # recorded by autograd: this is the result
        verdict = self._both_routes_agree(fast, stepwise, y0, "the sensitivity kernel")
        self._record_verdict(book, key, verdict, reverify)
        return stepwise

    def _integrate_program_with_grad(self, y0, ts, book, chain):
        """`_integrate_recognised_with_grad` for code the affine form does not hold: drift and diffusion as expression
        programs, gradients to y0 and to up to four per-channel constants of the user's module (the tensors their code hands
        to its operators: parameters, or what autograd saw it derive from them) through the sensitivity kernel."""
        from . import recognise
        sde = self.sde
        code = self._program_code()
        if code is None or sde.noise_type not in (NOISE_TYPES.diagonal, NOISE_TYPES.scalar):
            return None
        milstein = code in (_native.TRAJ_MILSTEIN_ITO, _native.TRAJ_MILSTEIN_STRAT)
        try:
            found = recognise.recognise_program(sde, ts[0], y0, sde.noise_type, differentiable=True)
            spec = found.spec(milstein)
        except recognise.NotElementwise:
            return None
        rows = found.trainable_rows(None)
        if rows is None:
            return None
        # every trainable parameter of the module must be reached through those constants, or its gradient would be lost
        reached = set()
        for k in rows:
            stack, seen = [found.consts[k].grad_fn], set()
            if found.consts[k].is_leaf:
                reached.add(id(found.consts[k]))
            while stack:
                fn = stack.pop()
                if fn is None or id(fn) in seen:
                    continue
                seen.add(id(fn))
                if hasattr(fn, "variable"):
                    reached.add(id(fn.variable))
                stack.extend(nxt for nxt, _ in fn.next_functions)
        if any(p.requires_grad and id(p) not in reached for p in self._params()):
            return None
        key = self._recognised_key(found, chain, y0) + ("autograd",)
        verdict = book["trusted"].get(key)
        launch = ("program_differentiable", spec[1], spec[2], spec[3], tuple(found.consts), tuple(rows), spec[5])
        reverify = verdict is True and self._due_for_reverification(book, key)
        if verdict is True and not reverify:
            return self._integrate_trajectory(launch, y0, ts)
        if verdict is not None and not reverify:
            return None
        try:
            again = recognise.recognise_program(sde, ts[0], y0, sde.noise_type, rows=5, differentiable=True)
            again_spec = again.spec(milstein)
        except recognise.NotElementwise:
            return None
        if (again.structure() != found.structure()
                or not torch.equal(again_spec[4], spec[4]) or again.trainable_rows(None) != rows):
            self._record_verdict(book, key, "two interpretations of the same code (probes of 2 and 5 rows) gave different "
                                 "programs", reverify)
            return None
        fast = self._integrate_trajectory(launch, y0, ts)       # values AND a grad_fn (the program sensitivity kernel)
        if fast is None:
            return None
        self._extra = ()
        stepwise = self._run(self._plan(y0, ts), y0)           # recorded by autograd: this is the result
        verdict = self._both_routes_agree(fast, stepwise, y0, "the program sensitivity kernel")
        self._record_verdict(book, key, verdict, reverify)
        return stepwise


    # ---- the verifying solve: values AND gradients -----------------------------------------------------------------
    def _both_routes_agree(self, fast, stepwise, y0, what, network=False, extra_inputs=()):
        """True, or the reason the kernel route is not to be trusted. `fast` and `stepwise` are the two routes' results of
        the SAME solve, both with their autograd graphs. Values: elementwise, at the tolerances of the forward route. Gradients:
        d<ys, r>/d(y0, every trainable parameter) for one fixed random cotangent r, the kernel's sensitivities against
        ordinary autograd through the stepwise loop -- what the reference computes (base_solver.py:143-149 under autograd,
        sdeint.py:27-112) -- relative to the largest entry of each gradient. (One extra backward pass through the stepwise
        graph, which is retained for the caller; once per form, scheme, batch size and SDE object.)"""
        f32 = y0.dtype == torch.float32
        rtol, atol = (1e-4, 1e-5) if f32 else (1e-9, 1e-11)
        if network:                          # (the matrix cores sum the layers' products in another order than the library)
            rtol, atol = 1e-3, 1e-4
        ref, got = stepwise.detach(), fast.detach()
        # (the absolute part scales with the solution: an element that crosses zero in a solve of magnitude 100 carries the
        #  rounding of its neighbours, not of its own size)
        finite = torch.where(torch.isfinite(ref), ref.abs(), torch.zeros_like(ref))
        atol = atol * torch.clamp(finite.max(), min=1.0)
        close = ((got - ref).abs() <= atol + rtol * ref.abs()) | (got.isnan() & ref.isnan()) | (got == ref)
        if not bool(close.all()):
            return f"{what}'s values differ from the stepwise solve"
        inputs = ([y0] if y0.requires_grad else []) + [p for p in self._params() if p.requires_grad]
        inputs += [p for p in extra_inputs if p.requires_grad and all(p is not q for q in inputs)]
        if not inputs or fast.grad_fn is None or stepwise.grad_fn is None:
            return True if fast.grad_fn is None and stepwise.grad_fn is None else \
                f"{what}: one route carries a gradient and the other does not"
        gen = torch.Generator(device=y0.device)
        gen.manual_seed(0x5DE)
        r = torch.randn(ref.shape, generator=gen, device=y0.device, dtype=y0.dtype)
        want = torch.autograd.grad((stepwise * r).sum(), inputs, retain_graph=True, allow_unused=True)
        have = torch.autograd.grad((fast * r).sum(), inputs, allow_unused=True)
        g_rtol, g_atol = (5e-3 if network else 2e-3, 1e-6) if f32 else (1e-8, 1e-12)
        for i, (w, h) in enumerate(zip(want, have)):
            w = torch.zeros_like(inputs[i]) if w is None else w
            h = torch.zeros_like(inputs[i]) if h is None else h
            scale = w.abs().max()
            bad = ((h - w).abs().max() > g_rtol * scale + g_atol) | ~torch.isfinite(h).all()
            if bool(bad) and bool(torch.isfinite(w).all()):
                name = "y0" if (i == 0 and y0.requires_grad) else f"parameter {i - (1 if y0.requires_grad else 0)}"
                return (f"{what}'s gradient with respect to {name} differs from autograd through the stepwise solve "
                        f"(max error {float((h - w).abs().max()):.3e} at scale {float(scale):.3e})")
        return True

    @staticmethod
    def _record_verdict(book, key, verdict, reverify=False):
        if len(book["trusted"]) >= 32:
            book["trusted"].clear()
            book.get("counter_rate", {}).clear()
            book.get("solves", {}).clear()
        book["trusted"][key] = verdict
        if reverify and verdict is not True:
            # TSDE_VERIFY_EVERY: a form that had earned trust and no longer reproduces the stepwise solve is a loud failure
            raise RuntimeError(f"torchsde_amd: periodic re-verification (TSDE_VERIFY_EVERY) of a trusted kernel route failed: "
                               f"{verdict}. Results of earlier solves of this object on that route are suspect; "
                               "options={'trajectory_kernel': False} keeps the stepwise path.")

    @staticmethod
    def _due_for_reverification(book, key):
        """TSDE_VERIFY_EVERY=N (or `solvers.VERIFY_EVERY`): every N-th solve of a trusted form runs both routes again and
        compares (values, and gradients where autograd records) -- a mis-recognition fails loudly in CI instead of quietly
        in training. 0 (the default): only the first solve verifies."""
        if VERIFY_EVERY <= 0:
            return False
        count = book.setdefault("solves", {})
        count[key] = count.get(key, 0) + 1
        return count[key] % VERIFY_EVERY == 0

    _STAGE_TIMES = {}
    # stage-time slots of the trajectory kernels (csrc/trajectory.hip stage_slots): offsets from t0 as multiples of dt
    _TIMED_SLOTS = {_native.TRAJ_EULER: (0,), _native.TRAJ_MILSTEIN_ITO: (0,), _native.TRAJ_MILSTEIN_STRAT: (0,),
                    _native.TRAJ_MIDPOINT: (0, 0.5), _native.TRAJ_SRK: (0, 0.25, 0.5, 1),
                    _native.TRAJ_HEUN: (0, 1), _native.TRAJ_EULER_HEUN: (0, 1)}

    def _stage_times(self, ts, device, slots=None):
        """Every time at which this scheme evaluates f and g during the solve -- (K * S,) in ts.dtype on the device, the S
        stage times of step 0, then of step 1, ... -- computed like `_plan` computes the times the stepwise loop hands to
        the user's code (`t0 + frac * dt` in ts.dtype). Remembered by content."""
        slots = self._TIMED_SLOTS.get(self._trajectory_code()) if slots is None else slots
        grid = timegrid.build(timegrid.ts_to_host(ts), self.dt)
        if slots is None or grid.n_steps == 0:
            return None
        key = (grid.t.tobytes(), str(grid.t.dtype), slots, str(device))
        hit = self._STAGE_TIMES.get(key)
        if hit is None:
            if len(self._STAGE_TIMES) >= 16:
                self._STAGE_TIMES.clear()
            np_dtype = grid.t.dtype.type
            table = np.empty((grid.n_steps, len(slots)), dtype=grid.t.dtype)
            for j, frac in enumerate(slots):
                table[:, j] = grid.t[:-1] if frac == 0 else grid.t[:-1] + np_dtype(frac) * grid.dt
            hit = torch.from_numpy(table.reshape(-1)).to(device=device)
            if hit.dtype != ts.dtype:
                hit = hit.to(ts.dtype)
            self._STAGE_TIMES[key] = hit
        return hit

    def _recognised_key(self, found, chain, y0):
        # The batch size is part of the key: the interpretation runs on a probe of a few rows, so whatever the user's code
        # derives from `y.shape[0]` (`-y if y.shape[0] > 1000 else -2 * y`) is evaluated for the probe; the both-routes
        # comparison that earns the trust therefore has to be made at every batch size the form is solved at.
        return (found.structure(), chain, type(self).__name__, self.sde.sde_type, y0.shape[1], y0.dtype, y0.shape[0])

    def recognised_perceptron(self, y0, ts):
        """For `sdeint_adjoint` (mlp_adjoint.route): the interpretation of an unchanged user module whose drift is a
        two-layer perceptron -- `recognise.Recognised` -- if this solver's forward solve of it through the sampling
        kernel is TRUSTED (verified against the stepwise solve; the check runs here, once, if it has not yet), else None."""
        from . import graph, recognise
        from .sde import ForwardSDE
        sde = self.sde
        if (not recognise.ENABLED or type(sde) is not ForwardSDE or sde.user_product
                or sde.noise_type != NOISE_TYPES.diagonal or y0.dim() != 2 or not y0.is_cuda):
            return None
        chain, base = graph._wrapper_chain(sde)
        assume_pure = self._assume_pure(base)
        if not self._may_be_interpreted(base) or (not assume_pure and graph.call_counters(base)):
            return None
        book = getattr(base, self._RECOGNISED_ATTR, None)
        if book is not None and book["refused"]:
            state = ("assumed pure",) if assume_pure else graph.python_state(base)
            if state is None or (state, chain, type(self).__name__) in book["refused"]:
                return None
        try:
            # (differentiable=True: the adjoint kernels differentiate the recognised network -- a stop-gradient in the user's
            #  code, which adjoint_sde.py:111-128 would honour, ends the interpretation: recognise.check_stop_gradient)
            found = recognise.recognise(sde, ts[0], y0.detach(), differentiable=True)
            if not found.perceptron:
                return None
            found.perceptron_spec()
        except recognise.NotElementwise:
            return None
        key = self._recognised_key(found, chain, y0)
        if book is None or key not in book["trusted"]:
            with torch.no_grad():                                  # the verifying solve (both routes, compared)
                self._integrate_recognised(y0.detach(), ts)
            book = getattr(base, self._RECOGNISED_ATTR, None)
            # ... and the DERIVATIVES the adjoint kernels will stand for: vector-Jacobian products of the user's f and g
            # (what adjoint_sde.py:111-128, 218-230 asks autograd for at every backward step) against those of the recognised
            # network built from the same parameter tensors, on real rows of this solve
            if book is not None and book["trusted"].get(key) is True:
                reason = self._perceptron_derivatives_agree(found, y0, ts)
                if reason is not True:
                    book["trusted"][key] = reason
        return found if book is not None and book["trusted"].get(key) is True else None

    def _perceptron_derivatives_agree(self, found, y0, ts):
        own = found.perceptron_parameters()
        if own is None:
            return True                     # (mlp_adjoint.route refuses such a module on its own)
        kind, amplitude, _, _ = found.perceptron_diffusion()
        w1, b1, w2, b2, rate, shift = own
        probe = y0.detach()[:32].clone().requires_grad_(True)
        gen = torch.Generator(device=y0.device)
        gen.manual_seed(0x5DE)
        r_f = torch.randn(probe.shape, generator=gen, device=y0.device, dtype=y0.dtype)
        r_g = torch.randn(probe.shape, generator=gen, device=y0.device, dtype=y0.dtype)
        with torch.enable_grad():
            f_user, g_user = self.sde.f_and_g(ts[0], probe)
            hidden = torch.addmm(b1, probe, w1.t())
            hidden = torch.tanh(hidden) if found.f.act == "tanh" else torch.nn.functional.softplus(hidden)
            f_kernel = torch.addmm(b2, hidden, w2.t())
            z = rate * probe + shift
            g_kernel = z if kind == _native.DIFF_AFFINE else amplitude * torch.sigmoid(z)
            inputs = [probe] + [p for p in own if p.requires_grad]
            want = torch.autograd.grad((f_user * r_f).sum() + (g_user * r_g).sum(), inputs, allow_unused=True)
            have = torch.autograd.grad((f_kernel * r_f).sum() + (g_kernel * r_g).sum(), inputs, allow_unused=True)
        for i, (w, h) in enumerate(zip(want, have)):
            w = torch.zeros_like(inputs[i]) if w is None else w
            h = torch.zeros_like(inputs[i]) if h is None else h
            if bool((h - w).abs().max() > 1e-4 * w.abs().max() + 1e-6):
                return ("the derivatives of the recognised network differ from autograd through the user's f and g "
                        f"(input {i} of [state, lin1.weight, lin1.bias, lin2.weight, lin2.bias, rate, shift])")
        return True

    def recognised_route(self):
        """{form key: True | reason} and {state: reason} of the SDE object this solver integrates (diagnostics)."""
        from . import graph
        _, base = graph._wrapper_chain(self.sde)
        return getattr(base, self._RECOGNISED_ATTR, None)

    def _integrate_trajectory(self, coefficients, y0, ts):
        """All steps in one kernel launch; None if the Brownian motion's cells do not line up with the steps."""
        bm = self.bm
        grid = timegrid.build(timegrid.ts_to_host(ts), self.dt)
        if grid.n_steps == 0:
            return None
        t64 = grid.t_f64()
        bm.adopt_grid(t64)
        cells = bm.match_grid(t64)
        if cells is None:
            return None
        cells = np.asarray(cells, dtype=np.int64)
        h = bm._edges[cells + 1] - bm._edges[cells]
        # (the step rows of a grid the process has just seen -- every iteration of a training loop: remembered on the
        #  grid object, which timegrid.build hands back for equal (ts, dt); valid for these cell widths)
        memo = getattr(grid, "_step_rows", None)
        if memo is not None and memo[0].shape == h.shape and np.array_equal(memo[0], h):
            rows, out_step, out_w = memo[1:]
        else:
            np_dtype = grid.t.dtype.type
            dt = grid.dt
            rows = np.zeros((grid.n_steps, 8), dtype=np.float64)
            # each entry is rounded in ts.dtype like the stepwise path's scalars, then (below) cast to the state dtype
            rows[:, 0] = dt
            rows[:, 1] = np_dtype(0.5) * dt
            rows[:, 2] = np_dtype(1) / dt
            rows[:, 3] = np.sqrt(dt)
            rows[:, 4] = np.sqrt(h)
            rows[:, 5] = np.sqrt(h / 12.0)
            rows[:, 6] = h
            rows[:, 7] = grid.t[:-1]         # t_k, the time a step starts at (read by tsde_trajectory_mlp_general only)
            out_step = [kc for (_, kc, _, _) in grid.outputs]
            out_w = [(w0, w1) for (_, _, w0, w1) in grid.outputs]
            grid._step_rows = (h.copy(), rows, out_step, out_w)
        if coefficients[0] == "mlp_differentiable":
            if any(not (w0 == 0.0 and w1 == 1.0) for (w0, w1) in out_w):
                return None
            every_step = list(range(1, grid.n_steps + 1))
            schedule_all = K.TrajectorySchedule.cached(rows, cells, every_step, [(0.0, 1.0)] * grid.n_steps, y0.device,
                                                       y0.dtype)
            return K.trajectory_mlp_diag_differentiable(y0, coefficients[3:], coefficients[1], coefficients[2],
                                                        self._trajectory_code(), schedule_all, out_step, bm)
        schedule = K.TrajectorySchedule.cached(rows, cells, out_step, out_w, y0.device, y0.dtype)
        if coefficients[0] == "program_differentiable":
            _, f_code, g_code, dg_code, const_values, rows_with_grad, scalar_noise = coefficients
            return K.trajectory_prog_diag_differentiable(y0, (f_code, g_code, dg_code), const_values, rows_with_grad,
                                                         scalar_noise, self._program_code(), schedule, bm)
        if coefficients[0] == "program_rows":
            y0c = y0.detach() if y0.is_contiguous() else y0.detach().contiguous()
            ys = torch.empty((len(grid.outputs) + 1,) + tuple(y0.shape), dtype=y0.dtype, device=y0.device)
            ys[0].copy_(y0c)
            done = K.trajectory_rows(ys[1:], y0c, coefficients[1], coefficients[2], self._program_code(), schedule, bm)
            return None if done is None else ys          # (None: the generated unit is still compiling -- stepwise for now)
        if coefficients[0] == "program_diagonal":
            y0c = y0.detach() if y0.is_contiguous() else y0.detach().contiguous()
            ys = torch.empty((len(grid.outputs) + 1,) + tuple(y0.shape), dtype=y0.dtype, device=y0.device)
            ys[0].copy_(y0c)
            K.trajectory_prog_diag(ys[1:], y0c, coefficients[1], coefficients[2], coefficients[3], coefficients[4],
                                   coefficients[5], self._program_code(), schedule, bm)
            return ys
        if coefficients[0] in ("program_additive", "neural_additive"):
            kind, f_code, const_table, table, m = coefficients
            code = self._additive_code()
            if table.dim() == 3:
                slots = len(self._ADDITIVE_SLOTS[code])
                if table.shape[0] != grid.n_steps * slots:
                    return None
                table = table.view(grid.n_steps, slots, m, y0.shape[1])
            if kind == "neural_additive" and y0.numel() >= 2 ** 30:
                return None
            y0c = self._aligned_start(y0)
            ys = torch.empty((len(grid.outputs) + 1,) + tuple(y0.shape), dtype=y0.dtype, device=y0.device)
            ys[0].copy_(y0c)
            if kind == "neural_additive":
                K.trajectory_mlp_additive(ys[1:], y0c, f_code, table, m, code, schedule, bm)       # (f_code: the drift net)
            else:
                K.trajectory_prog_additive(ys[1:], y0c, f_code, const_table, table, m, code, schedule, bm)
            return ys
        if coefficients[0] == "neural_rheun":
            # (a stateless scheme on the deep-network kernel, csrc/tsde_neural_rheun.h)
            from . import neural_rheun
            y0c = self._aligned_start(y0)
            ys = torch.empty((len(grid.outputs) + 1,) + tuple(y0.shape), dtype=y0.dtype, device=y0.device)
            ys[0].copy_(y0c)
            times = neural_rheun._device_times(np.ascontiguousarray(grid.t, dtype=np.float32), y0.device)
            neural_rheun.forward(ys[1:], torch.empty_like(y0c), y0c, coefficients[1], coefficients[2], coefficients[3],
                                 coefficients[4], schedule, times, bm, method=self._deep_code())
            return ys
        if coefficients[0] == "neural":
            y0c = self._aligned_start(y0)
            ys = torch.empty((len(grid.outputs) + 1,) + tuple(y0.shape), dtype=y0.dtype, device=y0.device)
            ys[0].copy_(y0c)
            K.trajectory_mlp_general(ys[1:], y0c, coefficients[1], coefficients[2], coefficients[3], coefficients[4],
                                     self._neural_code(), schedule, bm)
            return ys
        if coefficients[0] == "mlp_diagonal":
            y0c = self._aligned_start(y0)
            ys = torch.empty((len(grid.outputs) + 1,) + tuple(y0.shape), dtype=y0.dtype, device=y0.device)
            ys[0].copy_(y0c)
            K.trajectory_mlp_diag(ys[1:], y0c, *coefficients[1:], self._trajectory_code(), schedule, bm)
            return ys
        if coefficients[0] == "elementwise_diagonal":
            y0c = y0.detach() if y0.is_contiguous() else y0.detach().contiguous()
            ys = torch.empty((len(grid.outputs) + 1,) + tuple(y0.shape), dtype=y0.dtype, device=y0.device)
            ys[0].copy_(y0c)
            K.trajectory_expr_diag(ys[1:], y0c, coefficients[1], coefficients[2], coefficients[3:],
                                   self._trajectory_code(), schedule, bm)
            return ys
        if coefficients[0] == "differentiable":
            return K.trajectory_affine_diag_differentiable(y0, coefficients[1:], self._trajectory_code(), schedule, bm)
        y0c = y0.detach() if y0.is_contiguous() else y0.detach().contiguous()
        ys = torch.empty((len(grid.outputs) + 1,) + tuple(y0.shape), dtype=y0.dtype, device=y0.device)
        ys[0].copy_(y0c)
        K.trajectory_affine_diag(ys[1:], y0c, *coefficients, self._trajectory_code(), schedule, bm)
        return ys

    @staticmethod
    def _aligned_start(y0):
        """y0 as the network kernels read it: detached, contiguous, and -- an offset view of a larger tensor need not be --
        on a 16-byte boundary (their C entry points refuse anything else when rows are read as 16-byte groups)."""
        y0c = y0.detach() if y0.is_contiguous() else y0.detach().contiguous()
        return y0c.clone() if y0c.data_ptr() % 16 else y0c

    def _plan(self, y0, ts):
        """Host-side preparation of a solve: time grid, stage times (one upload), Brownian cell map, output map.
        Everything that synchronises or copies from the host happens here, none of it in `_run`."""
        device = y0.device
        grid = timegrid.build(timegrid.ts_to_host(ts), self.dt)
        n_steps = grid.n_steps
        np_dtype = grid.t.dtype.type
        t64 = grid.t_f64()
        # times[k][j] = t_k + frac_j * dt_k, rounded like the reference's 0-d tensor arithmetic (`t0 + C * dt`)
        fracs = self.stage_fracs
        stage = np.empty((max(n_steps, 1), len(fracs) + 1), dtype=grid.t.dtype)
        stage[:n_steps, -1] = grid.t[1:]     # the exact step end t1 (not t0 + dt), last entry of `_Step.times`
        for j, frac in enumerate(fracs):
            if frac == 0:
                stage[:n_steps, j] = grid.t[:-1]
            else:
                stage[:n_steps, j] = grid.t[:-1] + np_dtype(frac) * grid.dt
        stage_dev = torch.from_numpy(stage).to(device=device)
        if stage_dev.dtype != ts.dtype:
            stage_dev = stage_dev.to(ts.dtype)
        stage_rows = [row.unbind(0) for row in stage_dev.unbind(0)] if n_steps > 0 else []
        t_dev = None   # step boundaries as tensors, only needed for foreign Brownian objects
        bm = self._native_bm()
        cells = None
        if bm is not None and n_steps > 0:
            bm.adopt_grid(t64)
            cells = bm.match_grid(t64)
            if cells is None or not self.merges_half_steps or self.options.get("general_noise", False):
                bm._device_edges()   # upload the cell edges now: `_run` must not copy from the host
        elif n_steps > 0:
            t_dev = torch.from_numpy(grid.t.copy()).to(device=device).unbind(0)
        done_at = {}   # step index -> outputs it completes; exact hits are written in place
        for j, (kp, kc, w0, w1) in enumerate(grid.outputs):
            done_at.setdefault(kc, []).append((j + 1, kp, w0, w1))
        return dict(grid=grid, t64=t64, stage_rows=stage_rows, t_dev=t_dev, cells=cells, done_at=done_at,
                    T=len(grid.outputs) + 1)

    def _run(self, plan, y0):
        """Launch-only part of a solve (no host sync, no host->device copy): capturable in a HIP graph."""
        grid, t64, stage_rows, t_dev, cells, done_at, T = (plan[k] for k in ("grid", "t64", "stage_rows", "t_dev",
                                                                              "cells", "done_at", "T"))
        device = y0.device
        y0c = y0 if y0.is_contiguous() else y0.contiguous()
        track = self._tracks_grad(y0)
        ys_buf = None if track else torch.empty((T,) + tuple(y0.shape), dtype=y0.dtype, device=device)
        ys_list = [y0c] + [None] * (T - 1)
        if ys_buf is not None:
            ys_buf[0].copy_(y0c.detach())
            scratch = (torch.empty_like(ys_buf[0]), torch.empty_like(ys_buf[0]))
        in_place = ys_buf is not None

        cur = y0c
        for k in range(grid.n_steps):
            slot = None
            if in_place:
                slot = scratch[k & 1]
                for (i, kp, w0, w1) in done_at.get(k + 1, ()):
                    if w1 == 1.0 and w0 == 0.0:
                        slot = ys_buf[i]
            noise = self._noise_for(t64[k], t64[k + 1], None if t_dev is None else t_dev[k],
                                    None if t_dev is None else t_dev[k + 1],
                                    None if cells is None else int(cells[k]))
            st = _Step(stage_rows[k], grid.dt[k], noise, t64[k + 1] - t64[k], t64[k])
            nxt = self._advance(cur, st, slot)
            if in_place and (nxt.requires_grad or (slot is not None and nxt.data_ptr() != slot.data_ptr())):
                # A tensor that requires grad appeared mid-solve (e.g. a non-Parameter leaf inside the SDE):
                # the autograd wrappers allocate their own outputs, so assemble `ys` with torch.stack instead.
                in_place = False
            for (i, kp, w0, w1) in done_at.get(k + 1, ()):
                if w1 == 1.0 and w0 == 0.0:
                    ys_list[i] = nxt
                else:
                    dst = ys_buf[i] if in_place else None
                    ys_list[i] = K.linear_interp(cur, nxt, w0, w1, out=dst)
            cur = nxt
        for i in range(1, T):
            if ys_list[i] is None:   # output times that need no step (cannot happen for increasing ts)
                ys_list[i] = cur
        if in_place:
            return ys_buf
        return torch.stack(ys_list, dim=0)

    def _params(self):
        try:
            return list(self.sde.parameters())
        except AttributeError:
            return []

    # ---- helpers shared by the concrete methods ------------------------------------------------------------
    def _diag(self):
        return self.sde.noise_type == NOISE_TYPES.diagonal

    def _drift_diffusion_update(self, t, y, cf, cg, noise, out):
        """y + cf*f(t,y) + cg*g(t,y).dW with the product fused unless the user computes it."""
        sde = self.sde
        if sde.user_product:
            W, _ = noise.materialise()
            f, gp = sde.f_and_g_prod(t, y, W)
            return K.step_prod(y, f, gp, cf, cg, out=out)
        f, g = sde.f_and_g(t, y)
        if self._diag():
            return K.step_diag(y, f, g, cf, cg, noise, out=out)
        return K.step_general(y, f, g, cf, cg, noise, out=out)


class Euler(BaseSDESolver):
    """Euler-Maruyama (reference: methods/euler.py:19-37)."""
    weak_order = 1.0
    sde_type = SDE_TYPES.ito
    noise_types = NOISE_TYPES.all()
    levy_area_approximations = LEVY_AREA_APPROXIMATIONS.all()

    def __init__(self, sde, **kwargs):
        self.strong_order = 1.0 if sde.noise_type == NOISE_TYPES.additive else 0.5
        super().__init__(sde=sde, **kwargs)

    def _trajectory_code(self):
        return _native.TRAJ_EULER if self._diag() else None

    def _neural_code(self):
        return _native.TRAJ_EULER

    def _deep_code(self):
        return _native.TRAJ_EULER

    def _program_code(self):
        return _native.TRAJ_EULER

    def _additive_code(self):
        return _native.TRAJ_EULER

    def _advance(self, y0, st, out):
        return self._drift_diffusion_update(st.times[0], y0, st.dt, 1.0, st.noise, out)


class Midpoint(BaseSDESolver):
    """Stratonovich midpoint (reference: methods/midpoint.py:19-45)."""
    weak_order = 1.0
    sde_type = SDE_TYPES.stratonovich
    noise_types = NOISE_TYPES.all()
    levy_area_approximations = LEVY_AREA_APPROXIMATIONS.all()
    stage_fracs = (0, 0.5)

    def __init__(self, sde, **kwargs):
        self.strong_order = 0.5 if sde.noise_type == NOISE_TYPES.general else 1.0
        super().__init__(sde=sde, **kwargs)

    def _trajectory_code(self):
        return _native.TRAJ_MIDPOINT if self._diag() else None

    def _neural_code(self):
        return _native.TRAJ_MIDPOINT

    def _deep_code(self):
        return _native.TRAJ_MIDPOINT

    def _program_code(self):
        return _native.TRAJ_MIDPOINT

    def _additive_code(self):
        return _native.TRAJ_MIDPOINT

    def _advance(self, y0, st, out):
        dt, half_dt = st.dt, st.half_dt
        if self.sde.user_product:
            W, _ = st.noise.materialise()
            f, gp = self.sde.f_and_g_prod(st.times[0], y0, W)
            y_prime = K.step_prod(y0, f, gp, half_dt, 0.5)
            f2, gp2 = self.sde.f_and_g_prod(st.times[1], y_prime, W)
            return K.step_prod(y0, f2, gp2, dt, 1.0, out=out)
        diag = self._diag()
        f, g = self.sde.f_and_g(st.times[0], y0)
        upd = K.step_diag if diag else K.step_general
        y_prime = upd(y0, f, g, half_dt, 0.5, st.noise)
        f2, g2 = self.sde.f_and_g(st.times[1], y_prime)
        return upd(y0, f2, g2, dt, 1.0, st.noise, out=out)


class ReversibleHeun(BaseSDESolver):
    """Reversible Heun (reference: methods/reversible_heun.py:33-73): carries (f, g, z) between steps."""
    weak_order = 1.0
    sde_type = SDE_TYPES.stratonovich
    noise_types = NOISE_TYPES.all()
    levy_area_approximations = LEVY_AREA_APPROXIMATIONS.all()
    stateful = True

    def __init__(self, sde, **kwargs):
        self.strong_order = 1.0 if sde.noise_type == NOISE_TYPES.additive else 0.5
        super().__init__(sde=sde, **kwargs)

    wants_extra = True        # (sdeint clears it when the caller did not ask for `extra=True`)

    def init_extra_solver_state(self, t0, y0):
        self._own_init = tuple(self.sde.f_and_g(t0, y0)) + (y0,)
        return self._own_init

    def integrate(self, y0, ts, extra0):
        """Drift and diffusion both perceptrons of (t, y) (recognise.deep_spec): the whole solve as ONE launch of the
        reversible-Heun kernel, with autograd recording too (its backward pass is the pair's exact-gradient sweep on the
        matrix cores, neural_rheun.py) -- else the stepwise loop."""
        done = self._integrate_on_kernels(y0, ts, extra0)
        return done if done is not None else super().integrate(y0, ts, extra0)

    def _integrate_on_kernels(self, y0, ts, extra0):
        from . import neural_rheun_route
        own = getattr(self, "_own_init", None)
        if self.adaptive or own is None or extra0 is None or len(extra0) != 3 or any(a is not b for a, b in zip(extra0, own)):
            return None               # (a state handed in by the caller: the kernel starts from z_0 = y_0, f_0 = f(t_0, y_0))
        tracks = self._tracks_grad(y0)
        if tracks and self.wants_extra:
            return None               # (the final (f, g, z) with a graph: the stepwise loop has it)
        route = neural_rheun_route.plan(self, y0, ts, differentiable=tracks)
        if route is None:
            return None
        z_last = []
        ys = route.solve(y0, z_last)
        if not route.trusted:
            stepwise, extras = super().integrate(y0, ts, extra0)
            route.record(ys, stepwise, y0)
            return stepwise, extras
        if not self.wants_extra:
            return ys, ()
        with torch.no_grad():          # (f, g, z) after the last step (values; `extra=True` without autograd)
            return ys, tuple(self.sde.f_and_g(ts[-1], z_last[0])) + (z_last[0],)

    def _advance(self, y0, st, out):
        f0, g0, z0 = self._extra
        sde, dt, noise, half_dt = self.sde, st.dt, st.noise, st.half_dt
        t1 = st.times[-1]
        if self._diag():
            z1 = K.rheun_z(y0, z0, f0, g0, dt, 1.0, noise)
            f1, g1 = sde.f_and_g(t1, z1)
            y1 = K.rheun_y(y0, f0, f1, g0, g1, half_dt, 1.0, noise, out=out)
        else:
            base = K.lincomb2(y0, z0, 2.0, -1.0)
            z1 = K.step_general_weighted(base, f0, g0, 1.0, dt, 1.0, 0, 0.0, 0.0, 0.0, noise)
            f1, g1 = sde.f_and_g(t1, z1)
            y1 = K.step_general_weighted(y0, K.lincomb2(f0, f1, 1.0, 1.0), K.lincomb2(g0, g1, 1.0, 1.0), 1.0, half_dt,
                                         0.5, 0, 0.0, 0.0, 0.0, noise, out=out)
        self._extra = (f1, g1, z1)
        return y1


class _Milstein(BaseSDESolver):
    """Milstein, derivative-using or derivative-free (reference: methods/milstein.py:21-94)."""
    strong_order = 1.0
    weak_order = 1.0
    noise_types = (NOISE_TYPES.additive, NOISE_TYPES.diagonal, NOISE_TYPES.scalar)
    levy_area_approximations = LEVY_AREA_APPROXIMATIONS.all()
    ito = True

    def __init__(self, sde, options, **kwargs):
        from . import adjoint  # circular: the adjoint module builds solvers
        if METHOD_OPTIONS.grad_free not in options:
            options[METHOD_OPTIONS.grad_free] = False
        if options[METHOD_OPTIONS.grad_free] and sde.noise_type == NOISE_TYPES.additive:
            options[METHOD_OPTIONS.grad_free] = False   # dg = 0: the derivative form already handles it
        if options[METHOD_OPTIONS.grad_free] and isinstance(sde, adjoint.AdjointSDE):
            raise ValueError(f"Derivative-free Milstein cannot be used for adjoint SDEs, because it requires "
                             f"direct access to the diffusion, whilst adjoint SDEs rely on a more efficient "
                             f"diffusion-vector product. Use derivative-using Milstein instead: "
                             f"`adjoint_options=dict({METHOD_OPTIONS.grad_free}=False)`")
        if sde.noise_type == NOISE_TYPES.general and options.get("general_noise", False):
            # Opt-in EXTENSION (the reference rejects this combination, milstein.py:25): Milstein for general
            # noise with the iterated integrals I_kl = (W_k W_l - delta_kl dt)/2 + A_kl.
            self.noise_types = tuple(NOISE_TYPES.all())
        super().__init__(sde=sde, options=options, **kwargs)

    def _trajectory_code(self):
        if not self._diag() or self.options[METHOD_OPTIONS.grad_free]:
            return None
        return _native.TRAJ_MILSTEIN_ITO if self.ito else _native.TRAJ_MILSTEIN_STRAT

    def _program_code(self):
        if self.options[METHOD_OPTIONS.grad_free]:
            return None
        return _native.TRAJ_MILSTEIN_ITO if self.ito else _native.TRAJ_MILSTEIN_STRAT

    def _additive_code(self):
        # additive noise: the correction is zero (base_sde.py:157-158) and the step is Euler's (`_advance`)
        return _native.TRAJ_EULER

    def _row_noise(self, noise, d):
        """Scalar noise: one increment per batch row, broadcast over the d state channels."""
        W, U = noise.materialise()
        return NoiseSpec.external(W.reshape(-1), None if U is None else U.reshape(-1), bcast_d=d)

    def _advance_general(self, y0, st, out):
        """y1 = y0 + f dt + g.dW + sum_{j,k,l} dg_{i,l}/dy_j g_{j,k} I_{k,l}  (extension; see __init__)."""
        sde, dt = self.sde, st.dt
        t0 = st.times[0]
        bm = self._native_bm()
        if bm is not None and bm._have_A:
            W, _, integrals = bm.increment_with_levy_area(st.t0_64, st.t0_64 + st.h64, iterated=(float(dt), self.ito))
        else:
            W, _ = st.noise.materialise()
            integrals = K.iterated_integrals(W, None, dt, self.ito)
        f, g = sde.f_and_g(t0, y0)
        if self.options[METHOD_OPTIONS.grad_free]:
            # Derivative-free form (the reference's own idea for diagonal noise, milstein.py:58-67, per channel): the m
            # supporting states y0 + dt*f + g[:, :, k]*sqrt_dt as ONE batch of m*B rows through the user's g, then
            # corr = sum_{k,l} (g_l(Y_k) - g_l(y0)) I_kl / sqrt_dt in one kernel that streams that result once
            # (csrc/milstein_general.hip) -- no autograd in the step.
            B, d, m = g.shape
            support = K.milstein_gf_general_support(y0, f, g, dt, st.sqrt_dt, self.ito)
            g_support = sde.g(t0, support.reshape(m * B, d)).reshape(m, B, d, m)
            correction = K.milstein_gf_general_correction(g, g_support, integrals, st.sqrt_dt)
            y1 = K.step_general(y0, f, g, dt, 1.0, NoiseSpec.external(W))
            return K.lincomb2(y1, correction, 1.0, 1.0, out=out)
        # The m directional derivatives: the reference's batched formulation (base_sde.py:186-209, one JVP over an
        # m-times replicated batch) is 2.5x faster on this GPU than m separate JVPs (tools/bench_levy_jvp.py); it is
        # used for forward-only solves while the replicated diffusion stays under 4 GiB.
        m = g.shape[-1]
        batched = not torch.is_grad_enabled() and g.numel() * m * g.element_size() < (4 << 30)
        term = sde.dg_ga_jvp_column_sum_v2 if batched else sde.dg_ga_jvp_column_sum
        correction = term(t0, y0, integrals)
        y1 = K.step_general(y0, f, g, dt, 1.0, NoiseSpec.external(W))
        return K.lincomb2(y1, correction, 1.0, 1.0, out=out)

    def _advance(self, y0, st, out):
        sde, dt, noise = self.sde, st.dt, st.noise
        t0 = st.times[0]
        kind = sde.noise_type
        if kind == NOISE_TYPES.general:
            return self._advance_general(y0, st, out)
        if kind == NOISE_TYPES.additive:
            # gdg = 0 (base_sde.py:157-158): y1 = y0 + f*dt + g_prod + 0.
            if sde.user_g_prod:
                W, _ = noise.materialise()
                return K.step_prod(y0, sde.f(t0, y0), sde.g_prod(t0, y0, W), dt, 1.0, out=out)
            return K.step_general(y0, sde.f(t0, y0), sde.g(t0, y0), dt, 1.0, noise, out=out)
        scalar = kind == NOISE_TYPES.scalar
        if scalar:
            noise = self._row_noise(noise, y0.shape[1])
        if self.options[METHOD_OPTIONS.grad_free]:
            sqrt_dt = st.sqrt_dt
            f, g = sde.f_and_g(t0, y0)
            g_ = g.squeeze(2) if g.dim() == 3 else g
            y_prime = K.milstein_gf_prime(y0, f, g_, dt, sqrt_dt, self.ito)
            g_prime = sde.g(t0, y_prime)
            g_prime = g_prime.squeeze(2) if g_prime.dim() == 3 else g_prime
            return K.milstein_gf_diag(y0, f, g_, g_prime, dt, sqrt_dt, self.ito, noise, out=out)
        # derivative form: the VJP (dg/dy)^T (g * v/2) stays in autograd, everything else is fused
        if not scalar and not torch.is_grad_enabled():
            # forward-only solve: g * v/2 in ONE kernel (W regenerated in registers), no v tensor in HBM
            def v2(g):
                return K.milstein_weight(g, noise, dt, self.ito, 0.5)
        else:
            v2, _ = K.milstein_v(noise if not scalar else NoiseSpec.external(noise.W), dt, self.ito, 0.5, like=y0)
            if scalar:
                v2 = v2.reshape(-1, 1)
        f = sde.f(t0, y0)
        g, gdg = sde._g_and_gdg(t0, y0, v2, scalar_like=scalar)
        g_ = g.squeeze(2) if g.dim() == 3 else g
        return K.milstein_diag(y0, f, g_, gdg, dt, noise, out=out)


class MilsteinIto(_Milstein):
    sde_type = SDE_TYPES.ito
    ito = True


class MilsteinStratonovich(_Milstein):
    sde_type = SDE_TYPES.stratonovich
    ito = False


class SRK(BaseSDESolver):
    """Roessler's strong-order-1.5 SRI (diagonal/scalar, tableau SRID2) scheme (reference: methods/srk.py:30-88).

    The reference re-evaluates f and g of every earlier stage inside its double loop (10 f + 6 g + 4 g_prod
    per step); identical values are obtained here with 3 f and 4 g evaluations and four stage kernels that hand
    partial sums to each other (23 streams per step).
    """
    strong_order = 1.5
    weak_order = 1.5
    sde_type = SDE_TYPES.ito
    noise_types = (NOISE_TYPES.additive, NOISE_TYPES.diagonal, NOISE_TYPES.scalar)
    levy_area_approximations = (LEVY_AREA_APPROXIMATIONS.space_time, LEVY_AREA_APPROXIMATIONS.davie,
                                LEVY_AREA_APPROXIMATIONS.foster)
    needs_U = True
    stage_fracs = (0, 0.25, 0.5, 0.75, 1)   # t0, +dt/4, +dt/2, +3dt/4 (SRA1), +dt

    def __init__(self, sde, **kwargs):
        from . import adjoint
        if isinstance(sde, adjoint.AdjointSDE):
            raise ValueError("Stochastic Runge–Kutta methods cannot be used for adjoint SDEs, because it requires "
                             "direct access to the diffusion, whilst adjoint SDEs rely on a more efficient "
                             "diffusion-vector product. Use a different method instead.")
        super().__init__(sde=sde, **kwargs)

    def _trajectory_code(self):
        return _native.TRAJ_SRK if self._diag() else None

    def _program_code(self):
        return _native.TRAJ_SRK

    def _additive_code(self):
        return _native.TRAJ_SRK

    def _neural_code(self):
        # (SRID2: diagonal and scalar noise; additive noise takes SRA1 through `_additive_code`)
        return _native.TRAJ_SRK if self.sde.noise_type in (NOISE_TYPES.diagonal, NOISE_TYPES.scalar) else None

    def _advance(self, y0, st, out):
        if self.sde.noise_type == NOISE_TYPES.additive:
            return self._advance_additive(y0, st, out)
        sde, dt, noise = self.sde, st.dt, st.noise
        t_0, t_q, t_h, _t_3q, t_1, _t_end = st.times
        rdt, sqrt_dt = st.rdt, st.sqrt_dt
        if sde.noise_type == NOISE_TYPES.scalar:
            W, U = noise.materialise(need_U=True)
            noise = NoiseSpec.external(W.reshape(-1), U.reshape(-1), bcast_d=y0.shape[1])

        def g_of(t, y):
            g = sde.g(t, y)
            return g.squeeze(2) if g.dim() == 3 else g

        # C0 = (0, 1, 1/2, 0) for f, C1 = (0, 1/4, 1, 1/4) for g   (srid2.py:21-22)
        f0, g0 = sde.f(t_0, y0), g_of(t_0, y0)
        H0_1, H1_1, H1_2 = K.srk_diag_stage(1, (y0, f0, g0), dt, rdt, sqrt_dt, noise)
        f1, g1 = sde.f(t_1, H0_1), g_of(t_q, H1_1)
        H0_2, acc, P1_3 = K.srk_diag_stage(2, (y0, f0, g0, f1, g1), dt, rdt, sqrt_dt, noise)
        f2, g2 = sde.f(t_h, H0_2), g_of(t_1, H1_2)
        H1_3, acc = K.srk_diag_stage(3, (P1_3, acc, f2, g2), dt, rdt, sqrt_dt, noise)
        g3 = g_of(t_q, H1_3)
        (y1,) = K.srk_diag_stage(4, (acc, g3), dt, rdt, sqrt_dt, noise, out_last=out)
        return y1

    def _advance_additive(self, y0, st, out):
        """SRA1 (srk.py:90-111, tableaus/sra1.py): three weighted contractions g(t, y0) . w(W, U).
        C0 = (0, 3/4), C1 = (1, 0), A0[1][0] = 3/4, B0[1][0] = 3/2, alpha = (1/3, 2/3), beta1 = (1, 0),
        beta2 = (-1, 1). The diffusion always comes from `g` (a user `g_prod` computes the same product)."""
        sde, dt, noise = self.sde, st.dt, st.noise
        t_0, _t_q, _t_h, t_3q, t_1, _t_end = st.times
        rdt = st.rdt
        f0 = sde.f(t_0, y0)
        g_a = sde.g(t_1, y0)     # t0 + C1[0]*dt
        H0_1 = K.step_general_weighted(y0, f0, g_a, 3 / 4, dt, 1.0, 1, 0.0, 3 / 2, rdt, noise)
        acc = K.step_general_weighted(y0, f0, g_a, 1 / 3, dt, 1.0, 2, 1.0, -1.0, rdt, noise)
        f1 = sde.f(t_3q, H0_1)
        g_b = sde.g(t_0, y0)     # t0 + C1[1]*dt
        return K.step_general_weighted(acc, f1, g_b, 2 / 3, dt, 1.0, 2, 0.0, 1.0, rdt, noise, out=out)


class _TwoStageStratonovich(BaseSDESolver):
    """Shared body of Heun (heun.py:24-48) and Euler-Heun (euler_heun.py:19-42): a predictor with the step kernel,
    then the fused corrector `tsde_heun_final`."""
    weak_order = 1.0
    sde_type = SDE_TYPES.stratonovich
    noise_types = NOISE_TYPES.all()
    levy_area_approximations = LEVY_AREA_APPROXIMATIONS.all()
    mode = 0   # 0 = Heun, 1 = Euler-Heun

    def __init__(self, sde, **kwargs):
        self.strong_order = 0.5 if sde.noise_type == NOISE_TYPES.general else 1.0
        super().__init__(sde=sde, **kwargs)

    def _trajectory_code(self):
        return self._program_code() if self._diag() else None

    def _program_code(self):
        return _native.TRAJ_HEUN if self.mode == 0 else _native.TRAJ_EULER_HEUN

    def _deep_code(self):
        return self._program_code()

    def _advance(self, y0, st, out):
        sde, dt, noise = self.sde, st.dt, st.noise
        t0, t1 = st.times[0], st.times[-1]
        heun = self.mode == 0
        cf = dt if heun else 0.0             # predictor: y0 + dt*f + g.dW (Heun) or y0 + g.dW (Euler-Heun)
        if sde.user_product or not self._diag():
            # products are formed by the user / the contraction kernel; the corrector combines them elementwise
            if sde.user_product:
                W, _ = noise.materialise()
                f, gp = sde.f_and_g_prod(t0, y0, W)
                y_prime = K.step_prod(y0, f, gp, cf, 1.0)
                if heun:
                    f_prime, gp_prime = sde.f_and_g_prod(t1, y_prime, W)
                else:
                    f_prime, gp_prime = None, sde.g_prod(t1, y_prime, W)
            else:
                f, g = sde.f_and_g(t0, y0)
                zero = torch.zeros_like(y0)
                gp = K.step_general(zero, zero, g, 0.0, 1.0, noise)
                y_prime = K.step_prod(y0, f, gp, cf, 1.0)
                if heun:
                    f_prime, g_prime = sde.f_and_g(t1, y_prime)
                else:
                    f_prime, g_prime = None, sde.g(t1, y_prime)
                gp_prime = K.step_general(zero, zero, g_prime, 0.0, 1.0, noise)
            return K.heun_final(y0, f, f_prime, gp, gp_prime, dt, self.mode, None, prod=True, out=out)
        f, g = sde.f_and_g(t0, y0)
        y_prime = K.step_diag(y0, f, g, cf, 1.0, noise)
        if heun:
            f_prime, g_prime = sde.f_and_g(t1, y_prime)
        else:
            f_prime, g_prime = None, sde.g(t1, y_prime)
        return K.heun_final(y0, f, f_prime, g, g_prime, dt, self.mode, noise, out=out)


class Heun(_TwoStageStratonovich):
    mode = 0


class EulerHeun(_TwoStageStratonovich):
    mode = 1


class LogODEMidpoint(BaseSDESolver):
    """Log-ODE / midpoint scheme with Levy area (reference: methods/log_ode.py:25-56)."""
    weak_order = 1.0
    sde_type = SDE_TYPES.stratonovich
    noise_types = NOISE_TYPES.all()
    levy_area_approximations = (LEVY_AREA_APPROXIMATIONS.davie, LEVY_AREA_APPROXIMATIONS.foster)
    stage_fracs = (0, 0.5)
    merges_half_steps = False

    def __init__(self, sde, **kwargs):
        from . import adjoint
        if isinstance(sde, adjoint.AdjointSDE):
            raise ValueError("Log-ODE schemes cannot be used for adjoint SDEs, because they require "
                             "direct access to the diffusion, whilst adjoint SDEs rely on a more efficient "
                             "diffusion-vector product. Use a different method instead.")
        self.strong_order = 0.5 if sde.noise_type == NOISE_TYPES.general else 1.0
        super().__init__(sde=sde, **kwargs)

    def _noise_for(self, ta, tb, t0_tensor, t1_tensor, cell=None):
        # the Levy area is a (B, m, m) tensor for user autograd code: this method always materialises (W, A)
        bm = self._native_bm()
        if bm is not None:
            W, _, A = bm.increment_with_levy_area(ta, tb)
        else:
            W, A = self.bm(t0_tensor, t1_tensor, return_A=True)
        spec = NoiseSpec.external(W)
        self._levy_area = A
        return spec

    def _advance(self, y0, st, out):
        sde, dt, noise = self.sde, st.dt, st.noise
        A = self._levy_area
        half_dt = st.half_dt
        t0, t_prime = st.times[0], st.times[1]
        y_prime = self._drift_diffusion_update(t0, y0, half_dt, 0.5, noise, None)
        dg_ga = sde.dg_ga_jvp_column_sum(t_prime, y_prime, A)
        if not torch.is_tensor(dg_ga):   # 0. for non-general noise (base_sde.py:73-77, :208-209)
            dg_ga = None
        return self._final(y0, y_prime, t_prime, dt, noise, dg_ga, out)

    def _final(self, y0, y_prime, t_prime, dt, noise, dg_ga, out):
        """y1 = ((y0 + dt*f') + g'.dW) + dg_ga', f' and g' evaluated at (t', y')."""
        sde = self.sde
        if sde.user_product:
            W, _ = noise.materialise()
            f_p, gp_p = sde.f_and_g_prod(t_prime, y_prime, W)
            y1 = K.step_prod(y0, f_p, gp_p, dt, 1.0, out=out if dg_ga is None else None)
        else:
            f_p, g_p = sde.f_and_g(t_prime, y_prime)
            upd = K.step_diag if self._diag() else K.step_general
            y1 = upd(y0, f_p, g_p, dt, 1.0, noise, out=out if dg_ga is None else None)
        if dg_ga is None:
            return y1
        return K.lincomb2(y1, dg_ga, 1.0, 1.0, out=out)


def select(method, sde_type):
    """method name -> solver class (reference: methods/__init__.py:26-48)."""
    if method == METHODS.euler:
        return Euler
    if method == METHODS.milstein:
        return MilsteinIto if sde_type == SDE_TYPES.ito else MilsteinStratonovich
    if method == METHODS.srk:
        return SRK
    if method == METHODS.midpoint:
        return Midpoint
    if method == METHODS.reversible_heun:
        return ReversibleHeun
    if method == METHODS.heun:
        return Heun
    if method == METHODS.euler_heun:
        return EulerHeun
    if method == METHODS.log_ode_midpoint:
        return LogODEMidpoint
    if method == METHODS.adjoint_reversible_heun:
        raise ValueError(f"{METHODS.adjoint_reversible_heun} can only be used for adjoint_method.")
    raise ValueError(f"Method '{method}' does not match any known method.")

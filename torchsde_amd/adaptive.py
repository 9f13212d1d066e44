"""Adaptive step doubling with the control flow on the device.

The reference's adaptive branch (torchsde/_core/base_solver.py:117-142 + _core/adaptive_stepping.py:21-76) decides
accept / reject on the host after every attempted step -- one `.item()` per attempt -- and derives the next attempt's
times from that decision. Here the decision is taken by a one-thread controller kernel between the attempts
(csrc/adaptive.hip): it reads the error norm from device memory, applies the reference's PI controller, advances the
time, and writes the scalars the NEXT attempt needs into a small device table. Every kernel of an attempt reads its
scalars from that table (step size, dt/2, sqrt(dt), 1/dt through `TSDE_DEV_SCALAR` coefficients; the user's ``f`` and
``g`` get the stage times as 0-d views of it; the Brownian query kernel reads the half-step bounds from it), so an
attempt is the SAME sequence of launches with the SAME arguments every time. The OUTPUT TIMES live on the device too: the
controller walks their list, and the last launch of every attempt (`tsde_adaptive_emit`) interpolates and writes the rows of
`ys` whose time the step has just reached (interp.py:15-18). The host therefore takes no decision during a solve: it enqueues
a budget of attempts without synchronising and reads the controller's state back to learn whether the last output time has
been reached -- once per solve when the budget was right (later solves of an SDE object size it from the attempts the
previous one used), once per round otherwise. Attempts left over are inert (the controller leaves the state alone, the commit
kernel copies nothing, the emit kernel writes nothing).

Used by ``BaseSDESolver._integrate_adaptive`` whenever this package's ``BrownianInterval`` generates the path, autograd is
off and the solver keeps no state between steps; everything else takes the host-driven loop (one sync per attempt).
"""
import ctypes
import math
import warnings

import numpy as np
import torch

from . import _native
from . import kernels as K
from . import timegrid
from .kernels import NoiseSpec


class DeviceController:
    """The two device tables of one adaptive solve (`ctl`: float64 state, `scal`: scalars of the solve's dtype) and the
    launches that maintain them."""

    def __init__(self, device, dtype, stage_fracs):
        self.ctl = torch.zeros(_native.CTL_SIZE, dtype=torch.float64, device=device)
        self.scal = torch.zeros(_native.SCAL_SIZE, dtype=dtype, device=device)
        # the output times of the solve (at most OUT_CAPACITY at a time) and the address of the first output row of ys: static
        # buffers, so that a recorded attempt serves every later solve
        self.out_times = torch.zeros(self.OUT_CAPACITY, dtype=torch.float64, device=device)
        self.ys_slot = torch.zeros(1, dtype=torch.int64, device=device)
        # (t0, t1) of every accepted step, in order: what `integrate_with_grad` replays with autograd recording
        self.accept_log = torch.zeros(2 * self.LOG_CAPACITY, dtype=torch.float64, device=device)
        self.dtype, self.device = dtype, device
        self.n_fracs = len(stage_fracs)
        self._fracs = (ctypes.c_double * max(self.n_fracs, 1))(*[float(f) for f in stage_fracs])
        self._lib, self._dt_code, _ = K._launch_env(self.scal)

    OUT_CAPACITY = 1024
    LOG_CAPACITY = 8192

    def load(self, t0, t_end, step_size, dt_min):
        """The state a solve starts from: the one host->device copy of the solve."""
        host = np.zeros(_native.CTL_SIZE, dtype=np.float64)
        host[_native.CTL_CURR_T] = host[_native.CTL_PREV_T] = t0
        host[_native.CTL_STEP_SIZE] = step_size
        host[_native.CTL_PREV_ERROR_RATIO] = np.nan
        host[_native.CTL_T_END] = t_end
        host[_native.CTL_DT_MIN] = dt_min
        self.ctl.copy_(torch.from_numpy(host))

    def _stream(self):
        return K._launch_env(self.scal)[2]

    def begin(self, out_times, ys_rows):
        """Aim at `out_times` (host float64 array, ascending, at most OUT_CAPACITY) whose rows are `ys_rows` (n_out, ...)."""
        n_out = len(out_times)
        self.out_times[:n_out].copy_(torch.from_numpy(np.array(out_times, dtype=np.float64)))      # (a writable copy)
        self.ys_slot.copy_(torch.tensor([ys_rows.data_ptr()], dtype=torch.int64))
        code = self._lib.tsde_adaptive_begin_outputs(self.ctl.data_ptr(), self.scal.data_ptr(), self.out_times.data_ptr(),
                                                     n_out, self._fracs, self.n_fracs, self._dt_code, self._stream())
        _native.check(code, "tsde_adaptive_begin_outputs")

    def control(self, error):
        code = self._lib.tsde_adaptive_control_outputs(self.ctl.data_ptr(), self.scal.data_ptr(), error.data_ptr(),
                                                       self.out_times.data_ptr(), self.accept_log.data_ptr(),
                                                       self.LOG_CAPACITY, self._fracs, self.n_fracs, self._dt_code,
                                                       self._stream())
        _native.check(code, "tsde_adaptive_control_outputs")

    def emit(self, prev_y, curr_y):
        """Write the output rows the controller has marked (none: the kernel returns at once)."""
        code = self._lib.tsde_adaptive_emit(self.ys_slot.data_ptr(), prev_y.data_ptr(), curr_y.data_ptr(), curr_y.numel(),
                                            self.ctl.data_ptr(), self.out_times.data_ptr(), self._dt_code, self._stream())
        _native.check(code, "tsde_adaptive_emit")

    def commit(self, prev_y, curr_y, y_next):
        code = self._lib.tsde_adaptive_commit(prev_y.data_ptr(), curr_y.data_ptr(), y_next.data_ptr(), curr_y.numel(),
                                              self.scal.data_ptr(), self._dt_code, self._stream())
        _native.check(code, "tsde_adaptive_commit")

    def merge_halves(self, W, U, Wa, Ha, Wb, Hb):
        K.merge_halves(W, U, Wa, Ha, Wb, Hb, ctl=self.ctl)

    def bounds_ptr(self, half):
        offset = _native.CTL_BOUNDS_A if half == 0 else _native.CTL_BOUNDS_B
        return self.ctl.data_ptr() + 8 * offset

    def scalar(self, sub, which):
        """TSDE_DEV_SCALAR of entry `which` (SUB_DT, ...) of sub-step `sub` (0 whole, 1 first half, 2 second half)."""
        return _native.dev_scalar(self.scal, sub * _native.SUB_STRIDE + which)

    def times(self, sub):
        """The stage times of sub-step `sub` as 0-d views of the table (what the user's f(t, y), g(t, y) receive)."""
        base = sub * _native.SUB_STRIDE + _native.SUB_TIMES
        return tuple(self.scal[base + j] for j in range(self.n_fracs + 1))

    def read(self):
        """The controller's state on the host: the one synchronisation per round of attempts."""
        return self.ctl.cpu().numpy()


last_stats = None     # of the most recent device-controlled solve (tests, tools/bench_adaptive.py)
_last_controller = None   # ... and its DeviceController (the accepted-step log `integrate_with_grad` reads)


def query_dev(bm, bounds_ptr, out_W, out_U, out_H):
    """`BrownianInterval.increment` with the interval read from device memory (tsde_brownian_query_dev)."""
    lib = _native.load()
    edges = bm._device_edges()
    code = lib.tsde_brownian_query_dev(
        _native.ptr(out_W), _native.ptr(out_U), _native.ptr(out_H), bm._numel, bm._key, bm._elem0, _native.ptr(edges),
        bm._edges.size - 1, bounds_ptr, 1 if bm._have_H else 0, bm._max_depth,
        None if bm._entropy_dev is None else bm._entropy_dev.data_ptr(), _native.dtype_code(bm._dtype),
        _native.stream_ptr(bm._device))
    _native.check(code, "tsde_brownian_query_dev")


def _no_gradient_can_flow(solver, y0, ts):
    """The device-controlled loop runs under no_grad, so it may only take solves that autograd would not record anyway.
    `_tracks_grad` sees y0 and the module's parameters; a trainable tensor that is neither (a plain attribute with
    requires_grad=True, a leaf captured in a closure) only shows in what f and g return, so with grad mode on they
    are probed once (the reference and the host loop propagate such gradients: base_solver.py:117-142)."""
    if not torch.is_grad_enabled():
        return True
    if solver._tracks_grad(y0):
        return False
    # one probe per SDE object and state, not one per solve: the probe calls the user's code, which costs time and
    # triggers its side effects (call counters) once more than the reference does
    from . import graph as graph_module
    _, base = graph_module._wrapper_chain(solver.sde)
    key = ("grad-probe", graph_module.python_state(base), tuple(y0.shape), y0.dtype)
    try:
        verdicts = base.__dict__.setdefault("_tsde_grad_probe", {})
    except AttributeError:
        verdicts = {}
    if key[1] is not None and key in verdicts:
        return verdicts[key]
    verdict = _probe_no_gradient(solver, y0, ts)
    if key[1] is not None:
        if len(verdicts) >= 8:
            verdicts.clear()
        verdicts[key] = verdict
    return verdict


def _probe_no_gradient(solver, y0, ts):
    try:
        probes = solver.sde.f_and_g_prod(ts[0], y0, torch.zeros(solver.bm.shape, dtype=y0.dtype, device=y0.device)) \
            if solver.sde.user_product else solver.sde.f_and_g(ts[0], y0)
    except Exception:     # a provider the probe cannot call: leave the solve to the host loop
        return False
    return not any(torch.is_tensor(p) and p.requires_grad for p in probes)


def controllable(solver, y0, ts):
    """Everything device-side control needs, gradients aside."""
    bm = solver._native_bm()
    return (bm is not None and solver.options.get("device_adaptive", True)
            and bm._snap == 0 and bm._tol == 0. and bm._rootW is None and bm._rootH is None
            and not solver.stateful and solver.merges_half_steps and not solver.options.get("general_noise", False)
            and y0.dtype == ts.dtype == bm.dtype and y0.dtype in (torch.float32, torch.float64)
            and len(solver.stage_fracs) <= _native.ADAPTIVE_MAX_STAGES - 1 and y0.numel() > 0)


def usable(solver, y0, ts):
    """Can this solve run with device-side control and nothing else? (With gradients: `integrate_with_grad`; otherwise the
    host-driven loop, one sync per attempt.)"""
    return controllable(solver, y0, ts) and _no_gradient_can_flow(solver, y0, ts)


class _Attempt:
    """The buffers of a device-controlled adaptive solve and the launch sequence of ONE attempted step: two generator
    queries (the halves), the whole step merged from them, three steps, the error norm, the decision, the commit. Every
    launch reads what changes between attempts from the controller's tables, so the sequence is the same every time."""

    def __init__(self, solver, y0, step_cls):
        bm = solver._native_bm()
        device, dtype = y0.device, y0.dtype
        self.ctrl = ctrl = DeviceController(device, dtype, solver.stage_fracs)
        self.curr_y, self.prev_y = (torch.empty_like(y0, memory_format=torch.contiguous_format) for _ in range(2))
        self.y_full, self.y_mid, self.y_next = (torch.empty_like(self.curr_y) for _ in range(3))
        want_U = solver.needs_U and bm._have_H
        shape = tuple(bm.shape)

        def buf():
            return torch.empty(shape, dtype=bm.dtype, device=device)
        self.W, self.Wa, self.Wb = buf(), buf(), buf()
        self.U, self.Ua, self.Ub, self.Ha, self.Hb = (buf(), buf(), buf(), buf(), buf()) if want_U else (None,) * 5
        noises = (NoiseSpec.external(self.W, self.U), NoiseSpec.external(self.Wa, self.Ua),
                  NoiseSpec.external(self.Wb, self.Ub))
        self.steps = [step_cls(ctrl.times(s), ctrl.scalar(s, _native.SUB_DT), noises[s], None, None,
                               half_dt=ctrl.scalar(s, _native.SUB_HALF_DT), sqrt_dt=ctrl.scalar(s, _native.SUB_SQRT_DT),
                               rdt=ctrl.scalar(s, _native.SUB_RDT)) for s in range(3)]
        self.w0 = _native.dev_scalar(ctrl.scal, _native.SCAL_W0)
        self.w1 = _native.dev_scalar(ctrl.scal, _native.SCAL_W1)
        self.rtol, self.atol = solver.rtol, solver.atol
        self.norm_scratch = K.error_norm_scratch(device)

    def load(self, y0, t0, t_end, step_size, dt_min, bm):
        self.ctrl.load(t0, t_end, step_size, dt_min)
        self.curr_y.copy_(y0)
        self.prev_y.copy_(y0)

    def _launch(self, solver, bm):
        ctrl = self.ctrl

        def advance(y, st, out):
            res = solver._advance(y, st, out)
            if res.data_ptr() != out.data_ptr():      # a path that allocates its own result
                out.copy_(res)

        query_dev(bm, ctrl.bounds_ptr(0), self.Wa, self.Ua, self.Ha)
        query_dev(bm, ctrl.bounds_ptr(1), self.Wb, self.Ub, self.Hb)
        ctrl.merge_halves(self.W, self.U, self.Wa, self.Ha, self.Wb, self.Hb)
        advance(self.curr_y, self.steps[0], self.y_full)
        advance(self.curr_y, self.steps[1], self.y_mid)
        advance(self.y_mid, self.steps[2], self.y_next)
        ctrl.control(K.error_norm(self.y_full, self.y_next, self.rtol, self.atol, scratch=self.norm_scratch))
        ctrl.commit(self.prev_y, self.curr_y, self.y_next)
        ctrl.emit(self.prev_y, self.curr_y)

    def run(self, solver, bm, n):
        for _ in range(n):
            self._launch(solver, bm)


class _GraphedAttempt(_Attempt):
    """``options={"hip_graph": True}``: the attempt as a HIP graph, cached on the user's SDE object like the graphs of
    fixed-step solves (graph.py) and replayed by every later solve of the same structure. An attempt is ~30 short
    launches; capturing them costs 30-45 ms, so this only pays across solves -- then an attempt costs one launch of
    the host. What a new solve brings is loaded into the static buffers: the controller's state, y0, and the
    Brownian motion's entropy (the generator kernels read it from one device word)."""

    verified = True

    def __init__(self, solver, y0, step_cls, verify=False):
        super().__init__(solver, y0, step_cls)
        bm = solver._native_bm()
        device = y0.device
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=device)
        self._keepalive = bm            # the captured kernels read this generator's device copy of its cell edges
        self._set_seed(bm)
        bm._entropy_dev = self.seed_dev
        try:
            # any consistent state will do for the warm-up and the capture: the launches read it from the tables
            self.scratch_ys = torch.empty((1,) + tuple(y0.shape), dtype=y0.dtype, device=device)
            self.load(y0, float(bm._t0), float(bm._t1), float(bm._t1 - bm._t0), 0.0, bm)
            self.ctrl.begin([float(bm._t1)], self.scratch_ys)
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):           # warm-up outside capture (lazy inits, allocator)
                self._launch(solver, bm)
            torch.cuda.current_stream(device).wait_stream(side)
            from . import graph as graph_module
            self.graph = graph_module.new_graph()
            with graph_module._capturing(self.graph, device):
                self._launch(solver, bm)
            if verify:
                # "auto": the recorded attempt is trusted only if, from the same loaded state, it gives what the eager
                # attempt gives (state, controller tables) and keeps giving it with eager work in between
                # (graph.replays_are_stable: the second line of defence behind the memset-node rewrite)
                def fresh():
                    self.scratch_ys.zero_()
                    self.load(y0, float(bm._t0), float(bm._t1), float(bm._t1 - bm._t0), 0.0, bm)
                    self.ctrl.begin([float(bm._t1)], self.scratch_ys)

                def outputs():
                    return [self.curr_y, self.prev_y, self.y_next, self.ctrl.ctl, self.ctrl.scal, self.scratch_ys]

                def replay():
                    fresh()
                    self.graph.replay()

                def eagerly():
                    fresh()
                    self._launch(solver, bm)
                eagerly()
                want = [o.clone() for o in outputs()]
                replay()
                self.verified = (graph_module._same_tensors(outputs(), want, exact=True)
                                 and graph_module.replays_are_stable(replay, outputs, eagerly))
        finally:
            bm._entropy_dev = None
        self.steps = None               # (they reference the solver's SDE: no cycle through the cache on that object)

    def _set_seed(self, bm):
        key = bm._key
        self.seed_dev.fill_(key - (1 << 64) if key >= (1 << 63) else key)

    def load(self, y0, t0, t_end, step_size, dt_min, bm):
        super().load(y0, t0, t_end, step_size, dt_min, bm)
        self._set_seed(bm)

    def run(self, solver, bm, n):
        for _ in range(n):
            self.graph.replay()


def _attempt_for(solver, y0, ts_host, step_cls):
    """This solve's `_Attempt`: a fresh one, or with ``hip_graph`` the cached graph of the same structure."""
    from . import graph as graph_module
    bm = solver._native_bm()
    mode = graph_module.mode_of(solver.options)
    if mode == "auto" and not graph_module._auto_eligible(bm, y0, len(ts_host)):
        mode = False
    if not mode:
        return _Attempt(solver, y0, step_cls)
    chain, base = graph_module._wrapper_chain(solver.sde)
    params = tuple(p.data_ptr() for p in base.parameters()) if hasattr(base, "parameters") else ()
    sig = ("adaptive-attempt", type(solver).__name__, chain, getattr(solver.sde, "sde_type", None),
           getattr(solver.sde, "noise_type", None), params, tuple(y0.shape), y0.dtype, str(y0.device),
           float(solver.rtol), float(solver.atol), tuple(bm.shape), bm.levy_area_approximation, bm.row_offset,
           bm._edges.tobytes(), bm._max_depth,
           tuple(sorted((k, v) for k, v in solver.options.items() if isinstance(v, (bool, int, float, str)))))
    cache = graph_module._cache_of(base)
    if mode == "auto":
        # the drop-in default (graph.py): the first solve of a structure runs eagerly, the second records the attempt
        # and checks it against the eager one, later ones replay it; the Python-side state of the SDE object is part of
        # the key, and code that cannot be captured (it synchronises with the host, say) stays eager for good
        sig = graph_module.auto_key(cache, sig, base)
        if sig is None:
            return _Attempt(solver, y0, step_cls)
        entry = cache.get(sig)
        if entry is None:
            # this solve runs eagerly; its FIRST round of attempts is screened (graph.run_screened) and decides
            # whether the next solve of this structure may record the attempt
            attempt = _Attempt(solver, y0, step_cls)
            attempt.screen = lambda reason: graph_module._remember(
                cache, sig, graph_module._Seen() if reason is None else graph_module._Refused(reason))
            return attempt
        if isinstance(entry, graph_module._Refused):
            return _Attempt(solver, y0, step_cls)
        if isinstance(entry, graph_module._Seen):
            try:
                with graph_module._drift_then_diffusion(solver.sde):
                    entry = _GraphedAttempt(solver, y0, step_cls, verify=True)
            except Exception as e:
                cache[sig] = graph_module._Refused(f"capture failed: {type(e).__name__}: {e}")
                return _Attempt(solver, y0, step_cls)
            if not entry.verified:
                cache[sig] = graph_module._Refused("the recorded attempt did not reproduce the eager one")
                return _Attempt(solver, y0, step_cls)
            cache[sig] = entry
        return entry
    attempt = cache.get(sig)
    if attempt is None:
        attempt = _GraphedAttempt(solver, y0, step_cls)
        graph_module._remember(cache, sig, attempt)
    return attempt


# States up to this many elements: an attempt is a chain of launch-bound kernels, so an inert attempt costs little and a host
# synchronisation costs as much as several attempts -- aim the first round at the end of the solve.
_LATENCY_BOUND_ELEMENTS = 1 << 18


def round_budget(span, step_size, hint=None, cap=256):
    """Attempts to enqueue before the next look at the controller: `hint` + 1 when a previous solve of this structure said how
    many it used; else what the current step size needs to cover `span`, less one when that is more than two (accepted steps
    only ever grow, adaptive_stepping.py:35-37: an upper bound unless attempts are rejected); 0 when nothing is left to cover;
    never more than `cap`."""
    need = int(math.ceil(max(span, 0.0) / step_size))
    if need == 0:
        return 0
    budget = hint + 1 if hint is not None else (need - 1 if need > 2 else need)
    return min(max(1, budget), cap)


def _hints_of(solver, y0, ts_host):
    """({key: attempts the last such solve used}, key of this solve) kept on the user's SDE object, or (None, None)."""
    from . import graph as graph_module
    _, base = graph_module._wrapper_chain(solver.sde)
    try:
        hints = base.__dict__.setdefault("_tsde_adaptive_hints", {})
    except AttributeError:
        return None, None
    dt = solver.dt if not torch.is_tensor(solver.dt) else float(solver.dt)
    key = (type(solver).__name__, tuple(y0.shape), y0.dtype, float(solver.rtol), float(solver.atol), float(dt),
           float(solver.dt_min), ts_host.tobytes())
    return hints, key


def integrate_with_grad(solver, y0, ts, extra0, step_cls):
    """The adaptive solve when autograd is recording. The reference (base_solver.py:117-142) records EVERY attempt -- the
    full step, both half steps, rejected ones included -- and synchronises after each; only the half steps of the accepted
    attempts reach the result. Here the device-controlled loop runs first, under no_grad, and logs the accepted steps; then
    exactly those are run again with autograd recording (two half steps each, the same kernels on the same increments, so the
    same values), and the output rows are interpolated from them. Same gradients, no synchronisation per attempt, a third of
    the autograd graph. None when the log overflowed (the caller takes the host-driven loop)."""
    global last_stats
    with torch.no_grad():
        values, _ = integrate(solver, y0.detach(), ts, extra0, step_cls)
    stats = dict(last_stats)
    accepted = stats["accepted"]
    ctrl = _last_controller
    if accepted > ctrl.LOG_CAPACITY:
        return None
    log = ctrl.accept_log[:2 * accepted].cpu().numpy().reshape(-1, 2)
    ts_host = timegrid.ts_to_host(ts)
    np_dtype = ts_host.dtype.type
    prev_t = curr_t = ts_host[0]
    prev_y = curr_y = y0
    extra = tuple(extra0) if extra0 is not None else ()
    ys, i_out, T = [y0], 1, len(ts_host)
    for t0, t1 in log:
        t0, t1 = np_dtype(t0), np_dtype(t1)
        mid = np_dtype(0.5) * (t0 + t1)
        _, n_a, n_b = solver._step_doubling_noise(float(t0), float(mid), float(t1))
        host = solver._stage_times_host(t0, mid) + solver._stage_times_host(mid, t1)
        dev_times = torch.tensor(np.asarray(host, dtype=ts_host.dtype), device=y0.device).to(ts.dtype).unbind(0)
        k = len(host) // 2
        y_mid, mid_extra = solver.step(t0, mid, curr_y, extra, noise=n_a, times=dev_times[:k])
        y_next, extra = solver.step(mid, t1, y_mid, mid_extra, noise=n_b, times=dev_times[k:])
        prev_t, prev_y, curr_t, curr_y = t0, curr_y, t1, y_next
        while i_out < T and not curr_t < ts_host[i_out]:
            out_t = ts_host[i_out]
            w0 = (curr_t - out_t) / (curr_t - prev_t)
            w1 = (out_t - prev_t) / (curr_t - prev_t)
            ys.append(K.linear_interp(prev_y, curr_y, float(w0), float(w1)))
            i_out += 1
    if i_out != T:
        return None                     # (cannot happen: the device loop reached every output time with these steps)
    last_stats = dict(stats, control="device, accepted steps replayed with autograd", replayed_steps=int(accepted))
    return torch.stack(ys, dim=0), extra


def integrate(solver, y0, ts, extra0, step_cls):
    """The adaptive solve of `solver` (see the module docstring). Returns (ys, extra solver state)."""
    bm = solver._native_bm()
    device, dtype = y0.device, y0.dtype
    ts_host = timegrid.ts_to_host(ts)
    np_dtype = ts_host.dtype.type
    step_size = solver.dt if not torch.is_tensor(solver.dt) else float(solver.dt)
    solver._state_dtype = dtype
    solver._extra = tuple(extra0) if extra0 is not None else ()
    bm.locate(float(ts_host[0]), float(ts_host[-1]))        # freezes a generator that never saw a grid
    bm._device_edges()
    T = len(ts_host)
    ys = torch.empty((T,) + tuple(y0.shape), dtype=dtype, device=device)
    ys[0].copy_(y0)

    curr_t, dt_min_hits, syncs, attempts, state = float(ts_host[0]), 0.0, 0, 0, None
    t_end = float(ts_host[-1])
    # How many attempts the previous solve of this structure on this SDE object used: the budget of this solve's first round.
    # (Paths differ, so it is a guess; a shortfall costs one more round, a surplus a few inert attempts.)
    hints, hint_key = _hints_of(solver, y0, ts_host)
    hint = hints.get(hint_key) if hints is not None else None
    small = y0.numel() <= _LATENCY_BOUND_ELEMENTS
    with torch.no_grad():
        attempt = _attempt_for(solver, y0, ts_host, step_cls)
        attempt.load(y0.detach(), float(ts_host[0]), t_end, float(step_size), float(solver.dt_min), bm)
        ctrl = attempt.ctrl
        for lo in range(1, T, ctrl.OUT_CAPACITY):          # (more output times than the device list holds: in chunks)
            outs = np.asarray(ts_host[lo:lo + ctrl.OUT_CAPACITY], dtype=np.float64)
            ctrl.begin(outs, ys[lo:lo + len(outs)])
            ctrl.emit(attempt.prev_y, attempt.curr_y)       # output times the state already meets (ts[k] == ts[0], ...)
            reached, first_round = 0, True
            while True:
                # The attempts the current step size needs, less one when that is more than two: accepted steps only ever
                # grow (adaptive_stepping.py:35-37), so the estimate is an upper bound unless attempts are rejected. An
                # attempt enqueued after the last output time is reached is inert but still costs its kernels, a shortfall
                # costs one more round -- one more synchronisation:
                #   * with a hint, the first round is the hint;
                #   * a small state (latency-bound attempts) aims at the LAST output time at once;
                #   * a large one (attempts of hundreds of microseconds) at the next output time only, like a host loop.
                target = float(outs[-1]) if (small or hint is not None) else float(outs[min(reached, len(outs) - 1)])
                budget = round_budget(target - curr_t, max(step_size, solver.dt_min),
                                      hint if (first_round and lo == 1) else None)
                first_round = False
                screen = getattr(attempt, "screen", None)
                if screen is not None:      # (hip_graph="auto", first solve of this structure)
                    from . import graph as graph_module
                    _, reason = graph_module.run_screened(lambda: attempt.run(solver, bm, budget))
                    screen(reason)
                    attempt.screen = None
                else:
                    attempt.run(solver, bm, budget)
                attempts += budget
                state = ctrl.read()
                syncs += 1
                if state[_native.CTL_NAN_SEEN] != 0.0:
                    raise AssertionError("Found nans in the error estimate. Try increasing the tolerance or "
                                         "regularizing the dynamics.")
                if state[_native.CTL_DT_MIN_HITS] > dt_min_hits:
                    dt_min_hits = state[_native.CTL_DT_MIN_HITS]
                    warnings.warn("Hitting minimum allowed step size in adaptive time-stepping.")
                curr_t, step_size = float(state[_native.CTL_CURR_T]), float(state[_native.CTL_STEP_SIZE])
                reached = int(state[_native.CTL_OUT_IDX])
                if reached >= len(outs):
                    break
    global last_stats
    final = ctrl.read() if state is None else state
    if hints is not None:
        if len(hints) >= 16:
            hints.clear()
        hints[hint_key] = int(final[_native.CTL_ATTEMPTS])
    last_stats = {"control": "device", "host_syncs": syncs, "output_times": T - 1, "attempts_enqueued": attempts,
                  "attempts_used": int(final[_native.CTL_ATTEMPTS]), "accepted": int(final[_native.CTL_ACCEPTED]),
                  "dtype": np_dtype.__name__, "launch": "graph replay" if hasattr(attempt, "graph") else "eager"}
    global _last_controller
    _last_controller = ctrl
    return ys, solver._extra

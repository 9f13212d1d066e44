"""When a reversible-Heun solve takes the matrix-core kernels of neural_rheun.py.

An UNCHANGED user module whose drift and diffusion are perceptrons of (t, y) -- the reference's `Neural*` problems
(tests/problems.py:135-252) and the generator of examples/sde_gan.py:77-101 -- solved with ``method="reversible_heun"``:
``sdeint`` (with or without autograd recording) and ``sdeint_adjoint(..., adjoint_method="adjoint_reversible_heun")``, the
pair the reference recommends for training (DOCUMENTATION.md:97,118). The module is interpreted at every solve
(recognise.py, `deep_spec`), so the weights are this solve's live values and the kernels' gradients land on the user's own
tensors. Trust is earned as on the other recognised routes (solvers._integrate_recognised): the first solve of a (form, batch
size, route) on an SDE object runs BOTH ways and returns the stepwise result; values are compared elementwise and -- when a
gradient can flow -- the gradients with respect to y0 and every parameter for one random cotangent
(solvers._both_routes_agree). ``options={"trajectory_kernel": False}`` opts out; ``TSDE_VERIFY_EVERY`` re-verifies.
"""
import numpy as np
import torch

from . import kernels as K
from . import neural_rheun
from . import timegrid
from .brownian import BrownianInterval
from .settings import NOISE_TYPES


class Route:
    """One solve's plan on the kernels: `solve(y0)` launches it; `record(fast, stepwise, y0)` files the verdict of a verifying
    solve."""

    def __init__(self, solver, spec, schedule, times_host, book, key, trusted, reverify):
        self.solver, self.spec, self.schedule, self.times_host = solver, spec, schedule, times_host
        self.book, self.key, self.trusted, self.reverify = book, key, trusted, reverify

    def solve(self, y0, z_holder=None):
        _, drift, diffusion, noise, m = self.spec
        return neural_rheun.solve(y0, drift, diffusion, noise, m, self.schedule, self.times_host, self.solver._native_bm(),
                                  z_holder)

    def parameters(self):
        return self.spec[1].parameters() + self.spec[2].parameters()

    def record(self, fast, stepwise, y0, extra_inputs=()):
        verdict = self.solver._both_routes_agree(fast, stepwise, y0, "the reversible-Heun kernels", network=True,
                                                 extra_inputs=extra_inputs)
        self.solver._record_verdict(self.book, self.key, verdict, self.reverify)
        return verdict


def plan(solver, y0, ts, differentiable, need_boundaries=False, tag=()):
    """The `Route` of this solve, or None when it stays stepwise. `differentiable`: a gradient will be asked of the result (the
    interpretation then watches for stop-gradients); `need_boundaries`: every output must sit on a step boundary (the backward
    sweep of `sdeint_adjoint` steps to each of them, adjoint.py:97-112)."""
    from . import graph, recognise
    from .sde import ForwardSDE
    sde, bm = solver.sde, solver._native_bm()
    if (not recognise.ENABLED or not solver.options.get("trajectory_kernel", True) or solver.adaptive
            or type(sde) is not ForwardSDE or sde.user_product
            or sde.noise_type not in (NOISE_TYPES.diagonal, NOISE_TYPES.scalar, NOISE_TYPES.general)):
        return None
    if (not isinstance(bm, BrownianInterval) or y0.dim() != 2 or len(bm.shape) != 2 or bm.shape[0] != y0.shape[0]
            or not y0.is_cuda or y0.shape[0] < 1 or y0.dtype != torch.float32 or ts.dtype != y0.dtype or bm.dtype != y0.dtype
            or bm._rootW is not None or bm._rootH is not None or bm._snap or torch.cuda.is_current_stream_capturing()
            or y0.numel() >= 2 ** 30):
        return None
    chain, base = graph._wrapper_chain(sde)
    assume_pure = solver._assume_pure(base)
    if not solver._may_be_interpreted(base) or (not assume_pure and graph.call_counters(base)):
        return None
    try:
        book = base.__dict__.setdefault(solver._RECOGNISED_ATTR, {"refused": {}, "trusted": {}})
    except AttributeError:
        return None

    def state_of():
        return ("assumed pure",) if assume_pure else graph.python_state(base)
    who = (chain, type(solver).__name__ + (":kernels, with gradients" if differentiable else ":kernels"))
    state = None                 # (the fingerprint costs ~0.8 ms: taken only when there is a refusal to look up or to file)
    if book["refused"]:
        state = state_of()
        if state is None or (state,) + who in book["refused"]:
            return None

    def refuse(reason):
        key_state = state if state is not None else state_of()
        if key_state is not None:
            if len(book["refused"]) >= 16:
                book["refused"].clear()
            book["refused"][(key_state,) + who] = reason
        return None
    try:
        found = recognise.recognise(sde, ts[0], y0, differentiable=differentiable)
        if not found.neural:
            raise recognise.NotElementwise("drift and diffusion are not both networks of (t, y)")
        spec = found.deep_spec(sde.noise_type)
    except recognise.NotElementwise as e:
        return refuse(str(e))
    m = spec[4]
    if tuple(bm.shape) != (y0.shape[0], m):
        return None
    # the grid: steps on the generator's cells, outputs where the caller asked for them
    grid = timegrid.build(timegrid.ts_to_host(ts), solver.dt)
    if grid.n_steps == 0:
        return None
    t64 = grid.t_f64()
    bm.adopt_grid(t64)
    cells = bm.match_grid(t64)
    if cells is None:
        return None
    cells = np.asarray(cells, dtype=np.int64)
    out_step = [kc for (_, kc, _, _) in grid.outputs]
    out_w = [(w0, w1) for (_, _, w0, w1) in grid.outputs]
    if need_boundaries and any(not (w0 == 0.0 and w1 == 1.0) for (w0, w1) in out_w):
        return None
    h = bm._edges[cells + 1] - bm._edges[cells]
    np_dtype = grid.t.dtype.type
    rows = np.zeros((grid.n_steps, 8), dtype=np.float64)
    rows[:, 0] = grid.dt
    rows[:, 1] = np_dtype(0.5) * grid.dt
    rows[:, 2] = np_dtype(1) / grid.dt
    rows[:, 3] = np.sqrt(grid.dt)
    rows[:, 4] = np.sqrt(h)
    rows[:, 5] = np.sqrt(h / 12.0)
    rows[:, 6] = h
    rows[:, 7] = grid.t[:-1]
    schedule = K.TrajectorySchedule.cached(rows, cells, out_step, out_w, y0.device, y0.dtype)
    key = solver._recognised_key(found, chain, y0) + ("kernels",) + tuple(tag) + (("autograd",) if differentiable else ())
    verdict = book["trusted"].get(key)
    reverify = verdict is True and solver._due_for_reverification(book, key)
    if verdict is not None and verdict is not True and not reverify:
        return None
    trusted = verdict is True and not reverify
    if not trusted:
        # the verifying solve: a second interpretation on a probe of another height must find the same nets over the same
        # tensors, and the calls must leave the object's Python-side state and the random generators alone
        if state is None:
            state = state_of()
            if state is None:
                return None
        rng_before = solver._rng_states(y0.device)
        try:
            again = recognise.recognise(sde, ts[0], y0, differentiable=differentiable, rows=5).deep_spec(sde.noise_type)
        except recognise.NotElementwise as e:
            return refuse(str(e))
        same = again[3:] == spec[3:] and all(
            a.structure() == b.structure() and all(x is y for x, y in zip(a.parameters(), b.parameters()))
            for a, b in zip(again[1:3], spec[1:3]))
        if not same:
            solver._record_verdict(book, key, "two interpretations of the same code (probes of 2 and 5 rows) found different "
                                   "networks", reverify)
            return None
        if state_of() != state:
            return refuse("calling f and g changes the object's Python-side state")
        if any(not torch.equal(a, b) for a, b in zip(rng_before, solver._rng_states(y0.device))):
            return refuse("calling f and g advances a random number generator")
    route = Route(solver, spec, schedule, np.ascontiguousarray(grid.t, dtype=np.float32), book, key, trusted, reverify)
    route.cells, route.out_steps = cells, out_step
    return route


_GRID_MATCHES = {}


def backward_grid_matches(bm, ts_host, dt, schedule_cells, out_steps):
    """The backward solver builds its own grid on every [-ts[i], -ts[i-1]] (adjoint.py:97-112): its steps must be the forward
    cells walked backwards (cf. mlp_adjoint.route). Remembered by content: a training loop asks the same question every
    iteration, and an example with 64 output times (examples/sde_gan.py) builds 63 grids to answer it."""
    key = (ts_host.tobytes(), str(ts_host.dtype), float(dt), np.asarray(schedule_cells).tobytes(), tuple(out_steps),
           bm._edges.tobytes())
    hit = _GRID_MATCHES.get(key)
    if hit is None:
        if len(_GRID_MATCHES) >= 32:
            _GRID_MATCHES.clear()
        hit = _GRID_MATCHES[key] = _backward_grid_matches(bm, ts_host, dt, schedule_cells, out_steps)
    return hit


def _backward_grid_matches(bm, ts_host, dt, schedule_cells, out_steps):
    boundaries = [0] + list(out_steps)
    for i in range(len(ts_host) - 1, 0, -1):
        back = timegrid.build(np.array([-ts_host[i], -ts_host[i - 1]], dtype=ts_host.dtype), dt)
        k_lo, k_hi = boundaries[i - 1], boundaries[i]
        if back.n_steps != k_hi - k_lo:
            return False
        walked = bm.match_grid(-back.t_f64()[::-1])
        if walked is None or not np.array_equal(np.asarray(walked, dtype=np.int64), schedule_cells[k_lo:k_hi]):
            return False
    return True


def plan_adjoint(solver, sde, y0, ts, bm, dt, adjoint_params):
    """The `Route` of ``sdeint_adjoint(method="reversible_heun", adjoint_method="adjoint_reversible_heun")`` on the kernels, or
    None: as `plan`, and the gradients asked for must be exactly those of the two nets' tensors (a narrower or wider
    `adjoint_params` is the stepwise adjoint's business), every output on a step boundary, the backward grids the forward
    cells walked backwards."""
    route = plan(solver, y0, ts, differentiable=True, need_boundaries=True, tag=("adjoint",))
    if route is None:
        return None
    wanted = {id(p) for p in adjoint_params}
    held = {id(p) for p in route.parameters() if p.requires_grad}
    if wanted != held:
        return None
    if not backward_grid_matches(bm, timegrid.ts_to_host(ts), dt, route.cells, route.out_steps):
        return None
    return route

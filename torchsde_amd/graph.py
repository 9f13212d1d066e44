"""HIP-graph capture of a whole fixed-step solve (opt-in: ``options={"hip_graph": True}``).

A solve is thousands of short kernels (the user's ``f``/``g`` torch ops plus one fused step kernel per stage).
For small and medium batches the GPU finishes each of them faster than Python can issue the next one; the
reference is in the same regime and additionally syncs three times per step. Here the launch-only part of a
solve (``BaseSDESolver._run``) is captured ONCE into a HIP graph -- through torch's capture stream, which also
records the ``libtorchsde_amd.so`` launches because they are issued on torch's current stream -- and replayed
for every later solve with the same SDE object, shapes, method, time grid and Brownian structure.

What changes between solves is the initial state (copied into the graph's static input buffer) and the Brownian
seed: the kernels read the entropy from one device word (``tsde_noise_t.entropy_dev``) that is rewritten before
each replay, so a new ``BrownianInterval`` (new entropy) reuses the captured graph.

Constraints (checked, with a loud fallback to the eager path otherwise): forward solve without autograd
tracking; a native ``BrownianInterval``; the user's ``f``/``g`` must be capture-safe torch code (static shapes,
no host sync, no Python-side state that changes between solves).
"""
import warnings

import torch

from .brownian import BrownianInterval

_CACHE_ATTR = "_tsde_hip_graphs"


_MAX_GRAPHS_PER_SDE = 16


def _remember(cache, sig, captured):
    """Each captured graph pins its memory pool; a caller that keeps changing the structure (e.g. random `ts`) must
    not grow the cache without bound: the oldest entry goes first."""
    while len(cache) >= _MAX_GRAPHS_PER_SDE:
        cache.pop(next(iter(cache)))
    cache[sig] = captured


class _CapturedSolve:
    def __init__(self, solver, y0, ts, extra0=()):
        bm = solver.bm
        device = y0.device
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=device)
        self.y_in = torch.empty_like(y0, memory_format=torch.contiguous_format)
        self.y_in.copy_(y0)
        # solvers that carry state between steps (reversible Heun: f, g, z) get it as further static inputs
        self.extra_in = [torch.empty_like(e, memory_format=torch.contiguous_format).copy_(e) for e in extra0]
        self._set_seed(bm)
        bm._entropy_dev = self.seed_dev
        # the captured kernels hold raw pointers into this solver's Brownian motion (device copy of the cell edges)
        # and into the plan's stage-time tensors: keep both alive for as long as the graph
        self._keepalive = solver
        try:
            self.plan = solver._plan(self.y_in, ts)
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):           # warm-up outside capture (lazy inits, allocator)
                solver._extra = tuple(self.extra_in)
                solver._run(self.plan, self.y_in)
            torch.cuda.current_stream(device).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            # thread_local: API calls from other threads (e.g. the RCCL watchdog of a multi-GPU run) must not
            # invalidate this capture
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                solver._extra = tuple(self.extra_in)
                self.ys = solver._run(self.plan, self.y_in)
                self.extra_out = tuple(solver._extra)
        finally:
            bm._entropy_dev = None
        self.graph.replay()     # capture only records: run once so that `ys` holds this solve's result

    def _set_seed(self, bm):
        key = bm._key
        self.seed_dev.fill_(key - (1 << 64) if key >= (1 << 63) else key)   # two's complement into int64

    def result(self):
        return self.ys.clone(), tuple(e.clone() for e in self.extra_out)

    def replay(self, bm, y0, extra0=()):
        self.y_in.copy_(y0)
        for dst, src in zip(self.extra_in, extra0):
            dst.copy_(src)
        self._set_seed(bm)
        self.graph.replay()
        return self.result()


def _signature(solver, y0, ts_host):
    bm = solver.bm
    return (type(solver).__name__, tuple(y0.shape), y0.dtype, str(y0.device), tuple(ts_host.tolist()),
            float(solver.dt), tuple(bm.shape), bm.levy_area_approximation, bm.row_offset,
            None if bm._edges is None else bm._edges.tobytes(), bm._max_depth, bm._snap,
            tuple(sorted((k, v) for k, v in solver.options.items() if isinstance(v, (bool, int, float, str)))))


def replay_or_capture(solver, y0, ts, extra0=()):
    """Run the solve through a cached HIP graph (capturing it on first use); returns (ys, extra solver state)."""
    bm = solver.bm
    if not isinstance(bm, BrownianInterval) or bm._rootW is not None or bm._rootH is not None:
        warnings.warn("hip_graph=True needs a torchsde_amd.BrownianInterval without pinned W/H; running eagerly.")
        solver._extra = tuple(extra0)
        return solver._run(solver._plan(y0, ts), y0), solver._extra
    from . import timegrid
    ts_host = timegrid.ts_to_host(ts)
    base = solver.sde
    while hasattr(base, "_base_sde"):    # ForwardSDE / RenameMethodsSDE / SDELogqp wrappers are rebuilt per call
        base = base._base_sde
    cache = getattr(base, _CACHE_ATTR, None)
    if cache is None:
        cache = {}
        setattr(base, _CACHE_ATTR, cache)
    # the grid the Brownian motion will have after adoption is part of the signature
    probe_plan_needed = not bm.frozen
    if probe_plan_needed:
        bm.adopt_grid(timegrid.build(ts_host, solver.dt).t_f64())
    sig = _signature(solver, y0, ts_host)
    captured = cache.get(sig)
    if captured is None:
        captured = _CapturedSolve(solver, y0, ts, extra0)
        _remember(cache, sig, captured)
        return captured.result()
    return captured.replay(bm, y0, extra0)


class _CapturedBackward:
    """The launch-only backward sweep of ``sdeint_adjoint`` (``adjoint._run_backward``: re-materialised increments,
    the user's f/g and their VJPs through autograd, ``tsde_aug_update``) as ONE HIP graph. Static inputs: the stored
    forward states ``ys`` and the incoming gradients ``grad_ys`` -- plus, for the reversible-Heun pair, the solver's
    final (f, g, z) and their cotangents -- (copied in before each replay), the Brownian seed
    (device word) and the parameters themselves (read in place: an optimiser step is seen by the next replay)."""

    def __init__(self, run, bm, inputs):
        device = inputs[0].device
        self.seed_dev = torch.zeros(1, dtype=torch.int64, device=device)
        self.static = [torch.empty_like(x, memory_format=torch.contiguous_format) for x in inputs]
        self._load(bm, inputs)
        bm._entropy_dev = self.seed_dev
        # `run` owns the plan (stage-time tensors) and the Brownian motion (device copy of the cell edges) that the
        # captured kernels point into: keep it alive for as long as the graph
        self._keepalive = run
        try:
            side = torch.cuda.Stream(device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):           # warm-up outside capture (lazy inits, allocator)
                run(*self.static)
            torch.cuda.current_stream(device).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                self.out = list(run(*self.static))
        finally:
            bm._entropy_dev = None

    def _load(self, bm, inputs):
        for dst, src in zip(self.static, inputs):
            dst.copy_(src)
        key = bm._key
        self.seed_dev.fill_(key - (1 << 64) if key >= (1 << 63) else key)

    def replay(self, bm, inputs):
        self._load(bm, inputs)
        self.graph.replay()
        return [o.clone() for o in self.out]


def cached_backward(sde, bm, signature, capture):
    """The cached HIP graph of the adjoint's backward sweep for this structure; `capture()` builds it on a miss
    (and may return None: then nothing is cached and the backward pass runs eagerly).
    `signature` identifies the sweep's structure; the Brownian structure is appended here."""
    if bm._rootW is not None or bm._rootH is not None:
        warnings.warn("hip_graph=True needs a torchsde_amd.BrownianInterval without pinned W/H; running eagerly.")
        return None
    base = sde
    while hasattr(base, "_base_sde"):
        base = base._base_sde
    cache = getattr(base, _CACHE_ATTR, None)
    if cache is None:
        cache = {}
        setattr(base, _CACHE_ATTR, cache)
    sig = signature + (tuple(bm.shape), bm.levy_area_approximation, bm.row_offset,
                       None if bm._edges is None else bm._edges.tobytes(), bm._max_depth, bm._snap)
    captured = cache.get(sig)
    if captured is None:
        captured = capture()
        if captured is not None:
            _remember(cache, sig, captured)
    return captured

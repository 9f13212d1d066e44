"""Host-side time grid of a fixed-step solve.

The reference advances time inside its stepping loop with 0-d tensors in ``ts.dtype``:
``next_t = min(curr_t + step_size, ts[-1])`` (torchsde/_core/base_solver.py:114-116), never clipping to the
intermediate output times and linearly interpolating back to them (:147, interp.py:15-18). Each of those
tensor operations costs a device->host sync on a GPU. Here the whole grid -- step boundaries, per-step
``dt``, stage times and interpolation weights -- is computed once on the host with numpy scalars of the
same dtype (identical IEEE arithmetic, e.g. float32 ``dt=1e-3`` over [0,1] gives 1001 steps exactly like
the reference) and uploaded in one copy.
"""
import collections
from dataclasses import dataclass, field
from typing import List, Tuple

import numpy as np
import torch

_NP = {torch.float32: np.float32, torch.float64: np.float64, torch.float16: np.float16}


@dataclass
class TimeGrid:
    t: np.ndarray                      # (N+1,) step boundaries, in ts.dtype
    dt: np.ndarray                     # (N,)   t[k+1] - t[k] rounded in ts.dtype (what `f * dt` uses)
    outputs: List[Tuple[int, int, float, float]] = field(default_factory=list)
    # outputs[j] = (k_prev, k_curr, w0, w1): ys[j+1] = w0 * y[k_prev] + w1 * y[k_curr]

    @property
    def n_steps(self):
        return self.dt.shape[0]

    def t_f64(self):
        return self.t.astype(np.float64)


_GRIDS = collections.OrderedDict()       # (ts bytes, dtype, step) -> TimeGrid, the 32 most recent


def _step_value(dt, np_dtype):
    """`curr_t + dt` with a Python-number dt: torch rounds the scalar to the tensor dtype first."""
    return np_dtype(dt)


def build(ts_host: np.ndarray, dt) -> TimeGrid:
    """ts_host: 1-D numpy array in the dtype of `ts`; dt: Python number (or 0-d tensor of the same dtype)."""
    np_dtype = ts_host.dtype.type
    if torch.is_tensor(dt):
        if dt.numel() != 1:
            raise ValueError("`dt` must be a scalar.")
        if _NP.get(dt.dtype) is not np_dtype:
            # A 0-d tensor dt of another dtype would promote `curr_t + dt`; not worth emulating.
            raise ValueError("A tensor-valued `dt` must have the dtype of `ts`.")
        step = np_dtype(dt.item())
    else:
        step = _step_value(dt, np_dtype)
    if not step > 0:
        raise ValueError("`dt` must be positive.")
    # A pure function of (ts, dt) whose Python loop costs ~0.2 us per step: a training loop asks for the same grid at
    # every iteration (and a solve asks two or three times), so the most recent grids are remembered. The arrays of a
    # TimeGrid are read-only.
    key = (ts_host.tobytes(), ts_host.dtype.str, float(step))
    hit = _GRIDS.get(key)
    if hit is not None:
        _GRIDS.move_to_end(key)
        return hit
    t_end = ts_host[-1]
    times = [ts_host[0]]
    outputs = []
    prev_k = curr_k = 0
    curr_t = ts_host[0]
    for out_t in ts_host[1:]:
        while curr_t < out_t:
            nxt = curr_t + step          # rounded once in np_dtype
            next_t = nxt if nxt <= t_end else t_end
            times.append(next_t)
            prev_k, curr_k = curr_k, curr_k + 1
            curr_t = next_t
        t0, t1 = times[prev_k], times[curr_k]
        if t1 == t0:   # out_t == ts[0] cannot happen for strictly increasing ts; guard anyway
            w0, w1 = np_dtype(0), np_dtype(1)
        else:
            # (t1 - t) / (t1 - t0) and (t - t0) / (t1 - t0), each op rounded in ts.dtype (interp.py:17)
            w0 = (t1 - out_t) / (t1 - t0)
            w1 = (out_t - t0) / (t1 - t0)
        outputs.append((prev_k, curr_k, float(w0), float(w1)))
    t = np.asarray(times, dtype=ts_host.dtype)
    step_dt = (t[1:] - t[:-1]).astype(ts_host.dtype)
    t.setflags(write=False)
    step_dt.setflags(write=False)
    grid = TimeGrid(t=t, dt=step_dt, outputs=tuple(outputs))
    _GRIDS[key] = grid
    while len(_GRIDS) > 32:
        _GRIDS.popitem(last=False)
    return grid


def ts_to_host(ts: torch.Tensor) -> np.ndarray:
    """The output times on the host (read-only array): one device->host copy, the only sync of a fixed-step solve --
    and none at all when the same `ts` tensor comes back unchanged (every iteration of a training loop): the copy is
    remembered ON the tensor object together with its version counter, which every in-place op bumps (writes through
    `ts.data` bypass that counter, as they bypass autograd's own checks: do not edit `ts` that way between solves)."""
    if ts.dtype not in _NP:
        raise ValueError(f"Unsupported dtype for `ts`: {ts.dtype}")
    if ts.device.type == "cpu":
        # no sync to save, and a CPU tensor can be edited through its numpy alias without its version moving
        host = ts.detach().numpy().copy()
        host.setflags(write=False)
        return host
    cached = getattr(ts, "_tsde_host", None)
    if cached is not None and cached[0] == ts._version:
        return cached[1]
    host = ts.detach().cpu().numpy()
    remember(ts, host)
    return host


def remember(ts: torch.Tensor, host: np.ndarray) -> None:
    """Attach a host copy of `ts` to the tensor (for callers that built the tensor FROM host values)."""
    host.setflags(write=False)
    try:
        ts._tsde_host = (ts._version, host)
    except (AttributeError, RuntimeError):      # a tensor subclass that refuses attributes: just do not cache
        pass

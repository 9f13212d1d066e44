"""Brownian motion objects with the reference's interface, backed by the counter-RNG HIP generator.

Interface mirrored (reference paths under torchsde/_brownian/):
  BaseBrownian            brownian_base.py:18-50
  BrownianInterval        brownian_interval.py:353-785 (constructor arguments, properties, __call__ semantics,
                          warnings and error types)
  ReverseBrownian         derived.py:22-49
  BrownianPath / Tree     derived.py:52-191 (thin wrappers)
  brownian_interval_like  derived.py:194-205

What is different underneath: the reference stores a binary tree of intervals with per-node numpy
SeedSequence seeds, an LRU cache of (W, H) tensors and draws a full-size ``torch.randn`` per visited node.
Here the path is *stateless*: [t0, t1] is covered by top-level **cells** (the solver's time grid, a uniform
``dt`` grid, or one cell), each cell is the root of a virtual dyadic Brownian-bridge tree, and every normal
is Philox-4x32-10 of (entropy, element, cell, node). A query is one kernel launch
(``tsde_brownian_query``); fixed-step solvers do not even launch it -- they regenerate the increment of
"their" cell in registers inside the step kernel.
"""
import abc
import math
import warnings

import numpy as np
import torch

from . import _native
from .settings import LEVY_AREA_APPROXIMATIONS

# In-cell dyadic levels resolved before the leaf rule applies (exact split at the query point).
_EXACT_DEPTH = 32
_MAX_DEPTH = 40


class BaseBrownian(metaclass=abc.ABCMeta):
    """The Brownian plug-in protocol accepted by ``sdeint`` (reference: brownian_base.py:18-50)."""
    __slots__ = ()

    @abc.abstractmethod
    def __call__(self, ta, tb=None, return_U=False, return_A=False):
        raise NotImplementedError

    @abc.abstractmethod
    def __repr__(self):
        raise NotImplementedError

    @property
    @abc.abstractmethod
    def dtype(self):
        raise NotImplementedError

    @property
    @abc.abstractmethod
    def device(self):
        raise NotImplementedError

    @property
    @abc.abstractmethod
    def shape(self):
        raise NotImplementedError

    @property
    @abc.abstractmethod
    def levy_area_approximation(self):
        raise NotImplementedError

    def size(self):
        return self.shape


def _scalar_like(x):
    return isinstance(x, (int, float)) or (isinstance(x, torch.Tensor) and x.numel() == 1)


def _resolve_tensor_info(W, H, size, dtype, device):
    """size/dtype/device from the explicit arguments and/or the pinned W, H (brownian_interval.py:49-75)."""
    given = [t for t in (W, H) if torch.is_tensor(t)]
    if dtype is None and not given:
        dtype = torch.get_default_dtype()
    if device is None and not given:
        device = torch.device("cpu")
    sizes = ([] if size is None else [tuple(size)]) + [tuple(t.shape) for t in given]
    dtypes = ([] if dtype is None else [dtype]) + [t.dtype for t in given]
    devices = ([] if device is None else [torch.device(device)]) + [t.device for t in given]
    if not sizes:
        raise ValueError("Must either specify `size` or pass in `W` or `H` to implicitly define the size.")
    if any(s != sizes[0] for s in sizes):
        raise ValueError("Multiple sizes found. Make sure `size` and `W` or `H` are consistent.")
    if any(d != dtypes[0] for d in dtypes):
        raise ValueError("Multiple dtypes found. Make sure `dtype` and `W` or `H` are consistent.")
    if any(_dev_key(d) != _dev_key(devices[0]) for d in devices):
        raise ValueError("Multiple devices found. Make sure `device` and `W` or `H` are consistent.")
    return sizes[0], dtypes[0], devices[0]


def _dev_key(d):
    d = torch.device(d)
    return (d.type, 0 if (d.type == "cuda" and d.index is None) else d.index)


class BrownianInterval(BaseBrownian):
    """Brownian motion on [t0, t1] with fixed entropy, queryable on arbitrary sub-intervals.

    Same constructor, query semantics, warnings and errors as the reference class. ``pool_size`` and
    ``cache_size`` are accepted for compatibility and ignored (nothing is cached). One extension:
    ``row_offset`` -- the global index of this shard's first batch row -- so that a batch sharded over
    several GPUs draws exactly the rows an unsharded run would.
    """

    def __init__(self, t0=0., t1=1., size=None, dtype=None, device=None, entropy=None, dt=None, tol=0.,
                 pool_size=8, cache_size=45, halfway_tree=False,
                 levy_area_approximation=LEVY_AREA_APPROXIMATIONS.none, W=None, H=None, *, row_offset=0):
        if not _scalar_like(t0):
            raise ValueError("Initial time t0 should be a float or 0-d torch.Tensor.")
        if not _scalar_like(t1):
            raise ValueError("Terminal time t1 should be a float or 0-d torch.Tensor.")
        if dt is not None and not _scalar_like(dt):
            raise ValueError("Expected average time step dt should be a float or 0-d torch.Tensor.")
        if t0 > t1:
            raise ValueError(f"Initial time {t0} should be less than terminal time {t1}.")
        t0, t1 = float(t0), float(t1)
        dt = None if dt is None else float(dt)
        tol = float(tol)
        if halfway_tree:
            if tol <= 0.:
                raise ValueError("`tol` should be positive.")
            if dt is not None:
                raise ValueError("`dt` is not used and should be set to `None` if `halfway_tree` is True.")
        elif tol < 0.:
            raise ValueError("`tol` should be non-negative.")
        size, dtype, device = _resolve_tensor_info(W, H, size, dtype, device)
        if entropy is None:
            entropy = np.random.randint(0, 2 ** 31 - 1)
        if levy_area_approximation not in LEVY_AREA_APPROXIMATIONS:
            raise ValueError(f"`levy_area_approximation` must be one of {LEVY_AREA_APPROXIMATIONS}, but got "
                             f"'{levy_area_approximation}'.")
        for name, t in (("W", W), ("H", H)):
            if t is not None:
                if not torch.is_tensor(t):
                    raise ValueError(f"{name}={t} should be a Tensor.")
                if not t.is_floating_point():
                    raise ValueError(f"{name}={t} should be floating point.")
        if dtype not in (torch.float32, torch.float64):
            raise ValueError(f"torchsde_amd Brownian motion supports float32/float64, got {dtype}.")

        self._t0, self._t1 = t0, t1
        self._size, self._dtype, self._device = size, dtype, torch.device(device)
        self._entropy = int(entropy)
        self._key = self._entropy & 0xFFFFFFFFFFFFFFFF
        self._levy = levy_area_approximation
        self._dt, self._tol = dt, tol
        self._pool_size, self._cache_size, self._halfway_tree = pool_size, cache_size, halfway_tree
        self._have_H = levy_area_approximation in (LEVY_AREA_APPROXIMATIONS.space_time,
                                                   LEVY_AREA_APPROXIMATIONS.davie, LEVY_AREA_APPROXIMATIONS.foster)
        self._have_A = levy_area_approximation in (LEVY_AREA_APPROXIMATIONS.davie, LEVY_AREA_APPROXIMATIONS.foster)
        self._numel = int(np.prod(size)) if len(size) > 0 else 1
        self._channels = size[-1] if len(size) >= 2 else 1
        self._row_offset = int(row_offset)
        self._elem0 = self._row_offset * (self._numel // size[0] if len(size) >= 1 and size[0] > 0 else 1)
        self._rootW = None if W is None else _native.contiguous(W.detach())
        self._rootH = None if H is None else _native.contiguous(H.detach())
        if tol > 0.:
            ndigits = -int(math.log10(tol))
            self._round = lambda x: round(x, ndigits)
        else:
            self._round = lambda x: x

        # ---- the cell structure ("grid") -----------------------------------------------------------
        self._entropy_dev = None  # set by graph capture: device word that overrides the seed at replay
        self._edges = None       # np.float64 (n_cells + 1,)
        self._edges_dev = None   # the same on the device, for the query kernel
        self._max_depth, self._snap = _EXACT_DEPTH, 0
        if halfway_tree:
            # Path determined by entropy alone: one cell, dyadic descent down to `tol`, snapping leaf rule.
            self._freeze(np.array([t0, t1], dtype=np.float64))
            depth = int(math.ceil(math.log2(max((t1 - t0) / tol, 1.0)))) + 1 if t1 > t0 else 0
            self._max_depth, self._snap = min(max(depth, 0), _MAX_DEPTH), 1
        elif self._rootW is not None or self._rootH is not None:
            self._freeze(np.array([t0, t1], dtype=np.float64))
        elif dt is not None and t1 > t0:
            self._freeze(uniform_edges(t0, t1, dt))

    # ---- grid management ---------------------------------------------------------------------------
    @property
    def frozen(self):
        return self._edges is not None

    def _freeze(self, edges):
        self._edges = np.ascontiguousarray(edges, dtype=np.float64)
        self._edges_dev = None

    def adopt_grid(self, grid):
        """Let a fixed-step solver make its time grid the cell structure (only while no query has fixed one).

        The reference's sample path also depends on the query history (brownian_interval.py:443-447, 623-634);
        here that dependence is reduced to this single decision. Returns True if the grid was adopted.
        """
        if self.frozen:
            return False
        grid = np.asarray(grid, dtype=np.float64)
        if grid.ndim != 1 or grid.size < 2 or not np.all(np.diff(grid) > 0):
            return False
        if grid[0] < self._t0 or grid[-1] > self._t1:
            return False
        edges = grid
        if grid[0] > self._t0:
            edges = np.concatenate([[self._t0], edges])
        if grid[-1] < self._t1:
            edges = np.concatenate([edges, [self._t1]])
        self._freeze(edges)
        return True

    def match_grid(self, grid):
        """Cell index of every step [grid[k], grid[k+1]] if each one is exactly one cell, else None."""
        if not self.frozen or self._snap or self._rootW is not None or self._rootH is not None:
            return None
        grid = np.asarray(grid, dtype=np.float64)
        idx = np.searchsorted(self._edges, grid)
        if idx[-1] >= self._edges.size or not np.array_equal(self._edges[idx], grid):
            return None
        if not np.all(np.diff(idx) == 1):
            return None
        return idx[:-1]

    def cell_width(self, k):
        return float(self._edges[k + 1] - self._edges[k])

    def _device_edges(self):
        if self._edges_dev is None:
            self._edges_dev = torch.from_numpy(self._edges).to(self._device)
        return self._edges_dev

    def locate(self, ta, tb):
        """(ca, cb): cells containing the left end ta and the right end tb of a query (host logic only)."""
        if not self.frozen:
            self._freeze(np.array([self._t0, self._t1], dtype=np.float64))
        n_cells = self._edges.size - 1
        ca = int(np.searchsorted(self._edges, ta, side="right")) - 1
        cb = int(np.searchsorted(self._edges, tb, side="left")) - 1
        return min(max(ca, 0), n_cells - 1), min(max(cb, 0), n_cells - 1)

    # ---- queries -------------------------------------------------------------------------------------
    def __call__(self, ta, tb=None, return_U=False, return_A=False):
        if tb is None:
            warnings.warn(f"{self.__class__.__name__} is optimised for interval-based queries, not point evaluation.")
            ta, tb = self._t0, ta
            tb_name = "ta"
        else:
            tb_name = "tb"
        ta, tb = float(ta), float(tb)
        if ta < self._t0:
            warnings.warn(f"Should have ta>=t0 but got ta={ta} and t0={self._t0}.")
            ta = self._t0
        if tb < self._t0:
            warnings.warn(f"Should have {tb_name}>=t0 but got {tb_name}={tb} and t0={self._t0}.")
            tb = self._t0
        if ta > self._t1:
            warnings.warn(f"Should have ta<=t1 but got ta={ta} and t1={self._t1}.")
            ta = self._t1
        if tb > self._t1:
            warnings.warn(f"Should have {tb_name}<=t1 but got {tb_name}={tb} and t1={self._t1}.")
            tb = self._t1
        if ta > tb:
            raise RuntimeError(f"Query times ta={ta:.3f} and tb={tb:.3f} must respect ta <= tb.")
        A = None
        with _native.on_device_of(self._device):
            if return_A and self._have_A:
                W, U, A = self.increment_with_levy_area(ta, tb)
            else:
                W, U = self.increment(ta, tb, want_U=self._have_H)
        if return_U:
            return (W, U, A) if return_A else (W, U)
        return (W, A) if return_A else W

    def increment_with_levy_area(self, ta, tb, iterated=None):
        """(W, U, A) with A the Davie / Foster approximation of the Levy area of [ta, tb] built from the exact
        (W, H) of that interval (brownian_interval.py:78-99). Like the reference's, A is an approximation and is
        not additive over sub-intervals; its antisymmetric noise is keyed on the interval so re-queries agree.
        `iterated=(dt, ito)`: the third result is the matrix of iterated integrals I = (W W^T - [diag] dt)/2 + A
        (Ito; Stratonovich: no dt term) of the general-noise Milstein step instead, in one kernel where it applies."""
        import struct
        from . import kernels as K
        ta_r, tb_r = self._round(ta), self._round(tb)
        out_H = torch.empty(self._size, dtype=self._dtype, device=self._device)
        W, U = self.increment(ta, tb, want_U=True, out_H=out_H)
        if len(self._size) in (0, 1):   # one Brownian channel per batch element: no Levy area (:81-84)
            return W, U, torch.zeros_like(W)
        if not (ta_r < tb_r):
            A = torch.zeros(self._size + self._size[-1:], dtype=self._dtype, device=self._device)
            if iterated is not None:
                m = self._size[-1]
                A = K.iterated_integrals(W.reshape(-1, m), A.reshape(-1, m, m), iterated[0], iterated[1]).reshape(A.shape)
            return W, U, A
        ca, _ = self.locate(ta_r, tb_r)
        bits_a, bits_b = (struct.unpack("<Q", struct.pack("<d", x))[0] for x in (ta_r, tb_r))
        mix = (bits_a * 0x9E3779B97F4A7C15 + ((bits_b << 31) | (bits_b >> 33)) * 0xBF58476D1CE4E5B9) & ((1 << 64) - 1)
        node = (mix ^ (mix >> 29)) & ((1 << 42) - 1)
        m = self._size[-1]
        foster = self._levy == LEVY_AREA_APPROXIMATIONS.foster
        if iterated is not None:      # the general-noise Milstein step wants I = (W W^T - [diag] dt)/2 + A, not A itself
            dt, ito = iterated
            fused = K.levy_iterated_integrals(W.reshape(-1, m), out_H.reshape(-1, m), tb_r - ta_r, foster, self._key,
                                              self._elem0, ca, node, dt, ito, self._entropy_dev)
            if fused is not None:
                return W, U, fused.reshape(self._size + (m,))
        A = K.levy_area(W.reshape(-1, m), out_H.reshape(-1, m), tb_r - ta_r, foster, self._key, self._elem0, ca, node,
                        self._entropy_dev)
        if iterated is not None:
            return W, U, K.iterated_integrals(W.reshape(-1, m), A, iterated[0], iterated[1]).reshape(self._size + (m,))
        return W, U, A.reshape(self._size + (m,))

    def increment(self, ta, tb, want_U=False, out_W=None, out_U=None, out_H=None):
        """W (and U, H) over [ta, tb] for host floats ta <= tb inside [t0, t1]; one kernel launch."""
        ta, tb = self._round(ta), self._round(tb)
        want_U = want_U and self._have_H
        if out_W is None:
            out_W = torch.empty(self._size, dtype=self._dtype, device=self._device)
        if want_U and out_U is None:
            out_U = torch.empty(self._size, dtype=self._dtype, device=self._device)
        if not (ta < tb) or self._numel == 0:     # (an empty batch: empty increments, nothing to launch)
            out_W.zero_()
            if want_U:
                out_U.zero_()
            if out_H is not None:
                out_H.zero_()
            return out_W, (out_U if want_U else None)
        _native.require_device(out_W)
        lib = _native.load()
        ca, cb = self.locate(ta, tb)
        edges = self._device_edges()
        code = lib.tsde_brownian_query(
            _native.ptr(out_W), _native.ptr(out_U if want_U else None),
            _native.ptr(out_H if self._have_H else None), self._numel, self._key, self._elem0,
            _native.ptr(edges), ca, cb, ta, tb, _native.ptr(self._rootW), _native.ptr(self._rootH),
            1 if self._have_H else 0, self._max_depth, self._snap,
            None if self._entropy_dev is None else self._entropy_dev.data_ptr(), _native.dtype_code(self._dtype),
            _native.stream_ptr(self._device))
        _native.check(code, "tsde_brownian_query")
        return out_W, (out_U if want_U else None)

    # ---- description -----------------------------------------------------------------------------------
    def __repr__(self):
        dt = None if self._dt is None else f"{self._dt:.3f}"
        return (f"{self.__class__.__name__}(t0={self._t0:.3f}, t1={self._t1:.3f}, size={self._size}, "
                f"dtype={self._dtype}, device={repr(self._device)}, entropy={self._entropy}, dt={dt}, "
                f"tol={self._tol}, pool_size={self._pool_size}, cache_size={self._cache_size}, "
                f"levy_area_approximation={repr(self._levy)})")

    @property
    def shape(self):
        return self._size

    @property
    def dtype(self):
        return self._dtype

    @property
    def device(self):
        return self._device

    @property
    def entropy(self):
        return self._entropy

    @property
    def levy_area_approximation(self):
        return self._levy

    @property
    def dt(self):
        return self._dt

    @property
    def tol(self):
        return self._tol

    @property
    def pool_size(self):
        return self._pool_size

    @property
    def cache_size(self):
        return self._cache_size

    @property
    def halfway_tree(self):
        return self._halfway_tree

    @property
    def row_offset(self):
        return self._row_offset

    def size(self):
        return self._size


def uniform_edges(t0, t1, dt):
    """Cell edges t0 + k*dt (double), last cell truncated at t1."""
    n = int(math.ceil((t1 - t0) / dt - 1e-9))
    n = max(n, 1)
    edges = t0 + dt * np.arange(n + 1, dtype=np.float64)
    edges[-1] = t1
    if n >= 2 and edges[-2] >= t1:
        edges = np.concatenate([edges[:-2], [t1]])
    return edges


class ReverseBrownian(BaseBrownian):
    """Time-flipped view used by the adjoint: (ta, tb) -> base(-tb, -ta), no sign change (derived.py:22-49)."""

    def __init__(self, base_brownian):
        super().__init__()
        self.base_brownian = base_brownian

    def __call__(self, ta, tb=None, return_U=False, return_A=False):
        return self.base_brownian(-tb, -ta, return_U=return_U, return_A=return_A)

    def __repr__(self):
        return f"{self.__class__.__name__}(base_brownian={self.base_brownian})"

    @property
    def dtype(self):
        return self.base_brownian.dtype

    @property
    def device(self):
        return self.base_brownian.device

    @property
    def shape(self):
        return self.base_brownian.shape

    @property
    def levy_area_approximation(self):
        return self.base_brownian.levy_area_approximation


class _IntervalWrapper(BaseBrownian):
    """Shared body of the legacy BrownianPath / BrownianTree wrappers (derived.py:52-191)."""

    def __call__(self, t, tb=None, return_U=False, return_A=False):
        out = self._interval(t, tb, return_U=return_U, return_A=return_A)
        if tb is None and not return_U and not return_A:
            out = out + self._w0
        return out

    def __repr__(self):
        return f"{self.__class__.__name__}(interval={self._interval})"

    @property
    def dtype(self):
        return self._interval.dtype

    @property
    def device(self):
        return self._interval.device

    @property
    def shape(self):
        return self._interval.shape

    @property
    def levy_area_approximation(self):
        return self._interval.levy_area_approximation


class BrownianPath(_IntervalWrapper):
    """Brownian path started at (t0, w0) on [t0, t0+1]."""

    def __init__(self, t0, w0, window_size=8):
        self._w0 = w0
        self._interval = BrownianInterval(t0=t0, t1=t0 + 1, size=w0.shape, dtype=w0.dtype, device=w0.device,
                                          cache_size=None)
        super().__init__()


class BrownianTree(_IntervalWrapper):
    """Brownian motion whose sample path depends on `entropy` only (dyadic tree resolved to `tol`)."""

    def __init__(self, t0, w0, t1=None, w1=None, entropy=None, tol=1e-6, pool_size=24, cache_depth=9, safety=None):
        if t1 is None:
            t1 = t0 + 1
        self._w0 = w0
        self._interval = BrownianInterval(t0=t0, t1=t1, size=w0.shape, dtype=w0.dtype, device=w0.device,
                                          entropy=entropy, tol=tol, pool_size=pool_size, halfway_tree=True,
                                          W=None if w1 is None else w1 - w0)
        super().__init__()


def brownian_interval_like(y, t0=0., t1=1., size=None, dtype=None, device=None, **kwargs):
    """A BrownianInterval with the size, dtype and device of ``y`` unless overridden."""
    return BrownianInterval(t0=t0, t1=t1, size=y.shape if size is None else size,
                            dtype=y.dtype if dtype is None else dtype,
                            device=y.device if device is None else device, **kwargs)

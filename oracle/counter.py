"""ctypes front-end of oracle/counter_brownian.c (ORACLE: test infrastructure only)."""
import ctypes

import numpy as np

from . import build as _build

_u32p = ctypes.POINTER(ctypes.c_uint32)
_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(_build.build())
        L.orc_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
        L.orc_philox4x32_10.restype = None
        L.orc_noise_counter.argtypes = [ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint32, _u32p]
        L.orc_noise_counter.restype = None
        L.orc_normals.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                  ctypes.c_uint64, ctypes.c_uint32]
        L.orc_normals.restype = None
        for suf, real in (("f32", ctypes.c_float), ("f64", ctypes.c_double)):
            q = getattr(L, f"orc_query_{suf}")
            q.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_uint64,
                          ctypes.c_uint64, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_double,
                          ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
            q.restype = None
            s = getattr(L, f"orc_bridge_split_{suf}")
            rp = ctypes.POINTER(real)
            s.argtypes = [real, real, ctypes.c_double, ctypes.c_double, ctypes.c_double, real, real, ctypes.c_int,
                          rp, rp, rp, rp]
            s.restype = None
            m = getattr(L, f"orc_interval_merge_{suf}")
            m.argtypes = [rp, rp, ctypes.c_double, real, real, ctypes.c_double, ctypes.c_int]
            m.restype = None
        _lib = L
    return _lib


def philox(ctr, key):
    c = (ctypes.c_uint32 * 4)(*ctr)
    k = (ctypes.c_uint32 * 2)(*key)
    o = (ctypes.c_uint32 * 4)()
    lib().orc_philox4x32_10(c, k, o)
    return tuple(o)


def noise_counter(quad, cell, node, stream):
    o = (ctypes.c_uint32 * 4)()
    lib().orc_noise_counter(quad, cell, node, stream, o)
    return tuple(o)


def normals(n, entropy, elem0=0, cell=0, node=0, stream=0):
    """Standard normals (float64) of tree node `node` of `cell` for elements elem0 .. elem0+n-1."""
    out = np.empty(n, dtype=np.float64)
    lib().orc_normals(out.ctypes.data, n, entropy & 0xFFFFFFFFFFFFFFFF, elem0, cell, node, stream)
    return out


def query(n, entropy, edges, a, b, dtype=np.float32, elem0=0, have_h=False, max_depth=32, snap=0, rootW=None,
          rootH=None):
    """(W, U, H) over [a, b] of the counter-RNG Brownian path with cell edges `edges` (host array)."""
    edges = np.ascontiguousarray(edges, dtype=np.float64)
    n_cells = edges.size - 1
    ca = min(max(int(np.searchsorted(edges, a, side="right")) - 1, 0), n_cells - 1)
    cb = min(max(int(np.searchsorted(edges, b, side="left")) - 1, 0), n_cells - 1)
    W = np.empty(n, dtype=dtype)
    U = np.empty(n, dtype=dtype) if have_h else None
    H = np.empty(n, dtype=dtype) if have_h else None
    fn = lib().orc_query_f32 if dtype == np.float32 else lib().orc_query_f64
    rw = None if rootW is None else np.ascontiguousarray(rootW, dtype=dtype)
    rh = None if rootH is None else np.ascontiguousarray(rootH, dtype=dtype)
    fn(W.ctypes.data, None if U is None else U.ctypes.data, None if H is None else H.ctypes.data, n,
       entropy & 0xFFFFFFFFFFFFFFFF, elem0, edges.ctypes.data, ca, cb, float(a), float(b),
       None if rw is None else rw.ctypes.data, None if rh is None else rh.ctypes.data, int(have_h), max_depth, snap)
    return W, U, H


def bridge_split(W, H, lo, x, hi, X1, X2, have_h, dtype=np.float32):
    real = ctypes.c_float if dtype == np.float32 else ctypes.c_double
    fn = lib().orc_bridge_split_f32 if dtype == np.float32 else lib().orc_bridge_split_f64
    outs = [real() for _ in range(4)]
    fn(real(W), real(H), lo, x, hi, real(X1), real(X2), int(have_h), *[ctypes.byref(o) for o in outs])
    return tuple(o.value for o in outs)


def interval_merge(W, H, ha, Wi, Hi, hb, have_h, dtype=np.float32):
    real = ctypes.c_float if dtype == np.float32 else ctypes.c_double
    fn = lib().orc_interval_merge_f32 if dtype == np.float32 else lib().orc_interval_merge_f64
    w, h = real(W), real(H)
    fn(ctypes.byref(w), ctypes.byref(h), ha, real(Wi), real(Hi), hb, int(have_h))
    return w.value, h.value

"""Tensor-level wrappers of the C ABI: argument checking, ctypes marshalling, and autograd support.

Every update of the solver state goes through a function here, which launches a HIP kernel from
``libtorchsde_amd.so`` on torch's current stream. When gradients are requested *through the solver*
(``sdeint`` with tensors that require grad), the forward is still the fused kernel; the backward is
expressed with differentiable torch ops on the re-materialised increment (the step maps are linear in
``y0, f, g``), so higher-order derivatives work too.
"""
import collections
import ctypes

import numpy as np
import torch

from . import _native
from ._native import Noise, Seg


class NoiseSpec:
    """The Brownian increment of one step: either a generated grid cell or materialised tensors."""

    __slots__ = ("W", "U", "entropy", "elem0", "cell", "h", "bcast_d", "shape", "dtype", "device", "entropy_dev",
                 "_struct")

    def __init__(self, shape, dtype, device, W=None, U=None, entropy=0, elem0=0, cell=0, h=0.0, bcast_d=0,
                 entropy_dev=None):
        self.shape, self.dtype, self.device = tuple(shape), dtype, device
        self.W = None if W is None else _native.contiguous(W)
        self.U = None if U is None else _native.contiguous(U)
        self.entropy, self.elem0, self.cell, self.h, self.bcast_d = entropy, elem0, cell, h, bcast_d
        self.entropy_dev = entropy_dev   # int64 device tensor holding the run-time seed (HIP-graph replay)
        self._struct = None

    @classmethod
    def generated(cls, bm, cell, h):
        return cls(bm.shape, bm.dtype, bm.device, entropy=bm._key, elem0=bm._elem0, cell=int(cell), h=float(h),
                   entropy_dev=bm._entropy_dev)

    @classmethod
    def external(cls, W, U=None, bcast_d=0):
        return cls(W.shape, W.dtype, W.device, W=W, U=U, bcast_d=bcast_d)

    @property
    def is_generated(self):
        return self.W is None

    def struct(self):
        if self._struct is None:
            s = Noise()
            s.dW = None if self.W is None else self.W.data_ptr()
            s.dU = None if self.U is None else self.U.data_ptr()
            s.entropy, s.elem0, s.cell, s.reserved = self.entropy, self.elem0, self.cell, 0
            s.h, s.bcast_d = self.h, self.bcast_d
            s.entropy_dev = None if self.entropy_dev is None else self.entropy_dev.data_ptr()
            self._struct = s
        return ctypes.byref(self._struct)

    def materialise(self, need_U=False):
        """(W, U) as tensors of the noise shape (launches ``tsde_cell_increment`` for a generated cell)."""
        if self.W is not None:
            return self.W, self.U
        W = torch.empty(self.shape, dtype=self.dtype, device=self.device)
        U = torch.empty(self.shape, dtype=self.dtype, device=self.device) if need_U else None
        _native.require_device(W)
        lib = _native.load()
        code = lib.tsde_cell_increment(_native.ptr(W), _native.ptr(U), W.numel(), self.struct(),
                                       _native.dtype_code(self.dtype), _native.stream_ptr(self.device))
        _native.check(code, "tsde_cell_increment")
        return W, U


def _prep(ref, *tensors):
    """Contiguous, same dtype/shape as `ref`. (Kernels only see data pointers, so no detach is needed.)"""
    out = []
    for t in tensors:
        if t is None:
            out.append(None)
            continue
        if t.dtype != ref.dtype:
            t = t.to(ref.dtype)
        if t.shape != ref.shape:
            t = t.expand(ref.shape)
        out.append(t if t.is_contiguous() else t.contiguous())
    return out


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _launch_env(ref):
    """(library, dtype code, current HIP stream of ref's device) -- the per-launch constants, cheaply."""
    dev = ref.device
    if dev.type != "cuda":
        _native.require_device(ref)
    if _raw_stream is not None:
        stream = _raw_stream(dev.index if dev.index is not None else torch.cuda.current_device())
    else:
        stream = torch.cuda.current_stream(dev).cuda_stream
    return _native.load(), (_native.F32 if ref.dtype == torch.float32 else _native.dtype_code(ref.dtype)), stream


def _new_like(ref, out):
    return torch.empty_like(ref, memory_format=torch.contiguous_format) if out is None else out


def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(torch.is_tensor(t) and t.requires_grad for t in tensors)


# ---- raw launches --------------------------------------------------------------------------------------
def _raw_step_diag(y0, f, g, cf, cg, noise, out):
    if not y0.is_contiguous():
        y0 = y0.contiguous()
    f, g = _prep(y0, f, g)
    out = _new_like(y0, out)
    lib, dt_code, stream = _launch_env(y0)
    code = lib.tsde_step_diag(out.data_ptr(), y0.data_ptr(), f.data_ptr(), g.data_ptr(), y0.numel(),
                              cf, cg, noise.struct(), dt_code, stream)
    if code:
        _native.check(code, "tsde_step_diag")
    return out


def _raw_step_prod(y0, f, gp, cf, cg, out):
    if not y0.is_contiguous():
        y0 = y0.contiguous()
    f, gp = _prep(y0, f, gp)
    out = _new_like(y0, out)
    lib, dt_code, stream = _launch_env(y0)
    code = lib.tsde_step_prod(out.data_ptr(), y0.data_ptr(), f.data_ptr(), gp.data_ptr(), y0.numel(),
                              cf, cg, dt_code, stream)
    if code:
        _native.check(code, "tsde_step_prod")
    return out


def _shared_diffusion(g, B):
    """The (d, m) matrix behind a diffusion that is the SAME for every batch row (additive noise returned as
    ``sigma.expand(B, d, m)`` or with a leading 1), else None."""
    if g.dim() == 3 and B > 1 and (g.shape[0] == 1 or (g.shape[0] == B and g.stride(0) == 0)):
        return g[0]
    return None


def _raw_step_shared(y0, f, S, ca, cf, cg, weight_mode, cw, cu, rdt, noise, out):
    """y1 = (y0 + (ca*f)*cf) + cg*(S . w) for ONE (d, m) matrix S shared by the batch (``tsde_step_shared``)."""
    S = _native.contiguous(S if S.dtype == y0.dtype else S.to(y0.dtype))
    B, d = y0.shape
    out = _new_like(y0, out)
    lib, dt_code, stream = _launch_env(y0)
    code = lib.tsde_step_shared(out.data_ptr(), y0.data_ptr(), f.data_ptr(), S.data_ptr(), B, d, S.shape[-1],
                                float(ca), cf, cg, int(weight_mode), float(cw), float(cu), rdt, noise.struct(), dt_code,
                                stream)
    if code:
        _native.check(code, "tsde_step_shared")
    return out


def _raw_step_general(y0, f, g, cf, cg, noise, out):
    if not y0.is_contiguous():
        y0 = y0.contiguous()
    f, = _prep(y0, f)
    if g.dtype != y0.dtype:
        g = g.to(y0.dtype)
    B, d = y0.shape
    m = g.shape[-1]
    shared = _shared_diffusion(g, B)
    if shared is not None:
        # Batch-broadcast diffusion: g . dW is ONE dense (B, m) x (m, d) product -- the only matrix-core-shaped one on
        # this path -- instead of B copies of g streamed through the contraction kernel. One
        # launch: increments generated in registers as the MFMA B operand, the matrix staged in LDS, fused update.
        return _raw_step_shared(y0, f, shared, 1.0, cf, cg, 0, 0.0, 0.0, 0.0, noise, out)
    if g.shape != (B, d, m):
        g = g.expand(B, d, m)
    if not g.is_contiguous():
        g = g.contiguous()
    out = _new_like(y0, out)
    lib, dt_code, stream = _launch_env(y0)
    code = lib.tsde_step_general(out.data_ptr(), y0.data_ptr(), f.data_ptr(), g.data_ptr(), B, d, m,
                                 cf, cg, noise.struct(), dt_code, stream)
    if code:
        _native.check(code, "tsde_step_general")
    return out


# ---- differentiable wrappers -----------------------------------------------------------------------------
class _StepDiagFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y0, f, g, cf, cg, noise):
        ctx.cf, ctx.cg, ctx.noise = cf, cg, noise
        return _raw_step_diag(y0, f, g, cf, cg, noise, None)

    @staticmethod
    def backward(ctx, gy):
        W, _ = ctx.noise.materialise()
        if ctx.noise.bcast_d:
            W = W.reshape(-1, 1)
        return gy, gy * ctx.cf, gy * W * ctx.cg, None, None, None


class _StepProdFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y0, f, gp, cf, cg):
        ctx.cf, ctx.cg = cf, cg
        return _raw_step_prod(y0, f, gp, cf, cg, None)

    @staticmethod
    def backward(ctx, gy):
        return gy, gy * ctx.cf, gy * ctx.cg, None, None


class _StepGeneralFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y0, f, g, cf, cg, noise):
        ctx.cf, ctx.cg, ctx.noise = cf, cg, noise
        return _raw_step_general(y0, f, g, cf, cg, noise, None)

    @staticmethod
    def backward(ctx, gy):
        W, _ = ctx.noise.materialise()
        return gy, gy * ctx.cf, (gy.unsqueeze(-1) * W.unsqueeze(-2)) * ctx.cg, None, None, None


def step_diag(y0, f, g, cf, cg, noise, out=None):
    """y1 = (y0 + cf*f) + cg*(g*dW), diagonal noise."""
    if _needs_grad(y0, f, g):
        return _StepDiagFn.apply(y0, f.expand_as(y0), g.expand_as(y0), cf, cg, noise)
    return _raw_step_diag(y0, f, g, cf, cg, noise, out)


def step_prod(y0, f, gp, cf, cg, out=None):
    """y1 = (y0 + cf*f) + cg*gp."""
    if _needs_grad(y0, f, gp):
        return _StepProdFn.apply(y0, f.expand_as(y0), gp.expand_as(y0), cf, cg)
    return _raw_step_prod(y0, f, gp, cf, cg, out)


def step_general(y0, f, g, cf, cg, noise, out=None):
    """y1 = (y0 + cf*f) + cg*(g . dW), g:(B,d,m)."""
    if _needs_grad(y0, f, g):
        B, d = y0.shape
        return _StepGeneralFn.apply(y0, f.expand_as(y0), g.expand(B, d, g.shape[-1]), cf, cg, noise)
    return _raw_step_general(y0, f, g, cf, cg, noise, out)


def step_general_weighted(y0, f, g, ca, cf, cg, weight_mode, cw, cu, rdt, noise, out=None):
    """y1 = (y0 + (ca*f)*cf) + cg*(g . w), w built from (W, U): the stages of SRK for additive noise (SRA1)."""
    if _needs_grad(y0, f, g):
        W, U = noise.materialise(need_U=weight_mode != 0)
        w = W if weight_mode == 0 else ((cu * U) * rdt if weight_mode == 1 else (cw * W) + (cu * U) * rdt)
        return (y0 + (ca * f) * cf) + cg * torch.bmm(g, w.unsqueeze(-1)).squeeze(-1)
    y0 = _native.contiguous(y0)
    f, = _prep(y0, f)
    if g.dtype != y0.dtype:
        g = g.to(y0.dtype)
    B, d = y0.shape
    m = g.shape[-1]
    shared = _shared_diffusion(g, B)
    if shared is not None:      # SRA1's stages on a batch-broadcast diffusion: the matrix-core kernel, weighted form
        return _raw_step_shared(y0, f, shared, ca, float(cf), float(cg), weight_mode, cw, cu, float(rdt), noise, out)
    if g.shape != (B, d, m):
        g = g.expand(B, d, m)
    g = _native.contiguous(g)
    out = _new_like(y0, out)
    lib, dt_code, stream = _launch_env(y0)
    code = lib.tsde_step_general_w(_native.ptr(out), _native.ptr(y0), _native.ptr(f), _native.ptr(g), B, d, m,
                                   float(ca), float(cf), float(cg), int(weight_mode), float(cw), float(cu),
                                   float(rdt), noise.struct(), dt_code, stream)
    _native.check(code, "tsde_step_general_w")
    return out


# ---- Milstein --------------------------------------------------------------------------------------------
def milstein_v(noise, dt, ito, scale, like, want_W=False):
    """scale*(W^2 - dt) (Ito) or scale*W^2 as a tensor shaped like the noise (+ W itself if asked)."""
    v = torch.empty(noise.shape, dtype=like.dtype, device=like.device)
    Wt = torch.empty(noise.shape, dtype=like.dtype, device=like.device) if want_W else None
    lib, dt_code, stream = _launch_env(v)
    code = lib.tsde_milstein_v(_native.ptr(v), _native.ptr(Wt), v.numel(), float(dt), 1 if ito else 0, float(scale),
                               noise.struct(), dt_code, stream)
    _native.check(code, "tsde_milstein_v")
    return v, Wt


def milstein_weight(g, noise, dt, ito, scale):
    """g * scale*(W^2 - dt) (Ito) or g * scale*W^2: the cotangent of Milstein's diffusion VJP, in one launch."""
    g = _native.contiguous(g.detach())
    out = torch.empty_like(g)
    lib, dt_code, stream = _launch_env(g)
    code = lib.tsde_milstein_weight(out.data_ptr(), g.data_ptr(), g.numel(), float(dt), 1 if ito else 0, float(scale),
                                    noise.struct(), dt_code, stream)
    _native.check(code, "tsde_milstein_weight")
    return out


class _MilsteinDiagFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y0, f, g, gdg, dt, noise):
        ctx.dt, ctx.noise = dt, noise
        return _raw_milstein_diag(y0, f, g, gdg, dt, noise, None)

    @staticmethod
    def backward(ctx, gy):
        W, _ = ctx.noise.materialise()
        if ctx.noise.bcast_d:
            W = W.reshape(-1, 1)
        return gy, gy * ctx.dt, gy * W, gy, None, None


def _raw_milstein_diag(y0, f, g, gdg, dt, noise, out):
    y0 = _native.contiguous(y0)
    f, g, gdg = _prep(y0, f, g, gdg)
    out = _new_like(y0, out)
    lib, dt_code, stream = _launch_env(y0)
    code = lib.tsde_milstein_diag(_native.ptr(out), _native.ptr(y0), _native.ptr(f), _native.ptr(g),
                                  _native.ptr(gdg), y0.numel(), float(dt), noise.struct(), dt_code, stream)
    _native.check(code, "tsde_milstein_diag")
    return out


def milstein_diag(y0, f, g, gdg, dt, noise, out=None):
    """y1 = ((y0 + f*dt) + g*W) + gdg."""
    if _needs_grad(y0, f, g, gdg):
        return _MilsteinDiagFn.apply(y0, f.expand_as(y0), g.expand_as(y0), gdg.expand_as(y0), dt, noise)
    return _raw_milstein_diag(y0, f, g, gdg, dt, noise, out)


def milstein_gf_prime(y0, f, g, dt, sqrt_dt, ito, out=None):
    """y' = (y0 + dt*f) + g*sqrt_dt (Ito) or y0 + g*sqrt_dt (Stratonovich)."""
    if _needs_grad(y0, f, g):
        return (y0 + dt * f + g * sqrt_dt) if ito else (y0 + g * sqrt_dt)
    y0 = _native.contiguous(y0)
    f, g = _prep(y0, f, g)
    out = _new_like(y0, out)
    lib, dt_code, stream = _launch_env(y0)
    code = lib.tsde_milstein_gf_prime(_native.ptr(out), _native.ptr(y0), _native.ptr(f), _native.ptr(g), y0.numel(),
                                      float(dt), float(sqrt_dt), 1 if ito else 0, dt_code, stream)
    _native.check(code, "tsde_milstein_gf_prime")
    return out


class _MilsteinGfDiagFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y0, f, g, gprime, dt, sqrt_dt, ito, noise):
        ctx.dt, ctx.sqrt_dt, ctx.ito, ctx.noise = dt, sqrt_dt, ito, noise
        return _raw_milstein_gf_diag(y0, f, g, gprime, dt, sqrt_dt, ito, noise, None)

    @staticmethod
    def backward(ctx, gy):
        W, _ = ctx.noise.materialise()
        if ctx.noise.bcast_d:
            W = W.reshape(-1, 1)
        v = W * W - ctx.dt if ctx.ito else W * W
        q = v / (2 * ctx.sqrt_dt)
        return gy, gy * ctx.dt, gy * (W - q), gy * q, None, None, None, None


def _raw_milstein_gf_diag(y0, f, g, gprime, dt, sqrt_dt, ito, noise, out):
    y0 = _native.contiguous(y0)
    f, g, gprime = _prep(y0, f, g, gprime)
    out = _new_like(y0, out)
    lib, dt_code, stream = _launch_env(y0)
    code = lib.tsde_milstein_gf_diag(_native.ptr(out), _native.ptr(y0), _native.ptr(f), _native.ptr(g),
                                     _native.ptr(gprime), y0.numel(), float(dt), float(sqrt_dt), 1 if ito else 0,
                                     noise.struct(), dt_code, stream)
    _native.check(code, "tsde_milstein_gf_diag")
    return out


def milstein_gf_diag(y0, f, g, gprime, dt, sqrt_dt, ito, noise, out=None):
    """y1 = ((y0 + f*dt) + g*W) + ((g'-g)*v)/(2*sqrt_dt)."""
    if _needs_grad(y0, f, g, gprime):
        return _MilsteinGfDiagFn.apply(y0, f.expand_as(y0), g.expand_as(y0), gprime.expand_as(y0), dt, sqrt_dt, ito,
                                       noise)
    return _raw_milstein_gf_diag(y0, f, g, gprime, dt, sqrt_dt, ito, noise, out)


def milstein_gf_general_support(y0, f, g, dt, sqrt_dt, ito):
    """All m supporting states of derivative-free Milstein for general noise as one (m, B, d) batch:
    yk[k] = (y0 + dt*f) + g[:, :, k]*sqrt_dt (Ito) / (y0 + 0) + g[:, :, k]*sqrt_dt (Stratonovich)."""
    B, d, m = g.shape
    if _needs_grad(y0, f, g):
        base = (y0 + float(dt) * f) if ito else y0
        return base.unsqueeze(0) + g.permute(2, 0, 1) * float(sqrt_dt)
    y0 = _native.contiguous(y0)
    f, = _prep(y0, f)
    g = _native.contiguous(g if g.dtype == y0.dtype else g.to(y0.dtype))
    out = torch.empty((m, B, d), dtype=y0.dtype, device=y0.device)
    lib, dt_code, stream = _launch_env(y0)
    code = lib.tsde_milstein_gf_general_support(out.data_ptr(), y0.data_ptr(), f.data_ptr(), g.data_ptr(), B, d, m,
                                                float(dt), float(sqrt_dt), 1 if ito else 0, dt_code, stream)
    _native.check(code, "tsde_milstein_gf_general_support")
    return out


def milstein_gf_general_correction(g, gk, integrals, sqrt_dt):
    """corr[b, i] = (sum_{k,l} (gk[k, b, i, l] - g[b, i, l]) * I[b, k, l]) / sqrt_dt;  g (B, d, m), gk (m, B, d, m)."""
    B, d, m = g.shape
    if _needs_grad(g, gk, integrals):
        return torch.einsum("kbil,bkl->bi", gk - g.unsqueeze(0), integrals) / float(sqrt_dt)
    g = _native.contiguous(g)
    gk = _native.contiguous(gk if gk.dtype == g.dtype else gk.to(g.dtype))
    integrals = _native.contiguous(integrals if integrals.dtype == g.dtype else integrals.to(g.dtype))
    out = torch.empty((B, d), dtype=g.dtype, device=g.device)
    lib, dt_code, stream = _launch_env(g)
    code = lib.tsde_milstein_gf_general_correction(out.data_ptr(), g.data_ptr(), gk.data_ptr(), integrals.data_ptr(),
                                                   B, d, m, float(sqrt_dt), dt_code, stream)
    _native.check(code, "tsde_milstein_gf_general_correction")
    return out


# ---- SRK ---------------------------------------------------------------------------------------------------
# SRID2 tableau (tableaus/srid2.py:19-54), the Python twin of csrc/tsde_schemes.h `Srid2`
_SRID2_A0 = ((), (1,), (1 / 4, 1 / 4), (0, 0, 0))
_SRID2_A1 = ((), (1 / 4,), (1, 0), (0, 0, 1 / 4))
_SRID2_B0 = ((), (0,), (1, 1 / 2), (0, 0, 0))
_SRID2_B1 = ((), (-1 / 2,), (1, 0), (2, -1, 1 / 2))
_SRID2_ALPHA = (1 / 6, 1 / 6, 2 / 3, 0)
_SRID2_BETA = ((-1, 4 / 3, 2 / 3, 0), (1, -4 / 3, 1 / 3, 0), (2, -4 / 3, -2 / 3, 0), (-2, 5 / 3, -2 / 3, 1))


_SRK_N_IN, _SRK_N_OUT = (0, 3, 5, 4, 2), (0, 3, 3, 2, 1)


def _srk_diag_stage_autograd(stage, ins, dt, rdt, sqrt_dt, noise):
    """The stage arithmetic of `tsde_srk_diag_stage` as differentiable torch ops on the re-materialised (W, U), for
    back-propagation THROUGH the solver (same operation order as the kernel)."""
    W, U = noise.materialise(need_U=True)
    if noise.bcast_d:
        W, U = W.reshape(-1, 1), U.reshape(-1, 1)
    dt, rdt, sqrt_dt = float(dt), float(rdt), float(sqrt_dt)
    A0, A1, B0, B1 = _SRID2_A0, _SRID2_A1, _SRID2_B0, _SRID2_B1

    def h0_term(h, s, j, f, g):
        return (h + (A0[s][j] * f) * dt) + ((B0[s][j] * g) * U) * rdt

    def h1_term(h, s, j, f, g):
        return (h + (A1[s][j] * f) * dt) + (B1[s][j] * g) * sqrt_dt

    def final_term(acc, s, f, g):
        Ikk = (W * W - dt) * 0.5
        Ikkk = ((W * W) * W - (3 * dt) * W) * (1.0 / 6)
        gw = (((_SRID2_BETA[0][s] * W) + (_SRID2_BETA[1][s] * Ikk) / sqrt_dt) + (_SRID2_BETA[2][s] * U) * rdt) + \
             (_SRID2_BETA[3][s] * Ikkk) * rdt
        drift = (_SRID2_ALPHA[s] * f) * dt if s < 3 else 0.0
        return (acc + drift) + g * gw

    if stage == 1:
        y0, f0, g0 = ins
        return h0_term(y0, 1, 0, f0, g0), h1_term(y0, 1, 0, f0, g0), h1_term(y0, 2, 0, f0, g0)
    if stage == 2:
        y0, f0, g0, f1, g1 = ins
        zero = torch.zeros_like(y0)
        h02 = h0_term(h0_term(y0, 2, 0, f0, g0), 2, 1, f1, g1)
        acc = final_term(final_term(y0, 0, f0, g0), 1, f1, g1)
        p13 = h1_term(h1_term(y0, 3, 0, zero, g0), 3, 1, zero, g1)
        return h02, acc, p13
    if stage == 3:
        p13, acc, f2, g2 = ins
        return h1_term(p13, 3, 2, f2, g2), final_term(acc, 2, f2, g2)
    acc, g3 = ins
    return (final_term(acc, 3, None, g3),)


def srk_diag_stage(stage, ins, dt, rdt, sqrt_dt, noise, out_last=None):
    """One SRID2 stage kernel (include/torchsde_amd.h: 1: y0,f0,g0 -> H0_1,H1_1,H1_2; 2: y0,f0,g0,f1,g1 -> H0_2,acc,P;
    3: P,acc,f2,g2 -> H1_3,acc (updated in place); 4: acc,g3 -> y1). Returns the tuple of outputs; `out_last`: where
    stage 4 writes y1."""
    if _needs_grad(*ins):
        return _srk_diag_stage_autograd(stage, ins, dt, rdt, sqrt_dt, noise)
    ref = _native.contiguous(ins[0])
    ins = [ref] + _prep(ref, *ins[1:])
    n_out = _SRK_N_OUT[stage]
    if stage == 3:
        outs = [torch.empty_like(ref), ins[1]]          # acc is advanced in place: one buffer for s = 0..2
    elif stage == 4:
        outs = [_new_like(ref, out_last)]
    else:
        outs = [torch.empty_like(ref) for _ in range(n_out)]
    lib, dt_code, stream = _launch_env(ref)
    in_arr = _native._PTR5(*([t.data_ptr() for t in ins] + [None] * (5 - len(ins))))
    out_arr = _native._PTR3(*([t.data_ptr() for t in outs] + [None] * (3 - n_out)))
    code = lib.tsde_srk_diag_stage(stage, out_arr, in_arr, ref.numel(), float(dt), float(rdt), float(sqrt_dt),
                                   noise.struct(), dt_code, stream)
    _native.check(code, "tsde_srk_diag_stage")
    return tuple(outs)


# ---- adjoint / output --------------------------------------------------------------------------------------
def aug_update(segments, cF, cG, dtype, device):
    """segments: list of dicts(out, s, F, G, D, sF, sG, sD); one fused update per segment."""
    lib = _native.load()
    arr = (Seg * len(segments))()
    keep = []
    for i, sg in enumerate(segments):
        s = sg["s"]
        _native.require_device(s)
        terms = []
        for name in ("F", "G", "D"):
            t = sg.get(name)
            if t is not None:
                t = t.detach()
                if t.dtype != s.dtype:
                    t = t.to(s.dtype)
                t = _native.contiguous(t.reshape(s.shape) if t.numel() == s.numel() else t.expand(s.shape))
            terms.append(t)
        keep.append(terms)
        arr[i].out = sg["out"].data_ptr()
        arr[i].s = s.data_ptr()
        arr[i].F = None if terms[0] is None else terms[0].data_ptr()
        arr[i].G = None if terms[1] is None else terms[1].data_ptr()
        arr[i].D = None if terms[2] is None else terms[2].data_ptr()
        arr[i].n = s.numel()
        arr[i].sF, arr[i].sG, arr[i].sD = sg.get("sF", 1.0), sg.get("sG", 1.0), sg.get("sD", 1.0)
    code = lib.tsde_aug_update(arr, len(segments), float(cF), float(cG), _native.dtype_code(dtype),
                               _native.stream_ptr(device))
    _native.check(code, "tsde_aug_update")


def linear_interp(ya, yb, w0, w1, out=None):
    """out = w0*ya + w1*yb (interp.py:17)."""
    if _needs_grad(ya, yb):
        return w0 * ya + w1 * yb
    ya = _native.contiguous(ya)
    yb, = _prep(ya, yb)
    out = _new_like(ya, out)
    lib, dt_code, stream = _launch_env(ya)
    code = lib.tsde_linear_interp(_native.ptr(out), _native.ptr(ya), _native.ptr(yb), ya.numel(), float(w0),
                                  float(w1), dt_code, stream)
    _native.check(code, "tsde_linear_interp")
    return out


def merge_halves(W, U, Wa, Ha, Wb, Hb, ha=0.0, hb=0.0, ctl=None):
    """(W, U) of a whole step from the (W, H) of its halves, by the generator's own concatenation rule
    (brownian_interval.py:647-672); widths from the host (ha, hb) or from an adaptive controller's device table."""
    lib, dt_code, stream = _launch_env(W)
    code = lib.tsde_merge_halves(W.data_ptr(), _native.ptr(U), Wa.data_ptr(), _native.ptr(Ha), Wb.data_ptr(),
                                 _native.ptr(Hb), W.numel(), None if ctl is None else ctl.data_ptr(), float(ha), float(hb),
                                 dt_code, stream)
    _native.check(code, "tsde_merge_halves")
    return W, U


_ERROR_NORM_SCRATCH = {}


def error_norm_scratch(device):
    """A workspace + result buffer for `error_norm` (callers that replay their launches keep their own)."""
    return torch.empty(_native.ERROR_NORM_WORKSPACE + 1, dtype=torch.float64, device=device)


def error_norm(y_full, y_half, rtol, atol, eps=1e-7, scratch=None):
    """Scaled RMS difference of a full step and two half steps (adaptive_stepping.py:42-76) as a 0-d float64
    device tensor: one fused, deterministic reduction (``tsde_error_norm``). The caller reads it with ``.item()``."""
    y_full = _native.contiguous(y_full.detach())
    y_half, = _prep(y_full, y_half.detach())
    lib, dt_code, stream = _launch_env(y_full)
    dev = y_full.device
    buf = scratch if scratch is not None else _ERROR_NORM_SCRATCH.get((dev, stream))   # one scratch per stream:
    if buf is None:                                                                    # launches on a stream are ordered
        buf = torch.empty(_native.ERROR_NORM_WORKSPACE + 1, dtype=torch.float64, device=dev)
        _ERROR_NORM_SCRATCH[(dev, stream)] = buf
    out = buf[_native.ERROR_NORM_WORKSPACE:]
    code = lib.tsde_error_norm(out.data_ptr(), buf.data_ptr(), y_full.data_ptr(), y_half.data_ptr(), y_full.numel(),
                               float(rtol), float(atol), float(eps), dt_code, stream)
    _native.check(code, "tsde_error_norm")
    return out[0]


class TrajectorySchedule:
    """Device-resident ``tsde_traj_t`` of one solve: step rows, Brownian cells and the output map."""

    def __init__(self, step_rows, cells, out_step, out_w, device, dtype):
        np_dtype = np.float32 if dtype == torch.float32 else np.float64
        rows = np.ascontiguousarray(step_rows, dtype=np_dtype)
        assert rows.ndim == 2 and rows.shape[1] == 8
        self.n_steps, self.n_out = rows.shape[0], len(out_step)
        self.dtype = dtype
        self.host_rows, self.host_cells = rows, np.ascontiguousarray(cells, dtype=np.uint32)
        self.device = device
        self.rows = torch.from_numpy(rows).to(device)
        self.cells = torch.from_numpy(self.host_cells.view(np.int32)).to(device)
        self.out_step = torch.from_numpy(np.ascontiguousarray(out_step, dtype=np.int32)).to(device)
        self.out_w = torch.from_numpy(np.ascontiguousarray(out_w, dtype=np_dtype).reshape(-1, 2)).to(device)
        s = _native.Traj()
        s.step_rows, s.cells = self.rows.data_ptr(), self.cells.data_ptr()
        s.out_step, s.out_w = self.out_step.data_ptr(), self.out_w.data_ptr()
        s.n_steps, s.n_out = self.n_steps, self.n_out
        self._struct = s

    def struct(self):
        return ctypes.byref(self._struct)

    _recent = collections.OrderedDict()      # (host contents, device, dtype) -> schedule; device tensors are read-only

    @classmethod
    def cached(cls, step_rows, cells, out_step, out_w, device, dtype):
        """The schedule of a solve the process has seen before (same steps, cells and outputs -- every iteration of a
        training loop) without its four blocking host->device copies; otherwise a new one, remembered (the 64 most
        recent)."""
        key = (np.ascontiguousarray(step_rows).tobytes(), np.ascontiguousarray(cells).tobytes(),
               tuple(int(k) for k in out_step), tuple((float(a), float(b)) for a, b in out_w), str(device), dtype)
        hit = cls._recent.get(key)
        if hit is not None:
            cls._recent.move_to_end(key)
            return hit
        made = cls(step_rows, cells, out_step, out_w, device, dtype)
        cls._recent[key] = made
        while len(cls._recent) > 64:
            cls._recent.popitem(last=False)
        return made

    def window(self, k_lo, k_hi):
        """The schedule of steps k_lo .. k_hi - 1 alone, with one output per step (re-running a stretch of a solve)."""
        n = k_hi - k_lo
        return TrajectorySchedule.cached(self.host_rows[k_lo:k_hi], self.host_cells[k_lo:k_hi], range(1, n + 1),
                                         [(0.0, 1.0)] * n, self.device, self.dtype)


def _coefficient_tables(coefs, d, schedule, dtype, method):
    """False: every coefficient is a contiguous (d,) tensor (the same at every step). True: every one is a contiguous
    (n_steps * S, d) table -- one row per stage time of every step, for drift / diffusion coefficients that depend on t
    (the `_timed` entry points; S = 1 Euler / Milstein, 2 midpoint, 4 SRK). Anything else is an error."""
    shapes = {tuple(c.shape) for c in coefs}
    if any(c.dtype != dtype or not c.is_contiguous() for c in coefs):
        raise ValueError("coefficients must be contiguous tensors in the state dtype")
    if shapes == {(d,)}:
        return False
    slots = {0: 1, 1: 1, 2: 1, 3: 2, 4: 4, 5: 2, 6: 2}[int(method)]       # stage times per step (csrc/trajectory.hip stage_slots)
    if shapes == {(schedule.n_steps * slots, d)}:
        return True
    raise ValueError(f"coefficients must all be (d,) tensors or all (n_steps * {slots}, d) tables, got {sorted(shapes)}")


def trajectory_affine_diag(ys, y0, drift_rate, drift_shift, diff_rate, diff_shift, method, schedule, bm, sens=None):
    """All steps of an affine diagonal SDE in one launch (``tsde_trajectory_affine_diag``); writes ys[j] for the
    schedule's outputs. `bm` is the native BrownianInterval whose generated cells drive the steps. With
    `sens` (n_out, 5, rows, d) the path-wise sensitivities d ys / d (y0, the four coefficients) are written too
    (``tsde_trajectory_affine_diag_sens``)."""
    _native.require_device(ys, y0, drift_rate, drift_shift, diff_rate, diff_shift, sens)
    rows, d = y0.shape
    coefs = (drift_rate, drift_shift, diff_rate, diff_shift)
    timed = _coefficient_tables(coefs, d, schedule, y0.dtype, method)
    if timed and sens is not None:
        raise ValueError("per-step coefficient tables: values only")
    if schedule.dtype != y0.dtype or ys.dtype != y0.dtype:
        raise ValueError("schedule / output dtype must equal the state dtype")
    if not (ys.is_contiguous() and y0.is_contiguous()) or ys.shape != (schedule.n_out, rows, d):
        raise ValueError("ys must be a contiguous (n_out, rows, d) tensor and y0 contiguous")
    lib, dt_code, stream = _launch_env(y0)
    entropy_dev = bm._entropy_dev
    tail = (rows, d, drift_rate.data_ptr(), drift_shift.data_ptr(), diff_rate.data_ptr(), diff_shift.data_ptr(),
            int(method), schedule.struct(), bm._key, bm._elem0, None if entropy_dev is None else entropy_dev.data_ptr(),
            dt_code, stream)
    if timed:
        code = lib.tsde_trajectory_affine_diag_timed(ys.data_ptr(), y0.data_ptr(), *tail[:6], d, *tail[6:])
    elif sens is None:
        code = lib.tsde_trajectory_affine_diag(ys.data_ptr(), y0.data_ptr(), *tail)
    else:
        if (sens.dtype != y0.dtype or not sens.is_contiguous()
                or sens.shape != (schedule.n_out, _native.TRAJ_SENS, rows, d)):
            raise ValueError("sens must be a contiguous (n_out, 5, rows, d) tensor in the state dtype")
        code = lib.tsde_trajectory_affine_diag_sens(ys.data_ptr(), sens.data_ptr(), y0.data_ptr(), *tail)
    _native.check(code, "tsde_trajectory_affine_diag")
    return ys


def trajectory_expr_diag(ys, y0, f_kind, g_kind, coefs, method, schedule, bm):
    """All steps of a diagonal SDE with elementwise-expression drift and diffusion in one launch
    (``tsde_trajectory_expr_diag``); writes ys[j] for the schedule's outputs."""
    _native.require_device(ys, y0, *coefs)
    rows, d = y0.shape
    if len(coefs) != 8:
        raise ValueError("eight coefficient tensors are needed")
    timed = _coefficient_tables(coefs, d, schedule, y0.dtype, method)
    if schedule.dtype != y0.dtype or ys.dtype != y0.dtype:
        raise ValueError("schedule / output dtype must equal the state dtype")
    if not (ys.is_contiguous() and y0.is_contiguous()) or ys.shape != (schedule.n_out, rows, d):
        raise ValueError("ys must be a contiguous (n_out, rows, d) tensor and y0 contiguous")
    lib, dt_code, stream = _launch_env(y0)
    entropy_dev = bm._entropy_dev
    arr = (ctypes.c_void_p * 8)(*[c.data_ptr() for c in coefs])
    code = lib.tsde_trajectory_expr_diag_timed(ys.data_ptr(), y0.data_ptr(), rows, d, arr, d if timed else 0, int(f_kind),
                                               int(g_kind), int(method), schedule.struct(), bm._key, bm._elem0,
                                               None if entropy_dev is None else entropy_dev.data_ptr(), dt_code, stream)
    _native.check(code, "tsde_trajectory_expr_diag")
    return ys


def trajectory_prog_diag(ys, y0, f_code, g_code, dg_code, consts, scalar_noise, method, schedule, bm):
    """All steps of a diagonal- or scalar-noise SDE whose drift and diffusion are expression programs (tuples of
    instruction words, recognise.RecognisedProgram) in one launch (``tsde_trajectory_prog_diag``)."""
    _native.require_device(ys, y0, consts)
    rows, d = y0.shape
    if schedule.dtype != y0.dtype or ys.dtype != y0.dtype or consts.dtype != y0.dtype:
        raise ValueError("schedule / output / constant dtype must equal the state dtype")
    if not (ys.is_contiguous() and y0.is_contiguous() and consts.is_contiguous()) or ys.shape != (schedule.n_out, rows, d) \
            or consts.dim() != 2 or consts.shape[1] != d:
        raise ValueError("ys must be a contiguous (n_out, rows, d) tensor, y0 contiguous, consts (n_const, d)")
    words = tuple(f_code) + tuple(g_code) + tuple(dg_code)
    lib, dt_code, stream = _launch_env(y0)
    # the same programs as straight-line code, compiled at run time (specialise.py): used once the library is there AND its
    # first launch has reproduced the interpreter's result bit for bit
    from . import specialise
    key, compiled = specialise.lookup(f_code, g_code, dg_code, consts.shape[0], y0.dtype, method, y0.device)
    if compiled is not None and specialise.verified(key):
        specialise.launch(compiled, ys, y0, consts, scalar_noise, schedule, bm, stream)
        return ys
    code = (ctypes.c_uint32 * len(words))(*words)            # a host array: the words travel in the kernel arguments
    entropy_dev = bm._entropy_dev
    rc = lib.tsde_trajectory_prog_diag(ys.data_ptr(), y0.data_ptr(), rows, d, code, len(f_code), len(g_code),
                                       len(dg_code), consts.data_ptr(), consts.shape[0], int(bool(scalar_noise)), int(method),
                                       schedule.struct(), bm._key, bm._elem0,
                                       None if entropy_dev is None else entropy_dev.data_ptr(), dt_code, stream)
    _native.check(rc, "tsde_trajectory_prog_diag")
    if compiled is not None and specialise.verified(key) is None and not torch.cuda.is_current_stream_capturing():
        other = torch.empty_like(ys)
        specialise.launch(compiled, other, y0, consts, scalar_noise, schedule, bm, stream)
        same = ((other == ys) | (other.isnan() & ys.isnan())).all()
        specialise.set_verified(key, bool(same))             # (one synchronisation per program, ever)
    return ys


def trajectory_rows(ys, y0, structure, consts, method, schedule, bm):
    """All steps of a small ROW-COUPLED diagonal-noise system (recognise_rows.RecognisedRows) in one launch: the program
    kernel with one lane per row and the system's generated model (specialise.source_rows). Returns `ys`, or None while the
    generated unit is not compiled yet (there is no interpreter for such systems: the caller stays stepwise)."""
    from . import specialise
    _native.require_device(ys, y0, consts)
    rows, d = y0.shape
    if ys.dtype != y0.dtype or consts.dtype != y0.dtype or schedule.dtype != y0.dtype:
        raise ValueError("schedule / output / constant dtype must equal the state dtype")
    if not (ys.is_contiguous() and y0.is_contiguous() and consts.is_contiguous()) or ys.shape != (schedule.n_out, rows, d):
        raise ValueError("ys must be a contiguous (n_out, rows, d) tensor, y0 and consts contiguous")
    key, compiled = specialise.lookup_rows(structure, structure[1][1], y0.dtype, method, y0.device)
    if compiled is None:
        return None
    _, _, stream = _launch_env(y0)
    specialise.launch_rows(compiled, ys, y0, consts, schedule, bm, stream)
    return ys


def trajectory_prog_additive(ys, y0, f_code, consts, g_table, m, method, schedule, bm):
    """All steps of an additive-noise SDE in one launch (``tsde_trajectory_prog_additive``): the drift an expression program,
    the diffusion the table `g_table` -- (m, d), the transpose of the one matrix g, or (n_steps, slots, m, d) with the matrices
    at the scheme's stage times (recognise.RecognisedAdditive)."""
    _native.require_device(ys, y0, consts, g_table)
    rows, d = y0.shape
    if any(t.dtype != y0.dtype for t in (ys, consts, g_table)) or schedule.dtype != y0.dtype:
        raise ValueError("schedule / output / constant / table dtype must equal the state dtype")
    if not (ys.is_contiguous() and y0.is_contiguous() and consts.is_contiguous() and g_table.is_contiguous()) \
            or ys.shape != (schedule.n_out, rows, d) or consts.dim() != 2 or consts.shape[1] != d:
        raise ValueError("ys must be a contiguous (n_out, rows, d) tensor, y0 contiguous, consts (n_const, d)")
    slots = 1 if int(method) == _native.TRAJ_EULER else 2
    timed = g_table.dim() == 4
    if tuple(g_table.shape) != ((schedule.n_steps, slots, m, d) if timed else (m, d)):
        raise ValueError(f"g_table must be (m, d) or (n_steps, {slots}, m, d), got {tuple(g_table.shape)}")
    lib, dt_code, stream = _launch_env(y0)
    # the drift program as straight-line code, compiled at run time and verified bit for bit on first use (specialise.py)
    from . import specialise
    kind = "additive4" if m <= 4 else "additive8" if m <= 8 else "additive16"
    key, compiled = specialise.lookup(f_code, (), (), consts.shape[0], y0.dtype, method, y0.device, kind=kind)
    if compiled is not None and specialise.verified(key):
        specialise.launch_additive(compiled, ys, y0, consts, g_table, m, timed, schedule, bm, stream)
        return ys
    code = (ctypes.c_uint32 * len(f_code))(*f_code)
    entropy_dev = bm._entropy_dev
    rc = lib.tsde_trajectory_prog_additive(ys.data_ptr(), y0.data_ptr(), rows, d, int(m), code, len(f_code), consts.data_ptr(),
                                           consts.shape[0], g_table.data_ptr(), int(timed), int(method), schedule.struct(),
                                           bm._key, bm._elem0, None if entropy_dev is None else entropy_dev.data_ptr(),
                                           dt_code, stream)
    _native.check(rc, "tsde_trajectory_prog_additive")
    if compiled is not None and specialise.verified(key) is None and not torch.cuda.is_current_stream_capturing():
        other = torch.empty_like(ys)
        specialise.launch_additive(compiled, other, y0, consts, g_table, m, timed, schedule, bm, stream)
        specialise.set_verified(key, bool(((other == ys) | (other.isnan() & ys.isnan())).all()))
    return ys


class _ProgTrajectoryFn(torch.autograd.Function):
    """Differentiable whole-trajectory solve of an SDE stated as expression programs: the forward launch also produces the
    path-wise sensitivities of every output element with respect to y0 and to up to four constant rows (the per-channel
    parameters of the user's module), and the backward pass is a handful of torch reductions of cotangent x sensitivity --
    the gradient back-propagation through the stepwise solver gives, without storing or revisiting a step."""

    @staticmethod
    def forward(ctx, programs, scalar_noise, method, schedule, bm, const_values, param_rows, y0, *params):
        rows, d = y0.shape
        f_code, g_code, dg_code = programs
        y0c = _native.contiguous(y0.detach())
        table = []
        for k, c in enumerate(const_values):
            c = params[param_rows.index(k)] if k in param_rows else c
            if torch.is_tensor(c):
                table.append(c.detach().to(device=y0.device, dtype=y0.dtype).reshape(-1).expand(d))
            else:
                table.append(torch.full((d,), float(c), dtype=y0.dtype, device=y0.device))
        consts = torch.stack(table).contiguous() if table else torch.zeros(1, d, dtype=y0.dtype, device=y0.device)
        slots = (ctypes.c_int8 * max(len(const_values), 1))(*[(param_rows.index(k) + 1 if k in param_rows else -1)
                                                              for k in range(len(const_values))])
        ys = torch.empty((schedule.n_out + 1, rows, d), dtype=y0.dtype, device=y0.device)
        sens = torch.empty((schedule.n_out, _native.TRAJ_SENS, rows, d), dtype=y0.dtype, device=y0.device)
        ys[0].copy_(y0c)
        words = tuple(f_code) + tuple(g_code) + tuple(dg_code)
        lib, dt_code, stream = _launch_env(y0c)
        entropy_dev = bm._entropy_dev
        # the same programs on dual numbers as straight-line code, compiled at run time and verified bit for bit on first use
        from . import specialise
        key, compiled = specialise.lookup(f_code, g_code, dg_code, len(const_values), y0.dtype, method, y0.device, kind="sens")
        if compiled is not None and specialise.verified(key):
            specialise.launch_sens(compiled, ys[1:], sens, slots, y0c, consts, len(const_values), scalar_noise, schedule, bm, stream)
        else:
            code = (ctypes.c_uint32 * len(words))(*words)
            rc = lib.tsde_trajectory_prog_diag_sens(
                ys[1:].data_ptr(), sens.data_ptr(), y0c.data_ptr(), rows, d, code, len(f_code), len(g_code), len(dg_code),
                consts.data_ptr(), len(const_values), slots, int(bool(scalar_noise)), int(method), schedule.struct(), bm._key,
                bm._elem0, None if entropy_dev is None else entropy_dev.data_ptr(), dt_code, stream)
            _native.check(rc, "tsde_trajectory_prog_diag_sens")
            if compiled is not None and specialise.verified(key) is None:
                ys2, sens2 = torch.empty_like(ys[1:]), torch.empty_like(sens)
                specialise.launch_sens(compiled, ys2, sens2, slots, y0c, consts, len(const_values), scalar_noise, schedule, bm,
                                       stream)
                same = (((ys2 == ys[1:]) | (ys2.isnan() & ys[1:].isnan())).all()
                        & ((sens2 == sens) | (sens2.isnan() & sens.isnan())).all())
                specialise.set_verified(key, bool(same))
        ctx.save_for_backward(sens)
        ctx.param_shapes = [tuple(p.shape) for p in params]
        return ys

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, gys):
        sens, = ctx.saved_tensors
        weighted = (gys[1:].unsqueeze(1) * sens).sum(dim=0)           # (TRAJ_SENS, rows, d)
        grad_y0 = gys[0] + weighted[0] if ctx.needs_input_grad[7] else None
        grads = []
        for k, shape in enumerate(ctx.param_shapes):
            if not ctx.needs_input_grad[8 + k]:
                grads.append(None)
                continue
            per_channel = weighted[1 + k].sum(dim=0)                  # (d,): the batch shares the constants
            numel = 1
            for n in shape:
                numel *= n
            grads.append(per_channel.reshape(shape) if numel == per_channel.numel() else per_channel.sum().reshape(shape))
        return (None,) * 7 + (grad_y0,) + tuple(grads)


def trajectory_prog_diag_differentiable(y0, programs, const_values, param_rows, scalar_noise, method, schedule, bm):
    """ys (n_out + 1, rows, d) with a grad_fn towards y0 and the constants `const_values[k]`, k in `param_rows` (at most
    four tensors of one element or d elements each), through ``tsde_trajectory_prog_diag_sens``."""
    params = [const_values[k] for k in param_rows]
    return _ProgTrajectoryFn.apply(tuple(programs), bool(scalar_noise), int(method), schedule, bm, tuple(const_values),
                                   tuple(param_rows), y0, *params)


def trajectory_mlp_diag(ys, y0, w1, b1, w2, b2, diff_rate, diff_shift, activation, diffusion, method, schedule, bm):
    """All steps of a diagonal SDE with a two-layer perceptron drift in one launch (``tsde_trajectory_mlp_diag``);
    writes ys[j] for the schedule's outputs (an output time inside a step is interpolated in the kernel with the
    schedule's weights, _core/interp.py:15-18)."""
    tensors = (ys, y0, w1, b1, w2, b2, diff_rate, diff_shift)
    _native.require_device(*tensors)
    rows, d = y0.shape
    hidden = b1.numel()
    if any(t.dtype != torch.float32 or not t.is_contiguous() for t in tensors):
        raise ValueError("the perceptron-drift kernel takes contiguous float32 tensors")
    if w1.shape != (d, hidden) or w2.shape != (hidden, d) or b2.numel() != d or ys.shape != (schedule.n_out, rows, d):
        raise ValueError("shape mismatch: w1 (d, hidden), w2 (hidden, d), ys (n_out, rows, d)")
    lib, dt_code, stream = _launch_env(y0)
    entropy_dev = bm._entropy_dev
    code = lib.tsde_trajectory_mlp_diag(
        ys.data_ptr(), y0.data_ptr(), rows, d, hidden, w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
        diff_rate.data_ptr(), diff_shift.data_ptr(), int(diffusion[0]), float(diffusion[1]), int(activation),
        int(method), schedule.struct(), bm._key,
        bm._elem0, None if entropy_dev is None else entropy_dev.data_ptr(), dt_code, stream)
    _native.check(code, "tsde_trajectory_mlp_diag")
    return ys


class NeuralNet:
    """One perceptron of ``tsde_trajectory_mlp_general`` (``tsde_mlp_t``): out = scale * final(W2 . act(W1 . y + w1t * t + b1)
    + b2) with `w1` (in, hidden) and `w2` (hidden, out) input-major, `w1t` (hidden) the weight column of the time input or
    None. Holds contiguous float32 device tensors (and keeps them alive for the launch)."""

    def __init__(self, w1, w1t, b1, w2, b2, activation, final=_native.FINAL_NONE, scale=1.0, precision=_native.PRECISION_F32):
        self.tensors = [None if t is None else _native.contiguous(t.detach()) for t in (w1, w1t, b1, w2, b2)]
        self.activation, self.final, self.scale, self.precision = int(activation), int(final), float(scale), int(precision)
        self.hidden, self.out = int(self.tensors[2].numel()), int(self.tensors[4].numel())

    def struct(self):
        w1, w1t, b1, w2, b2 = self.tensors
        return _native.Mlp(w1.data_ptr(), 0 if w1t is None else w1t.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                           self.hidden, self.out, self.activation, self.final, self.scale, self.precision, 0)

    def __eq__(self, other):
        return isinstance(other, NeuralNet) and (self.activation, self.final, self.scale, self.precision) == (
            other.activation, other.final, other.scale, other.precision) and all(
            (a is None and b is None) or (a is not None and b is not None and a.shape == b.shape and torch.equal(a, b))
            for a, b in zip(self.tensors, other.tensors))

    __hash__ = None


def mlp_general_lds(d, m, drift_hidden, diffusion_hidden, diffusion_out, noise):
    """Bytes of LDS ``tsde_trajectory_mlp_general`` needs for a shape (0: no kernel covers it)."""
    return int(_native.load().tsde_trajectory_mlp_general_lds(d, m, drift_hidden, diffusion_hidden, diffusion_out, noise))


def trajectory_mlp_general(ys, y0, drift, diffusion, noise, m, method, schedule, bm):
    """All steps of a neural SDE (drift and diffusion two-layer perceptrons of (t, y)) in one launch
    (``tsde_trajectory_mlp_general``); slot 7 of the schedule's step rows must hold each step's start time
    (`BaseSDESolver._integrate_trajectory` fills it)."""
    _native.require_device(ys, y0, *[t for net in (drift, diffusion) for t in net.tensors])
    rows, d = y0.shape
    tensors = [ys, y0] + [t for net in (drift, diffusion) for t in net.tensors if t is not None]
    if any(t.dtype != torch.float32 or not t.is_contiguous() for t in tensors):
        raise ValueError("the neural-SDE kernel takes contiguous float32 tensors")
    if ys.shape != (schedule.n_out, rows, d):
        raise ValueError("shape mismatch: ys (n_out, rows, d)")
    for net in (drift, diffusion):
        w1, w1t, b1, w2, b2 = net.tensors
        if w1.shape != (d, net.hidden) or w2.shape != (net.hidden, net.out) or (w1t is not None and w1t.numel() != net.hidden):
            raise ValueError("shape mismatch: w1 (d, hidden), w1t (hidden), w2 (hidden, out)")
    lib, dt_code, stream = _launch_env(y0)
    entropy_dev = bm._entropy_dev
    code = lib.tsde_trajectory_mlp_general(
        ys.data_ptr(), y0.data_ptr(), rows, d, int(m), int(noise), ctypes.byref(drift.struct()),
        ctypes.byref(diffusion.struct()), int(method), schedule.struct(), bm._key, bm._elem0,
        None if entropy_dev is None else entropy_dev.data_ptr(), dt_code, stream)
    _native.check(code, "tsde_trajectory_mlp_general")
    return ys


def trajectory_mlp_additive(ys, y0, drift, g_table, m, method, schedule, bm):
    """All steps of an additive-noise SDE whose drift is a two-layer perceptron of (t, y) in one launch
    (``tsde_trajectory_mlp_additive``); `g_table` as for `trajectory_prog_additive`."""
    tensors = [ys, y0, g_table] + [t for t in drift.tensors if t is not None]
    _native.require_device(*tensors)
    rows, d = y0.shape
    if any(t.dtype != torch.float32 or not t.is_contiguous() for t in tensors):
        raise ValueError("the neural-SDE kernel takes contiguous float32 tensors")
    if ys.shape != (schedule.n_out, rows, d):
        raise ValueError("shape mismatch: ys (n_out, rows, d)")
    slots = 1 if int(method) == _native.TRAJ_EULER else 2
    timed = g_table.dim() == 4
    if tuple(g_table.shape) != ((schedule.n_steps, slots, m, d) if timed else (m, d)):
        raise ValueError(f"g_table must be (m, d) or (n_steps, {slots}, m, d), got {tuple(g_table.shape)}")
    lib, dt_code, stream = _launch_env(y0)
    entropy_dev = bm._entropy_dev
    code = lib.tsde_trajectory_mlp_additive(
        ys.data_ptr(), y0.data_ptr(), rows, d, int(m), ctypes.byref(drift.struct()), g_table.data_ptr(), int(timed),
        int(method), schedule.struct(), bm._key, bm._elem0, None if entropy_dev is None else entropy_dev.data_ptr(),
        dt_code, stream)
    _native.check(code, "tsde_trajectory_mlp_additive")
    return ys


class _TrajectoryFn(torch.autograd.Function):
    """Differentiable whole-trajectory solve of an affine diagonal SDE: the forward launch also produces the
    path-wise sensitivities of every output element (forward-mode tangents carried in registers), and the backward
    pass is a handful of torch reductions of cotangent x sensitivity -- the gradient back-propagation through the
    stepwise solver would give, without storing or revisiting a single step."""

    @staticmethod
    def forward(ctx, method, schedule, bm, y0, *params):
        rows, d = y0.shape
        y0c = _native.contiguous(y0.detach())
        coefs = [p.detach().reshape(-1).expand(d).contiguous() for p in params]
        ys = torch.empty((schedule.n_out + 1, rows, d), dtype=y0.dtype, device=y0.device)
        sens = torch.empty((schedule.n_out, _native.TRAJ_SENS, rows, d), dtype=y0.dtype, device=y0.device)
        ys[0].copy_(y0c)
        trajectory_affine_diag(ys[1:], y0c, *coefs, method, schedule, bm, sens=sens)
        ctx.save_for_backward(sens)
        ctx.param_shapes = [tuple(p.shape) for p in params]
        return ys

    @staticmethod
    @torch.autograd.function.once_differentiable      # kernels, not torch ops: no graph of the backward pass exists
    def backward(ctx, gys):
        sens, = ctx.saved_tensors
        g = gys[1:].unsqueeze(1)                                      # (n_out, 1, rows, d)
        weighted = (g * sens).sum(dim=0)                              # (5, rows, d)
        grad_y0 = gys[0] + weighted[0] if ctx.needs_input_grad[3] else None
        per_channel = weighted[1:].sum(dim=1)                         # (4, d): the batch shares the coefficients
        grads = []
        for k, shape in enumerate(ctx.param_shapes):
            if not ctx.needs_input_grad[4 + k]:
                grads.append(None)
            elif len(shape) == 1 and shape[0] == per_channel.shape[1]:
                grads.append(per_channel[k])
            else:                                                     # scalar coefficient broadcast over channels
                grads.append(per_channel[k].sum().reshape(shape))
        return (None, None, None, grad_y0) + tuple(grads)


def trajectory_affine_diag_differentiable(y0, params, method, schedule, bm):
    """ys (n_out + 1, rows, d) with a grad_fn towards y0 and the four coefficient tensors."""
    return _TrajectoryFn.apply(method, schedule, bm, y0, *params)


def gram(a, b, blocks=512, column_sums=False):
    """a^T b for tall row-major a (k, m), b (k, n) (``tsde_gram_partials`` + a fixed-order sum of its partials): the
    weight-gradient reduction of the perceptron-drift backward sweep. The kernel produces results of up to 128 x 128;
    wider operands are served in place as column blocks (row strides = the full widths). With `column_sums`, also
    a.sum(0) (the bias gradient), from the same pass over a."""
    _native.require_device(a, b)
    if a.dtype != torch.float32 or b.dtype != torch.float32 or not (a.is_contiguous() and b.is_contiguous()):
        raise ValueError("gram takes contiguous float32 matrices")
    if a.dim() != 2 or b.dim() != 2 or a.shape[0] != b.shape[0]:
        raise ValueError("gram takes a (k, m) and b (k, n)")
    k, m, n = a.shape[0], a.shape[1], b.shape[1]
    blocks = int(max(1, min(blocks, (k + 63) // 64)))
    lib, dt_code, stream = _launch_env(a)
    out = torch.empty((m, n), dtype=torch.float32, device=a.device)
    sums = torch.empty(m, dtype=torch.float32, device=a.device) if column_sums else None
    partials = torch.empty((blocks, min(m, 128), min(n, 128)), dtype=torch.float32, device=a.device)
    sum_partials = torch.empty((blocks, min(m, 128)), dtype=torch.float32, device=a.device) if column_sums else None
    for i0 in range(0, m, 128):
        mi = min(128, m - i0)
        for j0 in range(0, n, 128):
            nj = min(128, n - j0)
            want_sums = column_sums and j0 == 0
            p_view = partials.reshape(-1)[:blocks * mi * nj].view(blocks, mi, nj)
            s_view = sum_partials.reshape(-1)[:blocks * mi].view(blocks, mi) if want_sums else None
            _native.check(lib.tsde_gram_partials(p_view.data_ptr(), None if s_view is None else s_view.data_ptr(),
                                                 a.data_ptr() + 4 * i0, m, b.data_ptr() + 4 * j0, n, k, mi, nj, blocks,
                                                 dt_code, stream), "tsde_gram_partials")
            out[i0:i0 + mi, j0:j0 + nj] = p_view.sum(dim=0)
            if want_sums:
                sums[i0:i0 + mi] = s_view.sum(dim=0)
    return (out, sums) if column_sums else out


_boundary_tables = collections.OrderedDict()


def _boundary_table(out_steps, device):
    """Device table [0, *out_steps] (int32) of the step boundaries the outputs sit on; remembered like the schedules."""
    key = (tuple(out_steps), str(device))
    table = _boundary_tables.get(key)
    if table is None:
        table = torch.from_numpy(np.asarray((0,) + tuple(out_steps), dtype=np.int32)).to(device)
        _boundary_tables[key] = table
        while len(_boundary_tables) > 8:
            _boundary_tables.popitem(last=False)
    return table


class _MlpTrajectoryFn(torch.autograd.Function):
    """Differentiable whole-trajectory Euler / Milstein solve of a perceptron-drift diagonal SDE. Forward: the sampling
    kernel, writing the state at EVERY step (HBM is plentiful on this part; the reverse sweep needs them). Backward:
    the reverse sweep kernel over chunks of steps (last first), each followed by the two weight-gradient products over
    that chunk's stash -- the gradient back-propagation through the stepwise solver gives, without an autograd tape.

    When every step's state would not fit the budget (very large batch x steps), the forward launch keeps only the
    states at the chunk boundaries and the backward pass re-runs the sampling kernel over each chunk first (same
    kernel, same increments: the recomputed states are bit-identical), trading one more forward for O(steps / chunk +
    chunk) states instead of O(steps)."""

    # per-chunk stash budget of the reverse sweep (three (steps, rows, width) float32 arrays)
    STASH_BYTES = 3 << 30
    # budget for keeping the state of every step; None = half of the device memory that is free at forward time
    STATE_BYTES = None

    @staticmethod
    def _chunk(n_steps, rows, d, hidden):
        per_step = rows * (d + 2 * hidden) * 4
        return int(max(1, min(n_steps, _MlpTrajectoryFn.STASH_BYTES // max(per_step, 1))))

    @staticmethod
    def forward(ctx, activation, diffusion, method, schedule_all, out_steps, bm, y0, w1, b1, w2, b2, rate, shift):
        rows, d = y0.shape
        hidden, n_steps = b1.numel(), schedule_all.n_steps
        y0c = _native.contiguous(y0.detach())
        coefs = [p.detach().reshape(-1).expand(d).contiguous() for p in (rate, shift)]
        w1_in = w1.detach().t().contiguous()                # (d, hidden): input-major, as the kernels read it
        w2_in = w2.detach().t().contiguous()                # (hidden, d)
        b1c, b2c = b1.detach().contiguous(), b2.detach().contiguous()
        budget = _MlpTrajectoryFn.STATE_BYTES
        if budget is None:
            budget = torch.cuda.mem_get_info(y0.device)[0] // 2
        if (n_steps + 1) * rows * d * 4 <= budget:
            kept_at = None                                  # states[k] is the state at boundary k
            schedule = schedule_all
        else:
            chunk = _MlpTrajectoryFn._chunk(n_steps, rows, d, hidden)
            marks = sorted(set(range(n_steps, 0, -chunk)) | set(out_steps))
            kept_at = {0: 0}
            kept_at.update((k, i + 1) for i, k in enumerate(marks))
            schedule = TrajectorySchedule.cached(schedule_all.host_rows, schedule_all.host_cells, marks,
                                                 [(0.0, 1.0)] * len(marks), y0.device, y0.dtype)
        states = torch.empty((schedule.n_out + 1, rows, d), dtype=y0.dtype, device=y0.device)
        states[0].copy_(y0c)
        trajectory_mlp_diag(states[1:], y0c, w1_in, b1c, w2_in, b2c, coefs[0], coefs[1], activation, diffusion, method,
                            schedule, bm)
        ctx.grad_step = _boundary_table(out_steps, y0.device)
        ctx.save_for_backward(states, w1_in, b1c, w2_in, b2c, coefs[0], coefs[1])
        ctx.method, ctx.activation, ctx.hidden = int(method), int(activation), hidden
        ctx.diffusion = (int(diffusion[0]), float(diffusion[1]))
        ctx.schedule, ctx.bm, ctx.out_steps, ctx.kept_at = schedule_all, bm, out_steps, kept_at
        ctx.param_shapes = (tuple(rate.shape), tuple(shift.shape))
        # (a stack of views, not index_select: building an index tensor is a blocking host->device copy that would wait
        #  for the forward launch)
        return torch.stack([states[k if kept_at is None else kept_at[k]] for k in (0,) + tuple(out_steps)], dim=0)

    @staticmethod
    @torch.autograd.function.once_differentiable      # kernels, not torch ops: no graph of the backward pass exists
    def backward(ctx, gys):
        states, w1_in, b1c, w2_in, b2c, rate, shift = ctx.saved_tensors
        rows, d = states.shape[1], states.shape[2]
        hidden, schedule, kept_at, bm = ctx.hidden, ctx.schedule, ctx.kept_at, ctx.bm
        n_steps = schedule.n_steps
        dev = states.device
        gys = _native.contiguous(gys)
        boundaries = np.asarray([0] + list(ctx.out_steps), dtype=np.int32)
        chunk = _MlpTrajectoryFn._chunk(n_steps, rows, d, hidden)
        stash_lam = torch.empty((chunk, rows, d), dtype=torch.float32, device=dev)
        stash_hid = torch.empty((chunk, rows, hidden), dtype=torch.float32, device=dev)
        stash_delta = torch.empty((chunk, rows, hidden), dtype=torch.float32, device=dev)
        rerun = None if kept_at is None else torch.empty((chunk + 1, rows, d), dtype=torch.float32, device=dev)
        lam = torch.zeros((rows, d), dtype=torch.float32, device=dev)
        row_rate, row_shift = torch.zeros_like(lam), torch.zeros_like(lam)
        g_w1 = torch.zeros((hidden, d), dtype=torch.float32, device=dev)
        g_w2 = torch.zeros((d, hidden), dtype=torch.float32, device=dev)
        g_b1 = torch.zeros(hidden, dtype=torch.float32, device=dev)
        g_b2 = torch.zeros(d, dtype=torch.float32, device=dev)
        lib, dt_code, stream = _launch_env(states)
        entropy_dev = bm._entropy_dev
        for k_hi in range(n_steps, 0, -chunk):
            k_lo = max(0, k_hi - chunk)
            n = k_hi - k_lo
            if kept_at is None:
                ys, ys_first = states, 0
            else:                                   # the states of this chunk again, from the state kept at its start
                rerun[0].copy_(states[kept_at[k_lo]])
                trajectory_mlp_diag(rerun[1:n + 1], rerun[0], w1_in, b1c, w2_in, b2c, rate, shift, ctx.activation,
                                    ctx.diffusion, ctx.method, schedule.window(k_lo, k_hi), bm)
                ys, ys_first = rerun, k_lo
            grad_last = int(np.searchsorted(boundaries, k_hi, side="right")) - 1
            code = lib.tsde_trajectory_mlp_diag_backward(
                lam.data_ptr(), stash_lam.data_ptr(), stash_hid.data_ptr(), stash_delta.data_ptr(), row_rate.data_ptr(),
                row_shift.data_ptr(), ys.data_ptr(), ys_first, gys.data_ptr(), ctx.grad_step.data_ptr(), grad_last,
                rows, d, hidden, w1_in.data_ptr(), b1c.data_ptr(), w2_in.data_ptr(), rate.data_ptr(), shift.data_ptr(),
                ctx.diffusion[0], ctx.diffusion[1], ctx.activation, ctx.method, schedule.struct(), k_lo, k_hi, bm._key,
                bm._elem0,
                None if entropy_dev is None else entropy_dev.data_ptr(), dt_code, stream)
            _native.check(code, "tsde_trajectory_mlp_diag_backward")
            flat_lam = stash_lam[:n].reshape(n * rows, d)
            flat_hid = stash_hid[:n].reshape(n * rows, hidden)
            flat_delta = stash_delta[:n].reshape(n * rows, hidden)
            flat_y = ys[k_lo - ys_first:k_hi - ys_first].reshape(n * rows, d)
            for g_w, g_b, a, b in ((g_w2, g_b2, flat_lam, flat_hid), (g_w1, g_b1, flat_delta, flat_y)):
                weight, bias = gram(a, b, column_sums=True)
                g_w += weight
                g_b += bias
        grad_y0 = lam + gys[0] if ctx.needs_input_grad[6] else None
        diffusion = []
        for acc, shape in zip((row_rate, row_shift), ctx.param_shapes):
            per_channel = acc.sum(dim=0)
            diffusion.append(per_channel.reshape(shape) if int(np.prod(shape, dtype=np.int64)) == d and len(shape) == 1
                             else per_channel.sum().reshape(shape))
        return (None, None, None, None, None, None, grad_y0, g_w1, g_b1, g_w2, g_b2, diffusion[0], diffusion[1])


def trajectory_mlp_diag_differentiable(y0, module_params, activation, diffusion, method, schedule_all, out_steps, bm):
    """ys (len(out_steps) + 1, rows, d) with a grad_fn towards y0 and the six parameters
    (lin1.weight, lin1.bias, lin2.weight, lin2.bias, diff_rate, diff_shift)."""
    return _MlpTrajectoryFn.apply(activation, tuple(diffusion), method, schedule_all, tuple(int(k) for k in out_steps),
                                  bm, y0, *module_params)


# ---- in-library event timing (bench.py's roofline) ----------------------------------------------------------
def prof_begin(kid, capacity):
    _native.check(_native.load().tsde_prof_begin(kid, capacity), "tsde_prof_begin")


def gpu_delay(microseconds, device=None):
    """Keep the current stream busy for ~`microseconds` (lets the host run ahead of the GPU)."""
    _native.check(_native.load().tsde_delay_us(float(microseconds), _native.stream_ptr(device)), "tsde_delay_us")


def prof_bracket_overhead(n=100, spin_us=12.0, device=None):
    """Mean cost (ms) an event bracket adds to the kernel it brackets (calibrated with self-timed spin kernels)."""
    ms = ctypes.c_double(0.0)
    _native.check(_native.load().tsde_prof_bracket_overhead(n, float(spin_us), ctypes.byref(ms),
                                                            _native.stream_ptr(device)),
                  "tsde_prof_bracket_overhead")
    return ms.value


def prof_read(capacity):
    """The per-launch times (ms) of the running profile, in launch order (at most `capacity`)."""
    arr = (ctypes.c_double * capacity)()
    used = ctypes.c_int(0)
    _native.check(_native.load().tsde_prof_read(arr, capacity, ctypes.byref(used)), "tsde_prof_read")
    return list(arr[:used.value])


def prof_end():
    ms, n = ctypes.c_double(0.0), ctypes.c_int64(0)
    _native.check(_native.load().tsde_prof_end(ctypes.byref(ms), ctypes.byref(n)), "tsde_prof_end")
    return ms.value, n.value


# ---- reversible Heun --------------------------------------------------------------------------------------
def _noise_W(noise):
    W, _ = noise.materialise()
    return W.reshape(-1, 1) if noise.bcast_d else W


def rheun_z(y0, z0, f0, g0, dt, sign, noise, out=None):
    """z1 = ((2*y0 - z0) + sign*(f0*dt)) + sign*(g0*dW), diagonal noise (reversible_heun.py:69 / :109)."""
    if _needs_grad(y0, z0, f0, g0):
        return 2 * y0 - z0 + sign * (f0 * dt) + sign * (g0 * _noise_W(noise))
    y0 = _native.contiguous(y0)
    z0, f0, g0 = _prep(y0, z0, f0, g0)
    out = _new_like(y0, out)
    lib, dt_code, stream = _launch_env(y0)
    code = lib.tsde_rheun_z_diag(out.data_ptr(), y0.data_ptr(), z0.data_ptr(), f0.data_ptr(), g0.data_ptr(),
                                 y0.numel(), dt, sign, noise.struct(), dt_code, stream)
    if code:
        _native.check(code, "tsde_rheun_z_diag")
    return out


def rheun_y(y0, f0, f1, g0, g1, half_dt, sign, noise, out=None):
    """y1 = (y0 + sign*((f0+f1)*half_dt)) + sign*((g0+g1)*(0.5*dW)), diagonal noise (:71 / :130-131)."""
    if _needs_grad(y0, f0, f1, g0, g1):
        return y0 + sign * ((f0 + f1) * half_dt) + sign * ((g0 + g1) * (0.5 * _noise_W(noise)))
    y0 = _native.contiguous(y0)
    f0, f1, g0, g1 = _prep(y0, f0, f1, g0, g1)
    out = _new_like(y0, out)
    lib, dt_code, stream = _launch_env(y0)
    code = lib.tsde_rheun_y_diag(out.data_ptr(), y0.data_ptr(), f0.data_ptr(), f1.data_ptr(), g0.data_ptr(),
                                 g1.data_ptr(), y0.numel(), half_dt, sign, noise.struct(), dt_code, stream)
    if code:
        _native.check(code, "tsde_rheun_y_diag")
    return out


def lincomb2(x, y, a, b, out=None):
    """out = a*x + b*y."""
    if _needs_grad(x, y):
        return a * x + b * y
    x = _native.contiguous(x)
    y, = _prep(x, y)
    out = _new_like(x, out)
    lib, dt_code, stream = _launch_env(x)
    code = lib.tsde_lincomb2(out.data_ptr(), x.data_ptr(), y.data_ptr(), x.numel(), a, b, dt_code, stream)
    if code:
        _native.check(code, "tsde_lincomb2")
    return out


def rheun_adj_a(ay, af0, ag0, half_dt, noise):
    """(af0 + ay*half_dt, ag0 + ay*(0.5*dW)), diagonal noise (reversible_heun.py:106-117)."""
    ay = _native.contiguous(ay)
    af0, ag0 = _prep(ay, af0, ag0)
    of, og = torch.empty_like(ay), torch.empty_like(ay)
    lib, dt_code, stream = _launch_env(ay)
    code = lib.tsde_rheun_adj_a_diag(of.data_ptr(), og.data_ptr(), ay.data_ptr(), af0.data_ptr(), ag0.data_ptr(),
                                     ay.numel(), half_dt, noise.struct(), dt_code, stream)
    if code:
        _native.check(code, "tsde_rheun_adj_a_diag")
    return of, og


def rheun_adj_b(ay, az0, vjp_z, dt, half_dt, noise):
    """(ay1, az1, af1, ag1) after the VJP, diagonal noise (reversible_heun.py:127,134-137)."""
    ay = _native.contiguous(ay)
    az0, vjp_z = _prep(ay, az0, vjp_z)
    outs = [torch.empty_like(ay) for _ in range(4)]
    lib, dt_code, stream = _launch_env(ay)
    code = lib.tsde_rheun_adj_b_diag(outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(), outs[3].data_ptr(),
                                     ay.data_ptr(), az0.data_ptr(), vjp_z.data_ptr(), ay.numel(), dt, half_dt,
                                     noise.struct(), dt_code, stream)
    if code:
        _native.check(code, "tsde_rheun_adj_b_diag")
    return outs


# ---- Heun / Euler-Heun / log-ODE helpers ------------------------------------------------------------------------
def heun_final(y0, f, fp, g, gp, dt, mode, noise, prod=False, out=None):
    """mode 0 (Heun): y0 + (((dt*(f+fp)) + g.dW) + gp.dW)*0.5 ; mode 1 (Euler-Heun): (y0 + dt*f) + ((g.dW + gp.dW)*0.5).
    Diagonal noise, or already-formed products when `prod`."""
    if _needs_grad(y0, f, fp, g, gp):
        p0, p1 = (g, gp) if prod else (g * _noise_W(noise), gp * _noise_W(noise))
        if mode == 0:
            return y0 + (dt * (f + fp) + p0 + p1) * 0.5
        return y0 + dt * f + (p0 + p1) * 0.5
    y0 = _native.contiguous(y0)
    f, fp, g, gp = _prep(y0, f, fp, g, gp)
    out = _new_like(y0, out)
    lib, dt_code, stream = _launch_env(y0)
    code = lib.tsde_heun_final(out.data_ptr(), y0.data_ptr(), f.data_ptr(), None if fp is None else fp.data_ptr(),
                               g.data_ptr(), gp.data_ptr(), y0.numel(), dt, mode, 1 if prod else 0,
                               None if prod else noise.struct(), dt_code, stream)
    if code:
        _native.check(code, "tsde_heun_final")
    return out


def levy_area(W, H, h, foster, entropy, elem0, cell, node, entropy_dev=None):
    """Davie/Foster Levy-area approximation A:(B,m,m) of one interval from its (W, H):(B,m)."""
    W = _native.contiguous(W)
    H, = _prep(W, H)
    B, m = W.shape
    A = torch.empty((B, m, m), dtype=W.dtype, device=W.device)
    lib, dt_code, stream = _launch_env(W)
    code = lib.tsde_levy_area(A.data_ptr(), W.data_ptr(), H.data_ptr(), B, m, float(h), 1 if foster else 0, entropy,
                              elem0, cell, node, None if entropy_dev is None else entropy_dev.data_ptr(), dt_code,
                              stream)
    if code:
        _native.check(code, "tsde_levy_area")
    return A


def levy_iterated_integrals(W, H, h, foster, entropy, elem0, cell, node, dt, ito, entropy_dev=None):
    """I = 0.5*(W W^T - [diag] dt) + A with the Davie / Foster A of (W, H) in ONE kernel (`tsde_levy_iterated_integrals`);
    None where the fused kernel does not serve the shape (the caller then makes the two calls)."""
    W = _native.contiguous(W)
    H, = _prep(W, H)
    B, m = W.shape
    out = torch.empty((B, m, m), dtype=W.dtype, device=W.device)
    lib, dt_code, stream = _launch_env(W)
    code = lib.tsde_levy_iterated_integrals(out.data_ptr(), W.data_ptr(), H.data_ptr(), B, m, float(h),
                                            1 if foster else 0, entropy, elem0, cell, node,
                                            None if entropy_dev is None else entropy_dev.data_ptr(), float(dt),
                                            1 if ito else 0, dt_code, stream)
    if code == 801:          # hipErrorNotSupported
        return None
    if code:
        _native.check(code, "tsde_levy_iterated_integrals")
    return out


def iterated_integrals(W, A, dt, ito):
    """I[b,k,l] = 0.5*(W_k W_l - [k==l] dt) + A[b,k,l] (Ito) / 0.5*W_k W_l + A (Stratonovich); A may be None."""
    W = _native.contiguous(W)
    B, m = W.shape
    if A is not None:
        A = _native.contiguous(A.to(W.dtype))
    out = torch.empty((B, m, m), dtype=W.dtype, device=W.device)
    lib, dt_code, stream = _launch_env(W)
    code = lib.tsde_iterated_integrals(out.data_ptr(), W.data_ptr(), None if A is None else A.data_ptr(), B, m,
                                       float(dt), 1 if ito else 0, dt_code, stream)
    if code:
        _native.check(code, "tsde_iterated_integrals")
    return out

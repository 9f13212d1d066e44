"""The SDE-module contract and its adapters.

Same contract as the reference (torchsde/_core/base_sde.py): an SDE object carries ``noise_type`` and
``sde_type`` and provides drift/diffusion through any of ``f``, ``g``, ``f_and_g``, ``g_prod``,
``f_and_g_prod`` (resolution priority: base_sde.py:51-73). ``ForwardSDE`` normalises whatever the user
supplied into that full set; ``RenameMethodsSDE`` implements ``names=``; ``SDELogqp`` appends the KL column.

Beyond the reference, ``ForwardSDE`` records *how* the diffusion-vector product is obtained
(``user_product``): when the user supplies only ``g``/``f_and_g`` the solvers keep the product inside the
fused HIP step kernel (increment generated in registers); when the user computes the product themselves
the increment has to be materialised for them first.
"""
import abc

import torch
from torch import nn

from .settings import NOISE_TYPES, SDE_TYPES


class BaseSDE(abc.ABC, nn.Module):
    """Base class that validates and stores ``noise_type`` / ``sde_type`` (base_sde.py:25-39)."""

    def __init__(self, noise_type, sde_type):
        super().__init__()
        if noise_type not in NOISE_TYPES:
            raise ValueError(f"Expected noise type in {NOISE_TYPES}, but found {noise_type}")
        if sde_type not in SDE_TYPES:
            raise ValueError(f"Expected sde type in {SDE_TYPES}, but found {sde_type}")
        self.noise_type = noise_type
        self.sde_type = sde_type


class SDEIto(BaseSDE):
    def __init__(self, noise_type):
        super().__init__(noise_type=noise_type, sde_type=SDE_TYPES.ito)


class SDEStratonovich(BaseSDE):
    def __init__(self, noise_type):
        super().__init__(noise_type=noise_type, sde_type=SDE_TYPES.stratonovich)


# ---- small autograd helpers (reference: torchsde/_core/misc.py:71-99) --------------------------------
def _zeros_for_none(grads, inputs):
    return [torch.zeros_like(x) if g is None else g for g, x in zip(grads, inputs)]


def vjp(outputs, inputs, **kwargs):
    """``grad_outputs^T d(outputs)/d(inputs)`` with None gradients replaced by zeros."""
    inputs = [inputs] if torch.is_tensor(inputs) else list(inputs)
    outputs = [outputs] if torch.is_tensor(outputs) else list(outputs)
    outputs = [o if o.requires_grad else o.detach().requires_grad_(True) for o in outputs]
    return _zeros_for_none(torch.autograd.grad(outputs, inputs, **kwargs), inputs)


def jvp(outputs, inputs, grad_inputs=None, **kwargs):
    """Forward-mode product via the double-backward trick (no repeated forward evaluation)."""
    inputs = [inputs] if torch.is_tensor(inputs) else list(inputs)
    outputs = [outputs] if torch.is_tensor(outputs) else list(outputs)
    outputs = [o if o.requires_grad else o.detach().requires_grad_(True) for o in outputs]
    dummies = [torch.zeros_like(o, requires_grad=True) for o in outputs]
    back = torch.autograd.grad(outputs, inputs, grad_outputs=dummies, create_graph=True, allow_unused=True)
    back = [b if b.requires_grad else b.detach().requires_grad_(True) for b in _zeros_for_none(back, inputs)]
    return _zeros_for_none(torch.autograd.grad(back, dummies, grad_outputs=grad_inputs, **kwargs), dummies)


def batch_mvp(m, v):
    """(B,d,m) x (B,m) -> (B,d)."""
    return torch.bmm(m, v.unsqueeze(-1)).squeeze(dim=-1)


class ForwardSDE(BaseSDE):
    """Normalised view of a user SDE: every accessor of the contract exists after construction."""

    def __init__(self, sde, fast_dg_ga_jvp_column_sum=False):
        super().__init__(sde_type=sde.sde_type, noise_type=sde.noise_type)
        self._base_sde = sde
        diagonal = sde.noise_type == NOISE_TYPES.diagonal

        has = {name: hasattr(sde, name) for name in ("f", "g", "f_and_g", "g_prod", "f_and_g_prod")}
        self.f = sde.f if has["f"] else self._missing_f
        self.g = sde.g if has["g"] else self._missing_g
        self.f_and_g = sde.f_and_g if has["f_and_g"] else self._f_then_g
        self.prod = self.prod_diagonal if diagonal else self.prod_default
        self.g_prod = sde.g_prod if has["g_prod"] else self._g_then_prod

        # Who evaluates g*v inside a step?  (base_sde.py:51-56)
        if has["f_and_g_prod"]:
            self.f_and_g_prod = sde.f_and_g_prod
            self.user_product = True
        elif has["f"] and has["g_prod"]:
            self.f_and_g_prod = self._f_and_user_g_prod
            self.user_product = True
        else:
            self.f_and_g_prod = self._f_and_g_then_prod
            self.user_product = False
        # g_prod alone (SRK additive, adjoint g_prod): user code iff the user wrote g_prod.
        self.user_g_prod = has["g_prod"]

        if diagonal:
            self.g_prod_and_gdg_prod = self.g_prod_and_gdg_prod_diagonal
        elif sde.noise_type == NOISE_TYPES.additive:
            self.g_prod_and_gdg_prod = self.g_prod_and_gdg_prod_additive
        else:
            self.g_prod_and_gdg_prod = self.g_prod_and_gdg_prod_default

        if sde.noise_type == NOISE_TYPES.general:
            self.dg_ga_jvp_column_sum = (self.dg_ga_jvp_column_sum_v2 if fast_dg_ga_jvp_column_sum
                                         else self.dg_ga_jvp_column_sum_v1)
        else:
            self.dg_ga_jvp_column_sum = self._zero

    # -- fall-backs for missing user methods -------------------------------------------------------
    def _missing_f(self, t, y):
        raise RuntimeError("Method `f` has not been provided, but is required for this method.")

    def _missing_g(self, t, y):
        raise RuntimeError("Method `g` has not been provided, but is required for this method.")

    def _f_then_g(self, t, y):
        if self.overlap_f_g and y.is_cuda and torch.cuda.is_current_stream_capturing():
            return self._f_beside_g(t, y)
        from . import graph
        recorder = graph._RECORDER
        if recorder is None:
            return self.f(t, y), self.g(t, y)
        # a screened eager solve (hip_graph="auto"): tell the recorder which of the two is running, so that it can
        # tell whether they share memory (only independent drift and diffusion may become parallel graph branches)
        try:
            recorder.enter_phase("f")
            f = self.f(t, y)
            recorder.enter_phase("g")
            g = self.g(t, y)
        finally:
            recorder.enter_phase(None)
        return f, g

    # While a solve is being captured into a HIP graph (options={"hip_graph": True}) the user's drift and diffusion are
    # recorded as two PARALLEL branches of the graph instead of one after the other: both only read (t, y), and each
    # is a handful of short memory-bound kernels that leave most of the chip idle while they ramp up and drain (the
    # headline's `mu * y` and `sigma * y`: 7.4 us each, 4.5 TB/s). Fork and join are graph edges -- they cost nothing
    # at replay. No `record_stream`: g's output is consumed on the main stream before the next fork event, and the
    # side stream touches memory again only behind that event. `options={"overlap_f_g": False}` records them in
    # sequence (for drift and diffusion methods that write to shared buffers).
    overlap_f_g = True
    _side_streams = {}

    def _f_beside_g(self, t, y):
        device = y.device
        main = torch.cuda.current_stream(device)
        side = ForwardSDE._side_streams.get(device)
        if side is None:
            side = ForwardSDE._side_streams[device] = torch.cuda.Stream(device)
        fork = torch.cuda.Event()
        fork.record(main)
        side.wait_event(fork)
        f = self.f(t, y)                 # Python order stays drift, then diffusion (as in `_f_then_g`)
        with torch.cuda.stream(side):
            g = self.g(t, y)
        join = torch.cuda.Event()
        join.record(side)
        main.wait_event(join)
        return f, g

    def _g_then_prod(self, t, y, v):
        return self.prod(self.g(t, y), v)

    def _f_and_user_g_prod(self, t, y, v):
        return self.f(t, y), self.g_prod(t, y, v)

    def _f_and_g_then_prod(self, t, y, v):
        f, g = self.f_and_g(t, y)
        return f, self.prod(g, v)

    # -- products ----------------------------------------------------------------------------------
    def prod_diagonal(self, g, v):
        return g * v

    def prod_default(self, g, v):
        return batch_mvp(g, v)

    # -- Milstein pieces: g*v1 and (dg/dy)^T (g*v2)   (base_sde.py:127-158) -------------------------
    def _g_and_gdg(self, t, y, v2, scalar_like):
        """Returns (g, vjp(g, y, g*v2)) -- the part of Milstein that has to stay in autograd."""
        keep_graph = torch.is_grad_enabled()
        with torch.enable_grad():
            y = y if y.requires_grad else y.detach().requires_grad_(True)
            g = self.g(t, y)
            # `v2` is the tensor v/2, or a callable that forms the cotangent g * v/2 itself (one fused kernel)
            weight = v2(g) if callable(v2) else g * (v2.unsqueeze(-2) if scalar_like else v2)
            gdg, = vjp(outputs=g, inputs=y, grad_outputs=weight, retain_graph=True, create_graph=keep_graph,
                       allow_unused=True)
        return g, gdg

    def g_prod_and_gdg_prod_default(self, t, y, v1, v2):
        g, gdg = self._g_and_gdg(t, y, v2, scalar_like=True)
        return self.prod(g, v1), gdg

    def g_prod_and_gdg_prod_diagonal(self, t, y, v1, v2):
        g, gdg = self._g_and_gdg(t, y, v2, scalar_like=False)
        return self.prod(g, v1), gdg

    def g_prod_and_gdg_prod_additive(self, t, y, v1, v2):
        return self.g_prod(t, y, v1), 0.

    # -- Levy-area term of log-ODE / general Milstein: sum_{j,k,l} dg_{i,l}/dy_j g_{j,k} A_{k,l} ----
    def dg_ga_jvp_column_sum_v1(self, t, y, a):
        keep_graph = torch.is_grad_enabled()
        with torch.enable_grad():
            y = y if y.requires_grad else y.detach().requires_grad_(True)
            g = self.g(t, y)
            ga = torch.bmm(g, a)
            total = 0.
            for col in range(g.size(-1)):
                total = total + jvp(outputs=g[..., col], inputs=y, grad_inputs=ga[..., col], retain_graph=True,
                                    create_graph=keep_graph, allow_unused=True)[0]
        return total

    def dg_ga_jvp_column_sum_v2(self, t, y, a):
        keep_graph = torch.is_grad_enabled()
        with torch.enable_grad():
            y = y if y.requires_grad else y.detach().requires_grad_(True)
            g = self.g(t, y)
            ga = torch.bmm(g, a)
            batch, d, m = g.size()
            y_rep = torch.repeat_interleave(y, repeats=m, dim=0)
            g_rep = self.g(t, y_rep)
            out, = jvp(outputs=g_rep, inputs=y_rep, grad_inputs=ga.transpose(1, 2).flatten(0, 1),
                       create_graph=keep_graph, allow_unused=True)
            out = out.reshape(batch, m, d, m).permute(0, 2, 1, 3)
            return out.diagonal(dim1=-2, dim2=-1).sum(-1)

    def _zero(self, t, y, v):
        return 0.


class RenameMethodsSDE(BaseSDE):
    """Exposes user methods under the canonical names (``names=`` argument; base_sde.py:212-224)."""

    _CANONICAL = (("drift", "f"), ("diffusion", "g"), ("prior_drift", "h"), ("diffusion_prod", "g_prod"),
                  ("drift_and_diffusion", "f_and_g"), ("drift_and_diffusion_prod", "f_and_g_prod"))

    def __init__(self, sde, drift="f", diffusion="g", prior_drift="h", diffusion_prod="g_prod",
                 drift_and_diffusion="f_and_g", drift_and_diffusion_prod="f_and_g_prod"):
        super().__init__(noise_type=sde.noise_type, sde_type=sde.sde_type)
        self._base_sde = sde
        chosen = dict(drift=drift, diffusion=diffusion, prior_drift=prior_drift, diffusion_prod=diffusion_prod,
                      drift_and_diffusion=drift_and_diffusion, drift_and_diffusion_prod=drift_and_diffusion_prod)
        for key, canonical in self._CANONICAL:
            if hasattr(sde, chosen[key]):
                setattr(self, canonical, getattr(sde, chosen[key]))


def stable_division(a, b, epsilon=1e-7):
    b = torch.where(b.abs().detach() > epsilon, b, torch.full_like(b, fill_value=epsilon) * b.sign())
    return a / b


class SDELogqp(BaseSDE):
    """Augments the state with the running KL term 0.5*|u|^2, u = g^{-1}(f - h)  (base_sde.py:240-306)."""

    def __init__(self, sde):
        super().__init__(noise_type=sde.noise_type, sde_type=sde.sde_type)
        self._base_sde = sde
        try:
            self._base_f, self._base_g, self._base_h = sde.f, sde.g, sde.h
        except AttributeError as e:
            raise AttributeError("If using logqp then drift, diffusion and prior drift must all be specified.") from e
        self._diagonal = sde.noise_type == NOISE_TYPES.diagonal

    def _parts(self, t, y):
        y = y[:, :-1]
        f, g, h = self._base_f(t, y), self._base_g(t, y), self._base_h(t, y)
        if self._diagonal:
            u = stable_division(f - h, g)
            g_pad = y.new_zeros(size=(y.size(0), 1))
        else:
            u = batch_mvp(g.pinverse(), f - h)
            g_pad = y.new_zeros(size=(g.size(0), 1, g.size(-1)))
        f_logqp = .5 * (u ** 2).sum(dim=1, keepdim=True)
        return torch.cat([f, f_logqp], dim=1), torch.cat([g, g_pad], dim=1)

    def f(self, t, y):
        return self._parts(t, y)[0]

    def g(self, t, y):
        y = y[:, :-1]
        g = self._base_g(t, y)
        if self._diagonal:
            return torch.cat([g, y.new_zeros(size=(y.size(0), 1))], dim=1)
        return torch.cat([g, y.new_zeros(size=(g.size(0), 1, g.size(-1)))], dim=1)

    def f_and_g(self, t, y):
        return self._parts(t, y)

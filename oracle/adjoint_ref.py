"""ORACLE (test infrastructure only): torch-CPU restatement of the reference's stochastic adjoint.

Follows /root/reference/torchsde/_core/adjoint.py:64-127 (the backward pass of ``_SdeintAdjointMethod``) and
adjoint_sde.py:23-377 (``AdjointSDE``): the augmented state is ONE flat tensor ``(1, 2*B*d + P)`` that is
unpacked/re-packed around every VJP, pushed through the generic Euler / midpoint / Milstein step of
``solvers_ref`` with the time-reversed Brownian motion, reset to the stored forward state at every output time.
Pinned by tests/golden/adjoint_*.npz (gradients produced by the real reference under replayed increments,
tests/test_oracle_adjoint.py).
"""
import torch

from . import solvers_ref


def _flatten(seq):
    return torch.cat([p.reshape(-1) for p in seq]) if len(seq) > 0 else torch.tensor([])


def _flat_to_shape(flat, shapes):
    numels = [s.numel() for s in shapes]
    return [x.reshape(s) for x, s in zip(flat.split(split_size=numels), shapes)]


def _vjp(outputs, inputs, grad_outputs=None, **kw):
    """misc.py:71-81: autograd.grad with missing gradients as zeros and non-differentiable outputs tolerated."""
    outputs = [outputs] if torch.is_tensor(outputs) else list(outputs)
    outputs = [o if o.requires_grad else o.detach().requires_grad_(True) for o in outputs]
    grads = torch.autograd.grad(outputs, inputs, grad_outputs=grad_outputs, allow_unused=True, **kw)
    return [torch.zeros_like(x) if g is None else g for g, x in zip(grads, inputs)]


def _jvp(output, inp, direction):
    """misc.py:84-99 (double-backward trick)."""
    output = output if output.requires_grad else output.detach().requires_grad_(True)
    dummy = torch.zeros_like(output, requires_grad=True)
    back, = torch.autograd.grad(output, inp, grad_outputs=dummy, create_graph=True, allow_unused=True)
    if back is None:
        return torch.zeros_like(output)
    out, = torch.autograd.grad(back, dummy, grad_outputs=direction, create_graph=True, allow_unused=True)
    return torch.zeros_like(output) if out is None else out


class AdjointSDERef:
    """adjoint_sde.py:23-377 on the flat augmented state; exposes what the generic solver steps need."""

    def __init__(self, sde, params, shapes):
        self.fwd, self.params, self.shapes = sde, list(params), shapes
        self.sde_type = sde.sde_type
        self.noise_type = {"general": "general", "additive": "general", "scalar": "scalar",
                           "diagonal": "diagonal"}[sde.noise_type]
        ito = sde.sde_type == "ito"
        if not ito or sde.noise_type == "additive":
            self.corr = None
        elif sde.noise_type == "diagonal":
            self.corr = "diagonal"
        else:
            self.corr = "default"

    def _state(self, y_aug):
        numel = sum(s.numel() for s in self.shapes[:2])
        y, a = _flat_to_shape(y_aug.squeeze(0)[:numel], self.shapes[:2])
        return y.detach().requires_grad_(True), a

    def _pack(self, head, grads):
        return _flatten([-head.detach()] + grads).unsqueeze(0)

    def _f_parts(self, f, g, y, a):
        if self.corr is None:                                              # :111-128
            return self._pack(f, _vjp(f, [y] + self.params, grad_outputs=a, retain_graph=True))
        if self.corr == "diagonal":                                        # :177-216
            g_dg, = _vjp(g, [y], grad_outputs=g, create_graph=True)
            f = f - g_dg
            grads = _vjp(f, [y] + self.params, grad_outputs=a, retain_graph=True)
            a_dg, = _vjp(g, [y], grad_outputs=a, retain_graph=True)
            extra = _vjp(g, [y] + self.params, grad_outputs=a_dg, retain_graph=True)
            return self._pack(f, [p + q for p, q in zip(grads, extra)])
        cols = [c.squeeze(-1) for c in g.split(1, dim=-1)]                 # :130-175
        f = f - sum(_jvp(c, y, c) for c in cols)
        grads = _vjp(f, [y] + self.params, grad_outputs=a, retain_graph=True)
        for c in cols:
            a_dg, = _vjp(c, [y], grad_outputs=a, retain_graph=True)
            extra = _vjp(c, [y] + self.params, grad_outputs=a_dg, retain_graph=True)
            grads = [p + q for p, q in zip(grads, extra)]
        return self._pack(f, grads)

    def _g_parts(self, gp, y, a):                                          # :218-230
        return self._pack(gp, _vjp(gp, [y] + self.params, grad_outputs=a, retain_graph=True))

    def f(self, t, y_aug):                                                 # :236-252
        y, a = self._state(y_aug)
        with torch.enable_grad():
            if self.corr is None:
                return self._f_parts(self.fwd.f(-t, y), None, y, a)
            f, g = solvers_ref.f_and_g(self.fwd, -t, y)
            return self._f_parts(f, g, y, a)

    def f_and_g_prod(self, t, y_aug, v):                                   # :296-323
        y, a = self._state(y_aug)
        with torch.enable_grad():
            if self.corr is None:
                f, gp = solvers_ref.f_and_g_prod(self.fwd, -t, y, v)
                g = None
            else:
                f, g = solvers_ref.f_and_g(self.fwd, -t, y)
                gp = solvers_ref.prod(self.fwd, g, v)
            return self._f_parts(f, g, y, a), self._g_parts(gp, y, a)

    def g_prod(self, t, y_aug, v):                                         # :283-287
        y, a = self._state(y_aug)
        with torch.enable_grad():
            return self._g_parts(solvers_ref.g_prod(self.fwd, -t, y, v), y, a)

    def g_prod_and_gdg_prod(self, t, y_aug, v1, v2):                       # :332-377 (diagonal noise)
        y, a = self._state(y_aug)
        inputs = [y] + self.params
        with torch.enable_grad():
            g = self.fwd.g(-t, y)
            gp = solvers_ref.prod(self.fwd, g, v1)
            vg_dg, = _vjp(g, [y], grad_outputs=v2 * g, retain_graph=True)
            dgdy, = _vjp(g.sum(), [y], retain_graph=True)
            prod_partials = _vjp(g, inputs, grad_outputs=a * v2 * dgdy, retain_graph=True)
            avg_dg, = _vjp(g, [y], grad_outputs=(a * v2 * g).detach(), create_graph=True)
            mixed = _vjp(avg_dg.sum(), inputs, retain_graph=True)
            gdg = _flatten([vg_dg] + [p - q for p, q in zip(prod_partials, mixed)]).unsqueeze(0)
            return self._g_parts(gp, y, a), gdg


def _aug_step(adj, method, bm, t0, t1, aug):
    """The generic solver steps (euler.py:29-37, midpoint.py:29-45, milstein.py:52-74, heun.py:35-48,
    euler_heun.py:29-42) on the flat state."""
    dt = t1 - t0
    I_k = bm(t0, t1)
    if method == "euler":
        F, G = adj.f_and_g_prod(t0, aug, I_k)
        return aug + F * dt + G
    if method == "midpoint":
        F, G = adj.f_and_g_prod(t0, aug, I_k)
        half_dt = 0.5 * dt
        aug_prime = aug + half_dt * F + 0.5 * G
        F2, G2 = adj.f_and_g_prod(t0 + half_dt, aug_prime, I_k)
        return aug + dt * F2 + G2
    if method == "milstein":
        v = I_k ** 2 - dt if adj.sde_type == "ito" else I_k ** 2
        F = adj.f(t0, aug)
        G, D = adj.g_prod_and_gdg_prod(t0, aug, I_k, 0.5 * v)
        return aug + F * dt + G + D
    if method == "heun":                           # methods/heun.py:35-48
        F, G = adj.f_and_g_prod(t0, aug, I_k)
        aug_prime = aug + dt * F + G
        F2, G2 = adj.f_and_g_prod(t1, aug_prime, I_k)
        return aug + (dt * (F + F2) + G + G2) * 0.5
    if method == "euler_heun":                     # methods/euler_heun.py:29-42
        F, G = adj.f_and_g_prod(t0, aug, I_k)
        aug_prime = aug + G
        G2 = adj.g_prod(t1, aug_prime, I_k)
        return aug + dt * F + (G + G2) * 0.5
    raise ValueError(method)


def default_adjoint_method(sde, method):
    """adjoint.py:281-296."""
    if sde.sde_type == "stratonovich":
        return "midpoint"
    return "milstein" if sde.noise_type == "diagonal" else "euler"


def adjoint_gradients(sde, y0, ts, bm, dt, method, adjoint_method, loss_weights, options=None):
    """Forward solve + the reference's backward pass. Returns (ys, dL/dy0, [dL/dtheta]) for L = sum(ys * w)."""
    params = [p for p in sde.parameters() if p.requires_grad]
    adjoint_method = adjoint_method or default_adjoint_method(sde, method)
    with torch.no_grad():
        ys = solvers_ref.integrate(sde, bm, y0.detach(), ts, dt, method, options)
    grad_ys = loss_weights

    def reverse_bm(ta, tb, return_U=False):        # derived.py:27-30
        return bm(-tb, -ta, return_U=return_U)

    aug = [ys[-1], grad_ys[-1]] + [torch.zeros_like(p) for p in params]
    shapes = [t.size() for t in aug]
    adj = AdjointSDERef(sde, params, shapes)
    aug = _flatten(aug).unsqueeze(0)
    T = ys.size(0)
    for i in range(T - 1, 0, -1):
        t_lo, t_hi = -ts[i], -ts[i - 1]
        curr_t = t_lo
        while curr_t < t_hi:                        # base_solver.py:114-116 on [-ts[i], -ts[i-1]]
            next_t = min(curr_t + dt, t_hi)
            aug = _aug_step(adj, adjoint_method, reverse_bm, curr_t, next_t, aug).detach()
            curr_t = next_t
        parts = _flat_to_shape(aug.squeeze(0), shapes)
        parts[0] = ys[i - 1]
        parts[1] = parts[1] + grad_ys[i - 1]
        aug = _flatten(parts).unsqueeze(0)
    parts = _flat_to_shape(aug.squeeze(0), shapes)
    return ys, parts[1], parts[2:]

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


def adjoint_gradients(sde, y0, ts, bm, dt, method, adjoint_method, loss_weights, options=None, adjoint_adaptive=False,
                      adjoint_rtol=1e-5, adjoint_atol=1e-4, dt_min=1e-5, record=None):
    """Forward solve + the reference's backward pass. Returns (ys, dL/dy0, [dL/dtheta]) for L = sum(ys * w).
    `adjoint_adaptive`: the step-doubling branch of base_solver.py:117-142 on the flat augmented state, one
    integrate() call -- hence a restart from `dt` -- per output interval (adjoint.py:97-112)."""
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
        step_size, prev_error_ratio = dt, None
        while curr_t < t_hi:                        # base_solver.py:114-142 on [-ts[i], -ts[i-1]]
            next_t = min(curr_t + step_size, t_hi)
            if record is not None:
                record.append((float(curr_t), float(next_t)))
            if not adjoint_adaptive:
                aug = _aug_step(adj, adjoint_method, reverse_bm, curr_t, next_t, aug).detach()
                curr_t = next_t
                continue
            full = _aug_step(adj, adjoint_method, reverse_bm, curr_t, next_t, aug).detach()
            mid_t = 0.5 * (curr_t + next_t)
            half = _aug_step(adj, adjoint_method, reverse_bm, curr_t, mid_t, aug).detach()
            two = _aug_step(adj, adjoint_method, reverse_bm, mid_t, next_t, half).detach()
            error_estimate = solvers_ref._compute_error(full, two, adjoint_rtol, adjoint_atol)
            step_size, prev_error_ratio = solvers_ref._update_step_size(error_estimate, step_size,
                                                                        prev_error_ratio=prev_error_ratio)
            if step_size < dt_min:
                step_size, prev_error_ratio = dt_min, None
            if error_estimate <= 1 or step_size <= dt_min:
                curr_t, aug = next_t, two
        parts = _flat_to_shape(aug.squeeze(0), shapes)
        parts[0] = ys[i - 1]
        parts[1] = parts[1] + grad_ys[i - 1]
        aug = _flatten(parts).unsqueeze(0)
    parts = _flat_to_shape(aug.squeeze(0), shapes)
    return ys, parts[1], parts[2:]


# ---- the reversible-Heun pair (method="reversible_heun", adjoint_method="adjoint_reversible_heun") -----------------
def _rheun_adjoint_step(sde, params, bm, t0, t1, state):
    """methods/reversible_heun.py:98-144 on the unflattened state
    (y, a_y, a_f, a_g, a_z, [a_theta]) + carried (f, g, z); `bm` is the time-reversed Brownian motion."""
    (y0, a_y0, a_f0, a_g0, a_z0, a_theta), (f0, g0, z0) = state
    diagonal = sde.noise_type == "diagonal"

    def adjoint_of_prod(a, v):                                             # :85-88
        return a * v if diagonal else a.unsqueeze(-1) * v.unsqueeze(-2)

    dt = t1 - t0
    dW = bm(t0, t1)
    half_dt, half_dW = 0.5 * dt, 0.5 * dW
    a_y0_half_dt = a_y0 * half_dt
    a_y0_half_dW = adjoint_of_prod(a_y0, half_dW)
    z1 = 2 * y0 - z0 - f0 * dt - solvers_ref.prod(sde, g0, dW)
    a_f1, a_g1 = a_y0_half_dt, a_y0_half_dW
    a_f0 = a_f0 + a_y0_half_dt
    a_g0 = a_g0 + a_y0_half_dW
    z_leaf = z0.detach().requires_grad_(True)
    with torch.enable_grad():
        re_f, re_g = solvers_ref.f_and_g(sde, -t0, z_leaf)
        vjp_z, *vjp_theta = _vjp((re_f, re_g), [z_leaf] + list(params), grad_outputs=[a_f0, a_g0])
    a_z0 = a_z0 + vjp_z
    a_theta = [p + q for p, q in zip(a_theta, vjp_theta)]
    with torch.no_grad():
        f1, g1 = solvers_ref.f_and_g(sde, -t1, z1)
    y1 = y0 - (f0 + f1) * half_dt - solvers_ref.prod(sde, g0 + g1, half_dW)
    a_y1 = a_y0 + 2 * a_z0
    a_z1 = -a_z0
    a_f1 = a_f1 + a_z0 * dt
    a_g1 = a_g1 + adjoint_of_prod(a_z0, dW)
    return (y1.detach(), a_y1.detach(), a_f1.detach(), a_g1.detach(), a_z1.detach(),
            [p.detach() for p in a_theta]), (f1.detach(), g1.detach(), z1.detach())


def reversible_heun_adjoint_gradients(sde, y0, ts, bm, dt, loss_weights, adjoint_adaptive=False, adjoint_rtol=1e-5,
                                      adjoint_atol=1e-4, dt_min=1e-5):
    """Forward reversible-Heun solve + its algebraically reversed backward pass (adjoint.py:64-127 driving
    methods/reversible_heun.py:76-144). Returns (ys, dL/dy0, [dL/dtheta]) for L = sum(ys * w). With
    `adjoint_adaptive` the step goes through the generic step-doubling loop (base_solver.py:117-142), the error
    norm over the flat (y, a_y, a_f, a_g, a_z, a_theta), restarted from `dt` on every output interval."""
    params = [p for p in sde.parameters() if p.requires_grad]
    with torch.no_grad():
        ys, (f, g, z) = solvers_ref.integrate_reversible_heun(sde, bm, y0.detach(), ts, dt)
    grad_ys = loss_weights

    def reverse_bm(ta, tb, return_U=False):
        return bm(-tb, -ta, return_U=return_U)

    def flat(head):
        return _flatten(list(head[:5]) + list(head[5])).unsqueeze(0)

    head = (ys[-1], grad_ys[-1], torch.zeros_like(f), torch.zeros_like(g), torch.zeros_like(z),
            [torch.zeros_like(p) for p in params])
    extra = (f, g, z)
    for i in range(ys.size(0) - 1, 0, -1):
        curr_t, t_hi = -ts[i], -ts[i - 1]
        step_size, prev_error_ratio = dt, None
        while curr_t < t_hi:
            next_t = min(curr_t + step_size, t_hi)
            if not adjoint_adaptive:
                head, extra = _rheun_adjoint_step(sde, params, reverse_bm, curr_t, next_t, (head, extra))
                curr_t = next_t
                continue
            full, _ = _rheun_adjoint_step(sde, params, reverse_bm, curr_t, next_t, (head, extra))
            mid_t = 0.5 * (curr_t + next_t)
            half = _rheun_adjoint_step(sde, params, reverse_bm, curr_t, mid_t, (head, extra))
            two, two_extra = _rheun_adjoint_step(sde, params, reverse_bm, mid_t, next_t, half)
            error_estimate = solvers_ref._compute_error(flat(full), flat(two), adjoint_rtol, adjoint_atol)
            step_size, prev_error_ratio = solvers_ref._update_step_size(error_estimate, step_size,
                                                                        prev_error_ratio=prev_error_ratio)
            if step_size < dt_min:
                step_size, prev_error_ratio = dt_min, None
            if error_estimate <= 1 or step_size <= dt_min:
                curr_t, head, extra = next_t, two, two_extra
        head = (ys[i - 1], head[1] + grad_ys[i - 1]) + tuple(head[2:])              # adjoint.py:114-116
    # the solver's initial (f, g, z) = (f(t0, y0), g(t0, y0), y0) are computed OUTSIDE the autograd Function
    # (adjoint.py:262-264 -> reversible_heun.py:58-59), so their cotangents reach y0 and theta by plain autograd
    _, a_y, a_f, a_g, a_z, a_theta = head
    y_leaf = y0.detach().requires_grad_(True)
    with torch.enable_grad():
        f0, g0 = solvers_ref.f_and_g(sde, ts[0], y_leaf)
        vjp_y, *vjp_theta = _vjp((f0, g0), [y_leaf] + params, grad_outputs=[a_f, a_g])
    return ys, a_y + a_z + vjp_y, [p + q for p, q in zip(a_theta, vjp_theta)]

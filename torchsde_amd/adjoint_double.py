"""Second derivatives through ``sdeint_adjoint``: the backward sweep as a differentiable torch program.

The reference supports ``create_graph=True`` on the gradients of ``sdeint_adjoint`` by nesting: its backward pass is
itself a ``_SdeintAdjointMethod.apply`` on the adjoint SDE (adjoint.py:97-112), so differentiating it starts another
adjoint solve of the adjoint SDE (tests/test_adjoint.py:180-218 exercises it).

Here the first-order sweep is HIP kernels (no autograd graph). When the caller asks for a graph of the backward pass
(``torch.is_grad_enabled()`` inside ``backward``), the same sweep -- same reversed grid, same Brownian increments,
same update formulas per ``adjoint_method`` -- runs as plain differentiable torch operations instead, every
vector-Jacobian product taken with ``create_graph=True``. Autograd then differentiates the discrete backward map
exactly (discretise-then-differentiate), including its dependence on the stored forward states: those are outputs
of the first-order Function, so their cotangents re-enter it as an ordinary (kernel) backward pass. The memory of
this graph grows with the number of backward steps, unlike the reference's nested adjoint; the values agree with
the reference's to discretisation error, not to rounding (both approximate the same continuous second derivative).
"""
import torch

from .brownian import BrownianInterval, ReverseBrownian
from .kernels import NoiseSpec
from .sde import jvp, vjp
from .settings import NOISE_TYPES, SDE_TYPES

SUPPORTED = ("euler", "midpoint", "heun", "euler_heun", "milstein")


def _grad(outputs, inputs, weights):
    return vjp(outputs, inputs, grad_outputs=weights, allow_unused=True, create_graph=True)


class _Rewire(torch.autograd.Function):
    """``parts = P(y_in, a_in, theta)`` were computed on leaf copies (y_in, a_in) of the sweep's (y, a); hand them on
    with ``d parts / d (y, a) := d parts / d (y_in, a_in)``.

    Why not compute on (y, a) directly: in the sweep y and a are functions of theta (through the earlier steps), and
    ``autograd.grad(f(y), [y, theta])`` then returns the TOTAL derivative w.r.t. theta, through y's history too. The
    adjoint's terms are partial derivatives at fixed (y, a). The leaf copies give exactly those; this node then puts
    the dependence on the sweep's history back for the second-order pass."""

    @staticmethod
    def forward(ctx, n_leaves, *tensors):
        ctx.n_leaves = n_leaves
        ctx.save_for_backward(*tensors[n_leaves:])          # (y_in, a_in, *parts)
        return tuple(p.clone() for p in tensors[2 * n_leaves:])

    @staticmethod
    def backward(ctx, *cotangents):
        saved = ctx.saved_tensors
        leaves, parts = saved[:ctx.n_leaves], saved[ctx.n_leaves:]
        live = [(p, c) for p, c in zip(parts, cotangents) if p.requires_grad and c is not None]
        through = [None] * ctx.n_leaves
        if live:
            with torch.enable_grad():
                through = torch.autograd.grad([p for p, _ in live], leaves, grad_outputs=[c for _, c in live],
                                              allow_unused=True, retain_graph=True,
                                              create_graph=torch.is_grad_enabled())
        # the originals get what reached the copies; the parts' own graph carries the rest (to theta)
        return (None, *through, *([None] * ctx.n_leaves), *cotangents)


def _rewired(originals, copies, parts):
    flat = [p for group in parts for p in group]
    out = list(_Rewire.apply(len(originals), *originals, *copies, *flat))
    groups, k = [], 0
    for group in parts:
        groups.append(out[k:k + len(group)])
        k += len(group)
    return groups


def _leaf(x):
    return x.detach().requires_grad_(True)


class _GraphAdjoint:
    """Drift / diffusion-product parts of the augmented backward SDE, as graph-carrying tensors.

    Same quantities as ``adjoint.AdjointSDE`` (reference: adjoint_sde.py:111-377), every product taken with
    ``create_graph=True`` on leaf copies of (y, a) and re-attached to the sweep by `_Rewire`."""

    def __init__(self, adjoint_sde):
        self.fwd = adjoint_sde.forward_sde
        self.params = adjoint_sde.params
        self.correction = adjoint_sde._correction

    def _drift(self, f, g, y, a):
        inputs = [y] + self.params
        if self.correction is None:
            return [f] + _grad(f, inputs, a)
        if self.correction == "diagonal":
            f = f - _grad(g, y, g)[0]
            grads = _grad(f, inputs, a)
            a_dg, = _grad(g, y, a)
            return [f] + [p + q for p, q in zip(grads, _grad(g, inputs, a_dg))]
        columns = [c.squeeze(dim=-1) for c in g.split(1, dim=-1)]
        f = f - sum(jvp(c, y, grad_inputs=c, allow_unused=True, create_graph=True)[0] for c in columns)
        grads = _grad(f, inputs, a)
        for c in columns:
            a_dg, = _grad(c, y, a)
            grads = [p + q for p, q in zip(grads, _grad(c, inputs, a_dg))]
        return [f] + grads

    def f(self, t, y0, a0):
        y, a = _leaf(y0), _leaf(a0)
        if self.correction is None:
            parts = self._drift(self.fwd.f(t, y), None, y, a)
        else:
            f, g = self.fwd.f_and_g(t, y)
            parts = self._drift(f, g, y, a)
        return _rewired([y0, a0], [y, a], [parts])[0]

    def g_prod(self, t, y0, a0, v):
        y, a = _leaf(y0), _leaf(a0)
        gp = self.fwd.g_prod(t, y, v)
        return _rewired([y0, a0], [y, a], [[gp] + _grad(gp, [y] + self.params, a)])[0]

    def f_and_g_prod(self, t, y0, a0, v):
        y, a = _leaf(y0), _leaf(a0)
        f, g = self.fwd.f_and_g(t, y)
        gp = self.fwd.prod(g, v)
        return _rewired([y0, a0], [y, a], [self._drift(f, g, y, a), [gp] + _grad(gp, [y] + self.params, a)])

    def g_prod_and_gdg_prod(self, t, y0, a0, v1, v2):
        """adjoint_sde.py:332-377. The reference's mixed-partial term differentiates ``sum_j (c . dg/dy_j)`` holding the
        weight c = a v2 g fixed; with a graph-carrying c the same value is the full derivative minus the part that goes
        through c."""
        y, a = _leaf(y0), _leaf(a0)
        inputs = [y] + self.params
        g = self.fwd.g(t, y)
        gp = self.fwd.prod(g, v1)
        vg_dg, = _grad(g, y, v2 * g)
        dgdy, = _grad(g.sum(), y, None)
        prod_partials = _grad(g, inputs, a * v2 * dgdy)
        c = a * v2 * g
        weighted, = _grad(g, y, c)
        full = _grad(weighted.sum(), inputs, None)
        row_sums, = jvp(g, y, grad_inputs=torch.ones_like(y), allow_unused=True, create_graph=True)
        through_c = _grad(c, inputs, row_sums)
        mixed = [p - q for p, q in zip(full, through_c)]
        gdg = [vg_dg] + [p - q for p, q in zip(prod_partials, mixed)]
        return _rewired([y0, a0], [y, a], [[gp] + _grad(gp, inputs, a), gdg])


def _combine(state, terms):
    """state + sum of (weight, parts) with the reference's minus sign on the y segment for drift / diffusion parts;
    the Milstein correction parts (sign None) enter every segment with a plus."""
    out = []
    for i, s in enumerate(state):
        acc = s
        for weight, parts, signed in terms:
            p = parts[i]
            if p is None:
                continue
            if p.shape != s.shape:
                p = p.reshape(s.shape) if p.numel() == s.numel() else p.expand(s.shape)
            acc = acc + p * (-weight if (signed and i == 0) else weight)
        out.append(acc)
    return out


def _step(adj, kind, ito, state, t0, t_half, t1, h, v):
    """One backward step of the augmented state (list of tensors); the formulas of ``adjoint._aug_step``."""
    y, a = state[0], state[1]
    h = float(h)
    if kind == "euler":
        F, G = adj.f_and_g_prod(t0, y, a, v)
        return _combine(state, [(h, F, True), (1.0, G, True)])
    if kind == "midpoint":
        F, G = adj.f_and_g_prod(t0, y, a, v)
        mid = _combine(state, [(0.5 * h, F, True), (0.5, G, True)])
        F, G = adj.f_and_g_prod(t_half, mid[0], mid[1], v)
        return _combine(state, [(h, F, True), (1.0, G, True)])
    if kind == "heun":
        F, G = adj.f_and_g_prod(t0, y, a, v)
        prime = _combine(state, [(h, F, True), (1.0, G, True)])
        F2, G2 = adj.f_and_g_prod(t1, prime[0], prime[1], v)
        return _combine(state, [(0.5 * h, F, True), (0.5 * h, F2, True), (0.5, G, True), (0.5, G2, True)])
    if kind == "euler_heun":
        F, G = adj.f_and_g_prod(t0, y, a, v)
        prime = _combine(state, [(1.0, G, True)])
        G2 = adj.g_prod(t1, prime[0], prime[1], v)
        return _combine(state, [(h, F, True), (0.5, G, True), (0.5, G2, True)])
    v2 = 0.5 * (v * v - h) if ito else 0.5 * (v * v)        # milstein.py:56,70
    F = adj.f(t0, y, a)
    G, D = adj.g_prod_and_gdg_prod(t0, y, a, v, v2)
    return _combine(state, [(h, F, True), (1.0, G, True), (1.0, D, False)])


def run(adjoint_sde, kind, bm, plan, ys, grad_ys):
    """(a_y0, [a_theta...]) carrying the graph of the whole sweep. `plan` is ``adjoint._plan_backward``'s."""
    if kind not in SUPPORTED:
        raise NotImplementedError(f"torchsde_amd: double backward is not available for this adjoint method ({kind}); "
                                  f"use one of {SUPPORTED}.")
    if kind == "milstein" and adjoint_sde.forward_sde.noise_type != NOISE_TYPES.diagonal:
        raise NotImplementedError
    adj = _GraphAdjoint(adjoint_sde)
    ito = adjoint_sde.forward_sde.sde_type == SDE_TYPES.ito
    native = bm if isinstance(bm, BrownianInterval) else None
    reverse_bm = None if native is not None else ReverseBrownian(bm)
    with torch.enable_grad():
        state = [ys[-1], grad_ys[-1]] + [torch.zeros_like(p) for p in adjoint_sde.params]
        for (i, grid, tau64, stage_rows, cells, tau_dev) in plan:
            n = grid.n_steps
            for k in range(n):
                if native is not None:
                    if cells is not None:
                        c = int(cells[n - 1 - k])
                        v, _ = NoiseSpec.generated(native, c, native.cell_width(c)).materialise()
                    else:
                        v, _ = native.increment(-tau64[k + 1], -tau64[k])
                else:
                    v = reverse_bm(tau_dev[k], tau_dev[k + 1])
                state = _step(adj, kind, ito, state, *stage_rows[k], grid.dt[k], v.detach())
            state[0] = ys[i - 1]                          # adjoint.py:114-116
            state[1] = state[1] + grad_ys[i - 1]
    return state[1], state[2:]

"""ORACLE (test infrastructure only): torch-CPU restatement of the reference's solver steps and loop.

Each function follows the cited lines of /root/reference/torchsde and uses plain elementwise torch CPU ops
(one IEEE rounding per op, in the tensor dtype), i.e. the arithmetic the reference itself performs.
Pinned by tests/golden/solver_*.npz, which hold outputs of the real reference under replayed increments
(tests/golden/make_golden.py, checked in tests/test_oracle_solvers.py).

`bm` is any callable ``bm(t0, t1, return_U=False)`` returning W or (W, U) as CPU tensors; `sde` is a plain
object with ``noise_type``, ``sde_type`` and ``f``/``g`` (optionally ``g_prod`` / ``f_and_g`` /
``f_and_g_prod``), resolved with the reference's priority (base_sde.py:51-73).
"""
import torch

# tableaus/srid2.py:19-54
SRID2 = dict(
    STAGES=4,
    C0=(0, 1, 1 / 2, 0), C1=(0, 1 / 4, 1, 1 / 4),
    A0=((), (1,), (1 / 4, 1 / 4), (0, 0, 0)),
    A1=((), (1 / 4,), (1, 0), (0, 0, 1 / 4)),
    B0=((), (0,), (1, 1 / 2), (0, 0, 0)),
    B1=((), (-1 / 2,), (1, 0), (2, -1, 1 / 2)),
    alpha=(1 / 6, 1 / 6, 2 / 3, 0),
    beta1=(-1, 4 / 3, 2 / 3, 0), beta2=(1, -4 / 3, 1 / 3, 0), beta3=(2, -4 / 3, -2 / 3, 0),
    beta4=(-2, 5 / 3, -2 / 3, 1),
)
# tableaus/sra1.py:19-36
SRA1 = dict(STAGES=2, C0=(0, 3 / 4), C1=(1, 0), A0=((), (3 / 4,)), B0=((), (3 / 2,)), alpha=(1 / 3, 2 / 3),
            beta1=(1, 0), beta2=(-1, 1))


# ---- SDE method resolution (base_sde.py:42-158) ----------------------------------------------------------
def prod(sde, g, v):
    """base_sde.py:98-102 + misc.py:62-63."""
    if sde.noise_type == "diagonal":
        return g * v
    return torch.bmm(g, v.unsqueeze(-1)).squeeze(dim=-1)


def g_prod(sde, t, y, v):
    if hasattr(sde, "g_prod"):
        return sde.g_prod(t, y, v)
    return prod(sde, sde.g(t, y), v)


def f_and_g(sde, t, y):
    if hasattr(sde, "f_and_g"):
        return sde.f_and_g(t, y)
    return sde.f(t, y), sde.g(t, y)


def f_and_g_prod(sde, t, y, v):
    """base_sde.py:51-56, 115-120."""
    if hasattr(sde, "f_and_g_prod"):
        return sde.f_and_g_prod(t, y, v)
    if hasattr(sde, "f") and hasattr(sde, "g_prod"):
        return sde.f(t, y), sde.g_prod(t, y, v)
    f, g = f_and_g(sde, t, y)
    return f, prod(sde, g, v)


def g_prod_and_gdg_prod(sde, t, y, v1, v2):
    """base_sde.py:127-158."""
    if sde.noise_type == "additive":
        return g_prod(sde, t, y, v1), 0.
    requires_grad = torch.is_grad_enabled()
    with torch.enable_grad():
        y = y if y.requires_grad else y.detach().requires_grad_(True)
        g = sde.g(t, y)
        weight = g * v2 if sde.noise_type == "diagonal" else g * v2.unsqueeze(-2)
        gdg, = torch.autograd.grad(g, y, grad_outputs=weight, retain_graph=True, create_graph=requires_grad,
                                   allow_unused=True)
        if gdg is None:
            gdg = torch.zeros_like(y)
    return prod(sde, g, v1), gdg


# ---- one step of each method ------------------------------------------------------------------------------
def euler_step(sde, bm, t0, t1, y0, options=None):
    """methods/euler.py:29-37."""
    dt = t1 - t0
    I_k = bm(t0, t1)
    f, gp = f_and_g_prod(sde, t0, y0, I_k)
    return y0 + f * dt + gp


def midpoint_step(sde, bm, t0, t1, y0, options=None):
    """methods/midpoint.py:29-45."""
    dt = t1 - t0
    I_k = bm(t0, t1)
    f, gp = f_and_g_prod(sde, t0, y0, I_k)
    half_dt = 0.5 * dt
    t_prime = t0 + half_dt
    y_prime = y0 + half_dt * f + 0.5 * gp
    f_prime, gp_prime = f_and_g_prod(sde, t_prime, y_prime, I_k)
    return y0 + dt * f_prime + gp_prime


def milstein_step(sde, bm, t0, t1, y0, options=None):
    """methods/milstein.py:52-94."""
    options = options or {}
    ito = sde.sde_type == "ito"
    grad_free = options.get("grad_free", False) and sde.noise_type != "additive"
    dt = t1 - t0
    I_k = bm(t0, t1)
    v = I_k ** 2 - dt if ito else I_k ** 2
    if grad_free:
        f, g = f_and_g(sde, t0, y0)
        g_ = g.squeeze(2) if g.dim() == 3 else g
        sqrt_dt = dt.sqrt()
        y0_prime = y0 + (dt * f if ito else 0.) + g_ * sqrt_dt
        g_prime = sde.g(t0, y0_prime)
        gp = prod(sde, g, I_k)
        gdg = prod(sde, g_prime - g, v) / (2 * sqrt_dt)
    else:
        f = sde.f(t0, y0)
        gp, gdg = g_prod_and_gdg_prod(sde, t0, y0, I_k, 0.5 * v)
    return y0 + f * dt + gp + gdg


def srk_step(sde, bm, t0, t1, y0, options=None):
    """methods/srk.py:57-111 (the reference's loops, including its repeated f/g evaluations)."""
    dt = t1 - t0
    rdt = 1 / dt
    I_k, I_k0 = bm(t0, t1, return_U=True)
    if sde.noise_type == "additive":
        tb = SRA1
        y1 = y0
        H0 = []
        for i in range(tb["STAGES"]):
            H0i = y0
            for j in range(i):
                f = sde.f(t0 + tb["C0"][j] * dt, H0[j])
                gw = tb["B0"][i][j] * I_k0 * rdt
                H0i = H0i + tb["A0"][i][j] * f * dt + g_prod(sde, t0 + tb["C1"][j] * dt, y0, gw)
            H0.append(H0i)
            f = sde.f(t0 + tb["C0"][i] * dt, H0i)
            gw = tb["beta1"][i] * I_k + tb["beta2"][i] * I_k0 * rdt
            y1 = y1 + tb["alpha"][i] * f * dt + g_prod(sde, t0 + tb["C1"][i] * dt, y0, gw)
        return y1
    tb = SRID2
    sqrt_dt = dt.sqrt()
    I_kk = (I_k ** 2 - dt) * (1 / 2)
    I_kkk = (I_k ** 3 - 3 * dt * I_k) * (1 / 6)
    y1 = y0
    H0, H1 = [], []
    for s in range(tb["STAGES"]):
        H0s, H1s = y0, y0
        for j in range(s):
            f = sde.f(t0 + tb["C0"][j] * dt, H0[j])
            g = sde.g(t0 + tb["C1"][j] * dt, H1[j])
            g = g.squeeze(2) if g.dim() == 3 else g
            H0s = H0s + tb["A0"][s][j] * f * dt + tb["B0"][s][j] * g * I_k0 * rdt
            H1s = H1s + tb["A1"][s][j] * f * dt + tb["B1"][s][j] * g * sqrt_dt
        H0.append(H0s)
        H1.append(H1s)
        f = sde.f(t0 + tb["C0"][s] * dt, H0s)
        gw = (tb["beta1"][s] * I_k + tb["beta2"][s] * I_kk / sqrt_dt + tb["beta3"][s] * I_k0 * rdt +
              tb["beta4"][s] * I_kkk * rdt)
        y1 = y1 + tb["alpha"][s] * f * dt + g_prod(sde, t0 + tb["C1"][s] * dt, H1s, gw)
    return y1


def heun_step(sde, bm, t0, t1, y0, options=None):
    """methods/heun.py:35-48."""
    dt = t1 - t0
    I_k = bm(t0, t1)
    f, gp = f_and_g_prod(sde, t0, y0, I_k)
    y0_prime = y0 + dt * f + gp
    f_prime, gp_prime = f_and_g_prod(sde, t1, y0_prime, I_k)
    return y0 + (dt * (f + f_prime) + gp + gp_prime) * 0.5


def euler_heun_step(sde, bm, t0, t1, y0, options=None):
    """methods/euler_heun.py:29-42."""
    dt = t1 - t0
    I_k = bm(t0, t1)
    f, gp = f_and_g_prod(sde, t0, y0, I_k)
    y_prime = y0 + gp
    gp_prime = g_prod(sde, t1, y_prime, I_k)
    return y0 + dt * f + (gp + gp_prime) * 0.5


def dg_ga_jvp_column_sum(sde, t, y, a):
    """base_sde.py:164-183: sum_{j,k,l} d g_{i,l} / d y_j  g_{j,k} A_{k,l} (zero unless the noise is general)."""
    if sde.noise_type != "general":
        return 0.
    with torch.enable_grad():
        y = y.detach().requires_grad_(True)
        g = sde.g(t, y)
        ga = torch.bmm(g, a)
        total = 0.
        for col in range(g.size(-1)):
            out = g[..., col]
            dummy = torch.zeros_like(out, requires_grad=True)
            back, = torch.autograd.grad(out, y, grad_outputs=dummy, create_graph=True, allow_unused=True)
            if back is None:
                continue
            jv, = torch.autograd.grad(back, dummy, grad_outputs=ga[..., col], retain_graph=True, allow_unused=True)
            total = total + (0. if jv is None else jv)
    return total.detach() if torch.is_tensor(total) else total


def log_ode_step(sde, bm, t0, t1, y0, options=None):
    """methods/log_ode.py:39-56."""
    dt = t1 - t0
    I_k, A = bm(t0, t1, return_A=True)
    f, gp = f_and_g_prod(sde, t0, y0, I_k)
    half_dt = 0.5 * dt
    t_prime = t0 + half_dt
    y_prime = y0 + half_dt * f + .5 * gp
    f_prime, gp_prime = f_and_g_prod(sde, t_prime, y_prime, I_k)
    return y0 + dt * f_prime + gp_prime + dg_ga_jvp_column_sum(sde, t_prime, y_prime, A)


STEPS = {"euler": euler_step, "midpoint": midpoint_step, "milstein": milstein_step, "srk": srk_step,
         "heun": heun_step, "euler_heun": euler_heun_step, "log_ode": log_ode_step}


# ---- the stepping loop ------------------------------------------------------------------------------------
def _update_step_size(error_estimate, prev_step_size, safety=0.9, facmin=0.2, facmax=1.4, prev_error_ratio=None):
    """adaptive_stepping.py:21-39."""
    if error_estimate > 1:
        pfactor, ifactor = 0, 1 / 1.5
    else:
        pfactor, ifactor = 0.13, 1 / 4.5
    error_ratio = safety / error_estimate
    if prev_error_ratio is None:
        prev_error_ratio = error_ratio
    factor = error_ratio ** ifactor * (error_ratio / prev_error_ratio) ** pfactor
    if error_estimate <= 1:
        prev_error_ratio = error_ratio
        facmin = 1.0
    factor = min(facmax, max(facmin, factor))
    return prev_step_size * factor, prev_error_ratio


def _compute_error(y11, y12, rtol, atol, eps=1e-7):
    """adaptive_stepping.py:42-76."""
    tol = (rtol * torch.max(torch.abs(y11), torch.abs(y12)) + atol).clamp_min(eps)
    x = (y11 - y12) / tol
    return torch.sqrt((x ** 2.).sum() / x.numel()).clamp_min(eps).item()


def integrate(sde, bm, y0, ts, dt, method, options=None, record=None, adaptive=False, rtol=1e-5, atol=1e-4,
              dt_min=1e-5):
    """base_solver.py:92-149 (fixed-step and step-doubling adaptive branches) + interp.py:15-18.
    `ts` is a CPU tensor."""
    step = STEPS[method]
    step_size = dt
    prev_t = curr_t = ts[0]
    prev_y = curr_y = y0
    ys = [y0]
    prev_error_ratio = None
    for out_t in ts[1:]:
        while curr_t < out_t:
            next_t = min(curr_t + step_size, ts[-1])
            if record is not None:
                record.append((float(curr_t), float(next_t)))
            if adaptive:
                next_y_full = step(sde, bm, curr_t, next_t, curr_y, options)
                midpoint_t = 0.5 * (curr_t + next_t)
                midpoint_y = step(sde, bm, curr_t, midpoint_t, curr_y, options)
                next_y = step(sde, bm, midpoint_t, next_t, midpoint_y, options)
                error_estimate = _compute_error(next_y_full, next_y, rtol, atol)
                step_size, prev_error_ratio = _update_step_size(error_estimate, step_size,
                                                                prev_error_ratio=prev_error_ratio)
                if step_size < dt_min:
                    step_size = dt_min
                    prev_error_ratio = None
                if error_estimate <= 1 or step_size <= dt_min:
                    prev_t, prev_y = curr_t, curr_y
                    curr_t, curr_y = next_t, next_y
            else:
                prev_t, prev_y = curr_t, curr_y
                curr_y = step(sde, bm, curr_t, next_t, curr_y, options)
                curr_t = next_t
        ys.append((curr_t - out_t) / (curr_t - prev_t) * prev_y + (out_t - prev_t) / (curr_t - prev_t) * curr_y)
    return torch.stack(ys, dim=0)


class ReplayBrownian:
    """Serves pre-computed increments keyed by the exact (t0, t1) floats of each query."""

    def __init__(self, table):
        self.table = table   # {(ta, tb): (W, U or None)}

    def __call__(self, ta, tb, return_U=False, return_A=False):
        entry = self.table[(float(ta), float(tb))]
        W, U, A = entry if len(entry) == 3 else (entry[0], entry[1], None)
        if return_U:
            return (W, U, A) if return_A else (W, U)
        return (W, A) if return_A else W


# ---- reversible Heun (carries extra state) ------------------------------------------------------------------
def reversible_heun_step(sde, bm, t0, t1, y0, extra0):
    """methods/reversible_heun.py:61-73."""
    f0, g0, z0 = extra0
    dt = t1 - t0
    dW = bm(t0, t1)
    z1 = 2 * y0 - z0 + f0 * dt + prod(sde, g0, dW)
    f1, g1 = f_and_g(sde, t1, z1)
    y1 = y0 + (f0 + f1) * (0.5 * dt) + prod(sde, g0 + g1, 0.5 * dW)
    return y1, (f1, g1, z1)


def integrate_reversible_heun(sde, bm, y0, ts, dt):
    """base_solver.py:92-116,143-149 with the solver's extra state threaded through (reversible_heun.py:58-59)."""
    extra = tuple(f_and_g(sde, ts[0], y0)) + (y0,)
    prev_t = curr_t = ts[0]
    prev_y = curr_y = y0
    ys = [y0]
    for out_t in ts[1:]:
        while curr_t < out_t:
            next_t = min(curr_t + dt, ts[-1])
            prev_t, prev_y = curr_t, curr_y
            curr_y, extra = reversible_heun_step(sde, bm, curr_t, next_t, curr_y, extra)
            curr_t = next_t
        ys.append((curr_t - out_t) / (curr_t - prev_t) * prev_y + (out_t - prev_t) / (curr_t - prev_t) * curr_y)
    return torch.stack(ys, dim=0), extra


# ---- logqp (v0.1.1 compatibility path) ------------------------------------------------------------------------
class LogqpRef:
    """base_sde.py:240-306 (`SDELogqp`): the state gains one column, the running KL integrand 0.5 |u|^2 with
    u = (f - h) / g (diagonal; misc.stable_division, misc.py:66-68) or pinv(g) (f - h) (other noise types);
    its diffusion row is zero. `names` follows base_sde.py:212-237 (`RenameMethodsSDE`)."""

    def __init__(self, sde, names=None):
        names = names or {}
        self.noise_type, self.sde_type = sde.noise_type, sde.sde_type
        self._f = getattr(sde, names.get("drift", "f"))
        self._g = getattr(sde, names.get("diffusion", "g"))
        self._h = getattr(sde, names.get("prior_drift", "h"))
        self._params = list(sde.parameters()) if hasattr(sde, "parameters") else []

    def parameters(self):
        return iter(self._params)

    def f_and_g(self, t, y):
        y = y[:, :-1]
        f, g, h = self._f(t, y), self._g(t, y), self._h(t, y)
        if self.noise_type == "diagonal":
            eps = 1e-7
            safe = torch.where(g.abs().detach() > eps, g, torch.full_like(g, fill_value=eps) * g.sign())
            u = (f - h) / safe
            g_pad = y.new_zeros(size=(y.size(0), 1))
        else:
            u = torch.bmm(g.pinverse(), (f - h).unsqueeze(-1)).squeeze(-1)
            g_pad = y.new_zeros(size=(g.size(0), 1, g.size(-1)))
        f_logqp = .5 * (u ** 2).sum(dim=1, keepdim=True)
        return torch.cat([f, f_logqp], dim=1), torch.cat([g, g_pad], dim=1)

    def f(self, t, y):
        return self.f_and_g(t, y)[0]

    def g(self, t, y):
        return self.f_and_g(t, y)[1]


def integrate_logqp(sde, bm, y0, ts, dt, method, names=None, options=None):
    """sdeint.py:142-144 (augment) + sdeint.py:284-295 (`parse_return`): returns (ys, log_ratio increments)."""
    aug = torch.cat((y0, y0.new_zeros(size=(y0.size(0), 1))), dim=1)
    ys = integrate(LogqpRef(sde, names), bm, aug, ts, dt, method, options)
    ys, log_ratio = ys.split(split_size=(y0.size(1), 1), dim=2)
    return ys, log_ratio.squeeze(dim=2)[1:] - log_ratio.squeeze(dim=2)[:-1]

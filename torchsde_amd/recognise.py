"""Recognising elementwise drift / diffusion code, so that an UNCHANGED user module reaches the trajectory kernels.

The reference's SDE contract is two Python callables (torchsde/_core/base_sde.py:24-64) that the solver calls at every
step (base_solver.py:114-149). For the diagonal-noise SDEs people actually write --

    def f(self, t, y): return self.mu * y                        # tests/problems.py:47-64 (ExDiagonal)
    def g(self, t, y): return torch.exp(-y)                      # benchmarks/brownian.py:131-139

-- both are per-channel expressions of the form ``scale * phi(rate * y + shift) + offset``, which the kernels
``tsde_trajectory_affine_diag`` / ``tsde_trajectory_expr_diag`` integrate with the state in registers, one launch per
solve. Until round 3 only modules restated as ``AffineDiagonalSDE`` / ``ElementwiseDiagonalSDE`` got there.

This module finds that form by ABSTRACT INTERPRETATION of the user's code, once per solve: ``f`` and ``g`` run on a
probe state of a few rows under a dispatch mode that follows every ATen operator touching a value derived from ``y``
and keeps, for each such value, the five per-channel coefficient tensors of the form above. Operators among tensors that
do not depend on ``y`` (parameters, buffers, Python numbers) simply execute -- on (d,)-sized data -- so the coefficients
are computed from the LIVE parameter values of this very solve: an optimiser step, a changed module global or closure
are seen because the user's code has just run. Anything else -- an operator outside the small elementwise table, a
reshape of the state, an in-place update of it, a per-row constant, `float(t)` -- ends the interpretation and the solve
takes the stepwise path, as before.

Time. With t a 0-d tensor any arithmetic on t ends the interpretation (`DependsOnTime`). The caller then interprets once
more with t = the (K, 1, 1) tensor of ALL the times at which the scheme evaluates f and g (its stage times of every
step): code that only ever broadcasts t (`beta(t) * y`, `torch.sqrt(self.b0 + t * self.b1)`) runs unchanged, values
derived from the state become (K, rows, d), coefficients (K, 1, 1) or (K, 1, d) -- one row per stage time -- and the
`_timed` trajectory kernels read those rows step by step. Code that does anything else with t (`float(t)`, `if t > 0.5`,
`cat` with the state, a reshape, a reduction) fails on the (K, 1, 1) tensor by itself or leaves these shapes, and the
solve stays stepwise.

A second form is followed the same way: a drift that is a two-layer perceptron of the state shared by the batch,
``lin2(act(lin1(y)))`` with ``act`` tanh or softplus -- ``nn.Sequential(nn.Linear, nn.Softplus, nn.Linear)`` as in the
reference's latent-SDE examples (examples/latent_sde_lorenz.py:122-128) -- with an affine or ``scale * sigmoid`` diagonal
diffusion: the forms ``tsde_trajectory_mlp_diag`` (sampling) and ``tsde_adjoint_mlp_diag`` (``sdeint_adjoint``) integrate
on the matrix cores. There the interpretation hands back the user's own parameter tensors, so that gradients reach them.

Nothing here synchronises with the host (unless the user's code does, on its own constants). Whether a recognised form
may be trusted is decided once per (form, scheme, state width, dtype) on each SDE object by solving both ways and
comparing (`solvers.BaseSDESolver._integrate_recognised`); `describe(sde)` reports what was decided.
"""
import os

import torch
from torch.utils._python_dispatch import TorchDispatchMode

from . import _native

# TSDE_RECOGNISE=0 switches the route off for the process (tests of the stepwise machinery also clear `ENABLED`)
ENABLED = os.environ.get("TSDE_RECOGNISE", "1").strip().lower() not in ("0", "false", "off")


class NotElementwise(Exception):
    """The code is not (recognisably) of the per-channel form; the message says what stopped the interpretation."""


class DependsOnTime(NotElementwise):
    """Raised by the interpretation with a scalar t when drift or diffusion use t in their arithmetic: the caller may
    interpret again with ALL step times at once (`recognise(..., times=...)`), which yields one coefficient row per step."""


class _Form:
    """``scale * phi(rate * y + shift) + offset``. A coefficient is None (the neutral element: 1 for scale and rate, 0
    for shift and offset) or a per-channel value: a Python number or a tensor of shape (), (1,), (d,) or (1, d).
    `rate is ZERO` marks a value that does not depend on y at all (``torch.ones_like(y) * sigma``): then phi is the
    identity and the value is `shift` (+ offset folded in). `exact`: the kernel's evaluation order reproduces the
    user's operations one for one (no coefficient had to be folded with another)."""
    __slots__ = ("phi", "scale", "rate", "shift", "offset", "exact")

    def __init__(self, phi="identity", scale=None, rate=None, shift=None, offset=None, exact=True):
        self.phi, self.scale, self.rate, self.shift, self.offset, self.exact = phi, scale, rate, shift, offset, exact

    def constant(self):
        return self.rate is ZERO


ZERO = object()         # the rate of a y-independent value


class _Hidden:
    """``act(x @ W1^T + b1)`` (act None: before the activation): a (rows, hidden) value of a perceptron. `x` is the state,
    or ``cat([t.expand(rows, 1), y], 1)``: then `wt` is the weight column of the time input and `w1` the state columns.
    `mids`: the (weight, bias) of hidden-to-hidden layers already passed (deeper nets: examples/sde_gan.py:50-66,
    examples/latent_sde_lorenz.py:122-135); `act_scale`: a numeric factor after the activation (LipSwish = 0.909 * silu,
    examples/sde_gan.py:44-47); `w_full`: the first layer's weight as the user's module holds it (time column included)."""
    __slots__ = ("w1", "b1", "act", "wt", "mids", "act_scale", "w_full")

    def __init__(self, w1, b1, act=None, wt=None, mids=(), act_scale=1.0, w_full=None):
        self.w1, self.b1, self.act, self.wt = w1, b1, act, wt
        self.mids, self.act_scale, self.w_full = tuple(mids), float(act_scale), (w1 if w_full is None else w_full)

    def but(self, **changes):
        new = object.__new__(_Hidden)
        for slot in self.__slots__:
            setattr(new, slot, changes.get(slot, getattr(self, slot)))
        return new


class _Perceptron:
    """``scale * final(act(... act(x @ W1^T + b1) ...) @ W2^T + b2)``: weights as the (out, in) tensors `nn.Linear` holds; `wt`
    as in `_Hidden`; `final` None, or the function applied to the last layer's output ("sigmoid": ``nn.Sigmoid()`` closing a
    diffusion net; "tanh": ``MLP(..., tanh=True)``, examples/sde_gan.py:63-64 -- or an activation whose Linear is yet to come);
    `scale` a Python number; `shape` the shape the user's code gave the result (``.view(B, d, m)``); `mids`, `act_scale`,
    `w_full` as in `_Hidden`."""
    __slots__ = ("w1", "b1", "act", "w2", "b2", "exact", "wt", "final", "scale", "shape", "mids", "act_scale", "w_full")
    phi = "perceptron"

    def __init__(self, hidden, w2, b2, shape):
        self.w1, self.b1, self.act, self.wt = hidden.w1, hidden.b1, hidden.act, hidden.wt
        self.mids, self.act_scale, self.w_full = hidden.mids, hidden.act_scale, hidden.w_full
        self.w2, self.b2, self.exact, self.final, self.scale, self.shape = w2, b2, False, None, 1.0, tuple(shape)

    @property
    def out(self):
        return self.w2.shape[0]

    def but(self, **changes):
        new = object.__new__(_Perceptron)
        for slot in self.__slots__:
            setattr(new, slot, changes.get(slot, getattr(self, slot)))
        return new

    def classic(self):
        """What the two-layer kernels (mlp_trajectory.hip, mlp_general.hip, mlp_adjoint.hip) evaluate: one hidden layer, tanh or
        softplus, at most a closing sigmoid. Anything deeper or with other functions: the reversible-Heun kernels only."""
        return (not self.mids and self.act in ("tanh", "softplus") and self.act_scale == 1.0
                and self.final in (None, "sigmoid"))

    def plain(self):
        """The form the diagonal-noise perceptron kernels take: a network of the bare state, nothing after it."""
        return self.wt is None and self.final is None and self.scale == 1.0 and self.classic()

    def as_hidden(self):
        """This net's output read as one more hidden layer (its pending `final` the activation): what a further Linear acts on."""
        return _Hidden(self.w1, self.b1, self.act, self.wt, self.mids + ((self.w2, self.b2),), self.act_scale, self.w_full)


def _mul(a, b):
    """Product of two coefficients (None = 1)."""
    if a is None:
        return b
    if b is None:
        return a
    return a * b


def _add(a, b):
    """Sum of two coefficients (None = 0)."""
    if a is None:
        return b
    if b is None:
        return a
    return a + b


def _neg(a):
    return None if a is None else -a


def _times(a, c):
    """An additive coefficient (None = 0) times c."""
    return None if a is None else a * c


# ---- cubic polynomials of the state: sums and products of affine functions (y - y**3, r * y * (1 - y / K)) -----------
# A polynomial is (c3, c2, c1, c0), each a per-channel coefficient or None (= 0). As a `_Form` it has phi "poly3" and
# keeps its coefficients in the slots (scale, rate, shift, offset) -- the order TSDE_FN_POLY3 reads them in.
def _padd(a, b):
    return b if a is None else a if b is None else a + b


def _pmul(a, b):
    return None if a is None or b is None else a * b


def _as_poly(form):
    """(c3, c2, c1, c0) of a form that is a polynomial of the state (a plain `rate * y + shift`, or a cubic), else None."""
    if form.constant():
        return None
    if form.phi == "poly3":
        return (form.scale, form.rate, form.shift, form.offset)
    if form.phi == "identity" and form.scale is None and form.offset is None:
        return (None, None, 1.0 if form.rate is None else form.rate, form.shift)
    return None


def _poly_form(c):
    return _Form("poly3", c[0], c[1], c[2], c[3], exact=False)


def _poly_product(p, q):
    """Product of two polynomials given as (c3, c2, c1, c0); NotElementwise beyond degree 3."""
    out = [None] * 7                                      # coefficient of y^k at index k
    for i, a in enumerate(reversed(p)):                   # i = power of a
        for j, b in enumerate(reversed(q)):
            term = _pmul(a, b)
            if term is not None:
                out[i + j] = _padd(out[i + j], term)
    if any(c is not None for c in out[4:]):
        raise NotElementwise("a polynomial of the state of degree above 3")
    return (out[3], out[2], out[1], out[0])


class _Interpreter(TorchDispatchMode):
    def __init__(self, y, t, rows, d, steps=None):
        super().__init__()
        self.rows, self.d, self.steps = rows, d, steps        # steps = K: t is the (K, 1, 1) tensor of all step times
        self.forms = {id(y): _Form()}
        self.time = {id(t)}
        self.keep = [y, t]              # every tracked tensor stays alive: ids are not reused during the run
        self.seen = set()               # ids of every tensor the USER's code handed to an operator (kept alive): a
        #                                 coefficient among them is the user's own tensor -- a parameter, or something
        #                                 their code computed from parameters with autograd watching -- not one folded here
        self.hidden = {}                # id -> _Hidden: (rows, hidden)-shaped values of a perceptron drift
        self.transposed = {}            # id of `W.t()` -> W (nn.Linear hands addmm the transposed view of its weight)
        self.raw_time = {id(t)}         # t itself and value-preserving copies of it (`t.expand(B, 1)`, `.to(y.dtype)`)
        self.time_state = set()         # ids of `cat([t.expand(rows, 1), y], 1)`: the input of the reference's Neural* nets
        self.nets = {}                  # id -> _Perceptron for network outputs that are not state-shaped ((rows, d * m), ...)
        self.made = {id(y), id(t)}      # ids of tensors whose storage was allocated during the interpretation (kept alive):
        #                                 an in-place write to any OTHER tensor changes state that outlives the probe call
        self.differentiable = False     # True: the kernel's GRADIENTS will stand for autograd through this code, so the probe
        #                                 requires grad and a value derived from the state that does not is a stop-gradient

    # ---- stop-gradients --------------------------------------------------------------------------------------------
    def depends_on_state(self, x):
        """Is the tracked tensor `x` a non-constant function of the state?"""
        if id(x) in self.hidden or id(x) in self.nets or id(x) in self.time_state:
            return True
        form = self.forms.get(id(x))
        if form is None:
            return False
        if isinstance(form, _Form):
            return not form.constant()
        if isinstance(form, _Expr):
            return form.mentions_state()
        return True

    def check_stop_gradient(self, tensors, where="an operand"):
        """With autograd recording the solve, the reference differentiates the user's code as it stands (ordinary autograd
        through base_solver.py:143-149; adjoint_sde.py:111-128 for `sdeint_adjoint`): `y.detach() * mu`, `sigma * y.data` or
        state arithmetic inside `torch.no_grad()` are stop-gradients there, while the sensitivity / adjoint kernels would
        differentiate the recognised expression straight through them. The probe requires grad (`differentiable=True`), so a
        floating-point value derived from the state that does NOT is the trace of such a construct: the interpretation ends."""
        if not self.differentiable:
            return
        for a in tensors:
            if torch.is_tensor(a) and a.is_floating_point() and not a.requires_grad and self.depends_on_state(a):
                raise NotElementwise(f"{where} derived from the state carries no gradient (detach, .data or torch.no_grad() "
                                     "in the user's code): a stop-gradient the kernels' derivatives would ignore")

    # ---- bookkeeping ---------------------------------------------------------------------------------------------
    def form_of(self, x):
        return self.forms.get(id(x)) if torch.is_tensor(x) else None

    def state_shaped(self, tensor):
        shape = tuple(tensor.shape)
        return shape == (self.rows, self.d) or (self.steps is not None and shape == (self.steps, self.rows, self.d))

    def time_shaped(self, tensor):
        return self.steps is not None and tuple(tensor.shape) in ((self.steps, 1, 1), (self.steps, 1, self.d))

    def track(self, tensor, form):
        if not self.state_shaped(tensor):
            raise NotElementwise(f"a value derived from the state has shape {tuple(tensor.shape)}")
        self.forms[id(tensor)] = form
        self.keep.append(tensor)
        return tensor

    def coefficient(self, c):
        """`c` as a per-channel coefficient: a Python number, or a tensor that broadcasts over the rows."""
        if isinstance(c, (bool, int, float)):
            return c
        if not torch.is_tensor(c):
            raise NotElementwise(f"an operand of type {type(c).__name__}")
        if id(c) in self.time:
            if self.steps is None:
                raise DependsOnTime("drift or diffusion depends on t")
            if not self.time_shaped(c):
                raise NotElementwise(f"a function of t of shape {tuple(c.shape)} is not one value per step and channel")
            return c
        shape = tuple(c.shape)
        if shape not in ((), (1,), (self.d,), (1, self.d), (1, 1)):
            raise NotElementwise(f"an operand of shape {shape} is not one value per channel")
        return c

    # ---- the algebra ---------------------------------------------------------------------------------------------
    def scaled(self, x, c, exact=True):
        """x * c for a per-channel c."""
        if x.phi == "poly3":
            return _poly_form(tuple(_pmul(k, c) for k in _as_poly(x)))
        if x.constant():
            return _Form(rate=ZERO, shift=_times(_add(x.shift, x.offset), c), exact=x.exact)
        if x.phi == "identity" and x.scale is None and x.offset is None:
            if x.rate is None and x.shift is None:
                return _Form(rate=c, exact=x.exact and exact)
            return _Form(rate=_mul(x.rate, c), shift=_times(x.shift, c), exact=False)
        plain = x.scale is None and x.offset is None
        return _Form(x.phi, _mul(x.scale, c), x.rate, x.shift, _times(x.offset, c), exact=x.exact and plain and exact)

    def shifted(self, x, c):
        """x + c for a per-channel c."""
        if x.phi == "poly3":
            p = _as_poly(x)
            return _poly_form((p[0], p[1], p[2], _padd(p[3], c)))
        if x.constant():
            return _Form(rate=ZERO, shift=_add(_add(x.shift, x.offset), c), exact=x.exact)
        if x.phi == "identity" and x.scale is None and x.offset is None:
            return _Form(rate=x.rate, shift=_add(x.shift, c), exact=x.exact and x.shift is None)
        return _Form(x.phi, x.scale, x.rate, x.shift, _add(x.offset, c), exact=x.exact and x.offset is None)

    def negated(self, x):
        if x.phi == "poly3":
            return _poly_form(tuple(_neg(k) for k in _as_poly(x)))
        if x.constant():
            return _Form(rate=ZERO, shift=_neg(_add(x.shift, x.offset)), exact=x.exact)
        if x.phi == "identity" and x.scale is None and x.offset is None:
            return _Form(rate=-1.0 if x.rate is None else -x.rate, shift=_neg(x.shift), exact=x.exact)
        return _Form(x.phi, -1.0 if x.scale is None else -x.scale, x.rate, x.shift, _neg(x.offset), exact=x.exact)

    def summed(self, x, z, sign=1.0):
        """x + sign * z for two tracked values."""
        if z.constant():
            c = _add(z.shift, z.offset)
            return self.shifted(x, c if sign == 1.0 else _neg(c)) if c is not None else x
        if x.constant():
            c = _add(x.shift, x.offset)
            z = z if sign == 1.0 else self.negated(z)
            return self.shifted(z, c) if c is not None else z
        p, q = _as_poly(x), _as_poly(z)
        if p is not None and q is not None and "poly3" in (x.phi, z.phi):
            if sign != 1.0:
                q = tuple(_neg(k) for k in q)
            return _poly_form(tuple(_padd(a, b) for a, b in zip(p, q)))
        if x.phi == "identity" and z.phi == "identity" and all(v.scale is None and v.offset is None for v in (x, z)):
            one = 1.0
            zr, zs = (one if z.rate is None else z.rate), z.shift
            if sign != 1.0:
                zr, zs = -zr, _neg(zs)
            return _Form(rate=(one if x.rate is None else x.rate) + zr, shift=_add(x.shift, zs), exact=False)
        raise NotElementwise("a sum of two different functions of the state")

    def product(self, x, z):
        if z.constant():
            c = _add(z.shift, z.offset)
            return self.scaled(x, 0.0 if c is None else c)
        if x.constant():
            c = _add(x.shift, x.offset)
            return self.scaled(z, 0.0 if c is None else c)
        p, q = _as_poly(x), _as_poly(z)
        if p is not None and q is not None:
            return _poly_form(_poly_product(p, q))
        raise NotElementwise("a product of two functions of the state that are not both polynomials")

    def applied(self, name, x):
        if x.constant():
            raise NotElementwise(f"{name} of a constant")         # (legal, rare; leave it to the stepwise path)
        if x.phi != "identity" or x.scale is not None or x.offset is not None:
            raise NotElementwise(f"{name} of {x.phi}: nested functions")
        return _Form(name, None, x.rate, x.shift, None, exact=x.exact)

    # ---- a perceptron drift: lin2(act(lin1(y))) ------------------------------------------------------------------
    def weight_of(self, operand, name):
        """The (out, in) weight behind the second operand of a product `x @ operand` (nn.Linear passes `W.t()`)."""
        if not torch.is_tensor(operand) or operand.dim() != 2 or id(operand) in self.forms or id(operand) in self.hidden:
            raise NotElementwise(f"{name} with an operand that is not a weight matrix")
        if id(operand) in self.time:
            raise NotElementwise("the coefficients depend on t")
        w = self.transposed.get(id(operand))
        return w if w is not None else operand.t()

    def layer_operands(self, name, args):
        """(input, weight (out, in), bias or None) of addmm(bias, x, Wt) / mm(x, Wt) / linear(x, W, bias)."""
        if name == "addmm":
            if len(args) != 3:
                raise NotElementwise("addmm with beta / alpha")
            return args[1], self.weight_of(args[2], name), args[0]
        if name == "mm":
            return args[0], self.weight_of(args[1], name), None
        w = args[1]
        if not torch.is_tensor(w) or w.dim() != 2:
            raise NotElementwise("linear with an operand that is not a weight matrix")
        return args[0], w, (args[2] if len(args) > 2 else None)

    def bias_of(self, b, width):
        if b is None:
            return None
        if not torch.is_tensor(b) or id(b) in self.time or id(b) in self.forms or id(b) in self.hidden \
                or tuple(b.shape) != (width,):
            raise NotElementwise("a layer bias that is not one value per output channel")
        return b

    def first_layer(self, name, args, out):
        x, w, b = self.layer_operands(name, args)
        wt = None
        w_full = w
        if torch.is_tensor(x) and id(x) in self.time_state:
            # the first layer of a net that is fed cat([t, y]): column 0 of its weight multiplies t
            if tuple(w.shape)[1:] != (self.d + 1,) or tuple(out.shape) != (self.rows, w.shape[0]):
                raise NotElementwise("a matrix product that does not act on [t, state channels]")
            wt, w = w[:, 0], w[:, 1:]
        else:
            form = self.form_of(x)
            if form is None or isinstance(form, _Perceptron) or form.constant() or form.phi != "identity" or any(
                    c is not None for c in (form.scale, form.rate, form.shift, form.offset)):
                raise NotElementwise("a matrix product of something other than the state itself")
            if tuple(w.shape)[1:] != (self.d,) or tuple(out.shape) != (self.rows, w.shape[0]):
                raise NotElementwise("a matrix product that does not act on the state channels")
        self.hidden[id(out)] = _Hidden(w, self.bias_of(b, w.shape[0]), wt=wt, w_full=w_full)
        self.keep.append(out)
        return out

    def time_state_cat(self, func, args, kwargs):
        """``torch.cat([t.expand(rows, 1), y], dim=1)`` -- how every Neural* problem of the reference feeds t to its nets
        (tests/problems.py:153-159, 183-189, 215-217, 246-252)."""
        tensors = args[0] if args else ()
        dim = args[1] if len(args) > 1 else kwargs.get("dim", 0)
        if not isinstance(tensors, (list, tuple)) or len(tensors) != 2 or dim not in (1, -1):
            raise NotElementwise("cat of t with the state other than [t, y] along the channels")
        tc, yv = tensors
        form = self.form_of(yv)
        if (id(tc) not in self.raw_time or tuple(tc.shape) != (self.rows, 1) or form is None or isinstance(form, _Perceptron)
                or form.constant() or form.phi != "identity"
                or any(c is not None for c in (form.scale, form.rate, form.shift, form.offset))):
            raise NotElementwise("cat of t with the state other than [t.expand(rows, 1), y]")
        out = func(*args, **kwargs)
        self.time_state.add(id(out))
        self.keep.append(out)
        return out

    def register_net(self, out, net):
        """Remember `out` as the value of perceptron `net`: among the tracked state-shaped values, or (other shapes) the nets."""
        if not torch.is_tensor(out) or out.dim() < 2 or out.shape[0] != self.rows or out.numel() != self.rows * net.out:
            raise NotElementwise(f"the output of a network was given shape {tuple(getattr(out, 'shape', ()))}")
        net = net.but(shape=tuple(out.shape))
        self.keep.append(out)
        if self.state_shaped(out):
            self.forms[id(out)] = net
        else:
            self.nets[id(out)] = net
        return out

    def net_of(self, x):
        if not torch.is_tensor(x):
            return None
        form = self.forms.get(id(x))
        return form if isinstance(form, _Perceptron) else self.nets.get(id(x))

    _RESHAPES = ("view", "reshape", "_unsafe_view", "_reshape_alias", "unsqueeze", "squeeze")

    def output_step(self, func, args, kwargs):
        """An operator applied to the output of a network: the closing sigmoid, a numeric factor, the reshape to
        (rows, d, m). Anything else ends the interpretation."""
        schema = func._schema
        name = schema.name.split("::")[1]
        if schema.is_mutable:
            raise NotElementwise(f"in-place {name} on the output of a network")
        self.check_kwargs(name, schema, kwargs)
        net = self.net_of(args[0]) if args else None
        if name == "mul" and len(args) == 2:
            other = args[1] if net is not None else args[0]
            net = net if net is not None else self.net_of(args[1])
            if self.net_of(other) is not None or self.form_of(other) is not None:
                raise NotElementwise("a product of the output of a network with a function of the state")
            if torch.is_tensor(other) and (other.dim() != 0 or other.device.type != "cpu"):
                raise NotElementwise("the output of a network times something that is not a plain number")
            out = func(*args, **kwargs)
            return self.register_net(out, net.but(scale=net.scale * float(other)))
        if name in ("addmm", "mm", "linear"):
            # one more Linear on what so far looked like a net's output: that output was a hidden layer (depth > 2)
            x, w, b = self.layer_operands(name, args)
            net = self.net_of(x)
            if net is None or self.net_of(w) is not None or (b is not None and self.net_of(b) is not None):
                raise NotElementwise("a matrix product whose input is not the output of the layer before")
            if net.final not in ("tanh", "softplus", "silu") or net.final != net.act or net.scale != net.act_scale \
                    or tuple(net.shape) != (self.rows, net.out) or net.out != net.w1.shape[0]:
                raise NotElementwise("a further layer after a net's output: hidden layers must share one width, one activation "
                                     "and one factor")
            out = func(*args, **kwargs)
            hidden = net.as_hidden()
            if tuple(w.shape)[1:] != (net.out,) or tuple(out.shape) != (self.rows, w.shape[0]):
                raise NotElementwise("a layer that does not act on the hidden units")
            return self.register_net(out, _Perceptron(hidden, w, self.bias_of(b, w.shape[0]), out.shape))
        if net is None:
            raise NotElementwise(f"{name} applied to the output of the drift network")
        out = func(*args, **kwargs)
        if name in ("sigmoid", "tanh", "softplus", "silu") and net.final is None and net.scale == 1.0 \
                and (len(args) == 1 or name == "softplus"):
            if name == "softplus":
                beta = args[1] if len(args) > 1 else kwargs.get("beta", 1)
                threshold = args[2] if len(args) > 2 else kwargs.get("threshold", 20)
                if beta != 1 or threshold != 20:
                    raise NotElementwise("softplus with a non-default beta or threshold")
            return self.register_net(out, net.but(final=name))
        if name in self._SAME or name in self._RESHAPES:
            if not torch.is_tensor(out) or out.dtype != args[0].dtype or out.device != args[0].device:
                raise NotElementwise(f"{name} changes the dtype or device of the output of a network")
            return self.register_net(out, net)
        raise NotElementwise(f"{name} applied to the output of the drift network")

    def perceptron_step(self, func, args, kwargs):
        schema = func._schema
        name = schema.name.split("::")[1]
        if schema.is_mutable:
            raise NotElementwise(f"in-place {name} inside the drift network")
        out = func(*args, **kwargs)
        h = self.hidden.get(id(args[0])) if args and torch.is_tensor(args[0]) else None
        if name in ("tanh", "softplus", "silu") and h is not None and h.act is None and len(args) >= 1 \
                and (name == "softplus" or len(args) == 1):
            if name == "softplus":
                beta = args[1] if len(args) > 1 else kwargs.get("beta", 1)
                threshold = args[2] if len(args) > 2 else kwargs.get("threshold", 20)
                if beta != 1 or threshold != 20:
                    raise NotElementwise("softplus with a non-default beta or threshold")
            self.hidden[id(out)] = h.but(act=name)
            self.keep.append(out)
            return out
        if name == "mul" and len(args) == 2 and (h is not None or self.hidden.get(id(args[1])) is not None):
            # a numeric factor after a hidden activation (LipSwish: 0.909 * silu(x), examples/sde_gan.py:44-47)
            hv, other = (h, args[1]) if h is not None else (self.hidden.get(id(args[1])), args[0])
            if hv.act is None or hv.act_scale != 1.0:
                raise NotElementwise("a factor inside a network before its activation (or a second one after it)")
            if torch.is_tensor(other) and (other.dim() != 0 or other.device.type != "cpu" or id(other) in self.time):
                raise NotElementwise("a hidden layer times something that is not a plain number")
            self.hidden[id(out)] = hv.but(act_scale=float(other))
            self.keep.append(out)
            return out
        if name in ("addmm", "mm", "linear"):
            x, w, b = self.layer_operands(name, args)
            h = self.hidden.get(id(x))
            if h is None or h.act is None:
                raise NotElementwise("a second layer without an activation before it")
            if tuple(w.shape)[1:] != (h.w1.shape[0],) or tuple(out.shape) != (self.rows, w.shape[0]):
                raise NotElementwise("a second layer that does not act on the hidden units")
            return self.register_net(out, _Perceptron(h, w, self.bias_of(b, w.shape[0]), out.shape))
        if name in self._SAME and h is not None and torch.is_tensor(out) and out.shape == args[0].shape \
                and out.dtype == args[0].dtype:
            self.hidden[id(out)] = h
            self.keep.append(out)
            return out
        raise NotElementwise(f"operator {schema.name} inside the drift network")

    # ---- the dispatch hook ---------------------------------------------------------------------------------------
    _UNARY = {"exp": "exp", "sigmoid": "sigmoid", "tanh": "tanh", "sin": "sin", "cos": "cos"}
    _SAME = {"alias", "detach", "clone", "lift_fresh", "positive", "contiguous", "_to_copy", "view", "reshape",
             "_unsafe_view", "expand", "_reshape_alias"}
    _LIKE = {"zeros_like": 0.0, "ones_like": 1.0}
    _TIME_COPIES = ("expand", "view", "reshape", "_unsafe_view", "_reshape_alias", "unsqueeze", "squeeze", "alias", "detach",
                    "clone", "contiguous", "_to_copy", "lift_fresh")

    # kwargs whose effect the handlers model; every other non-default kwarg of an operator on a tracked value ends the
    # interpretation (`torch.div(y, 2, rounding_mode="floor")` is not `0.5 * y`)
    _MODELLED = {"add": ("alpha",), "sub": ("alpha",), "rsub": ("alpha",), "softplus": ("beta", "threshold"),
                 "clamp": ("min", "max")}
    # ... and kwargs whose effect is checked on the RESULT (shape, dtype and device of the output must be the input's)
    _CHECKED_ON_RESULT = ("dtype", "layout", "device", "pin_memory", "memory_format", "non_blocking", "copy", "implicit")

    def check_kwargs(self, name, schema, kwargs):
        allowed = self._MODELLED.get(name, ())
        result_checked = name in self._SAME or name in self._LIKE or name == "full_like"
        for key, value in kwargs.items():
            if value is None or key in allowed or (result_checked and key in self._CHECKED_ON_RESULT):
                continue
            if not torch.is_tensor(value) and any(a.name == key and a.has_default_value() and a.default_value == value
                                                  for a in schema.arguments):
                continue
            raise NotElementwise(f"{name} with {key}={value!r}: an argument the interpretation does not model")

    def check_side_effects(self, func, args, kwargs):
        """The interpretation CALLS the user's code once per solve where the reference calls it once per step
        (base_solver.py:114-149): code whose calls leave something behind cannot take this route. Random draws
        (dropout, `randn_like`) and in-place writes to tensors that existed before the call (a buffer bumped with
        `.add_`, spectral norm's power iteration) end the interpretation."""
        schema = func._schema
        if torch.Tag.nondeterministic_seeded in func.tags:
            raise NotElementwise(f"{schema.name} draws random numbers: once per solve here, once per step in the reference")
        if not schema.is_mutable:
            return
        for i, arg in enumerate(schema.arguments):
            if arg.alias_info is None or not arg.alias_info.is_write:
                continue
            value = args[i] if i < len(args) else kwargs.get(arg.name)
            for x in (value if isinstance(value, (list, tuple)) else (value,)):
                if torch.is_tensor(x) and id(x) not in self.made:
                    raise NotElementwise(f"in-place {schema.name} on a tensor that existed before f and g were called "
                                         "(state that outlives the call: once per solve here, once per step in the reference)")

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        self.check_side_effects(func, args, kwargs)
        out = self.interpret(func, args, kwargs)
        # storage made during the interpretation: fresh results, and views (aliasing returns) of such tensors
        returns = func._schema.returns
        outs = out if isinstance(out, (list, tuple)) else (out,)
        for k, o in enumerate(outs):
            if not torch.is_tensor(o) or id(o) in self.made:
                continue
            aliasing = k < len(returns) and returns[k].alias_info is not None and func._schema.name != "aten::lift_fresh"
            if not aliasing or (args and torch.is_tensor(args[0]) and id(args[0]) in self.made):
                self.made.add(id(o))
                self.keep.append(o)
        return out

    def interpret(self, func, args, kwargs):
        flat = list(args) + list(kwargs.values())
        involved = [a for a in flat if torch.is_tensor(a)]
        for a in flat:
            if isinstance(a, (list, tuple)):
                involved.extend(x for x in a if torch.is_tensor(x))
        for a in involved:
            if id(a) not in self.seen:
                self.seen.add(id(a))
                self.keep.append(a)
        self.check_stop_gradient(involved)
        timed = any(id(a) in self.time for a in involved)
        if timed and not any(id(a) in self.forms or id(a) in self.hidden or id(a) in self.nets or id(a) in self.time_state
                             for a in involved):
            # arithmetic among t and constants: executes; its results are functions of t
            name = func._schema.name
            if name in ("aten::_local_scalar_dense", "aten::item"):
                raise NotElementwise("the code reads t on the host")
            out = func(*args, **kwargs)
            for o in (out if isinstance(out, (list, tuple)) else (out,)):
                if torch.is_tensor(o):
                    if self.steps is not None and not self.time_shaped(o):
                        raise NotElementwise(f"{name} makes a function of t of shape {tuple(o.shape)}: not one value per "
                                             "step and channel")
                    self.time.add(id(o))
                    self.keep.append(o)
            # t itself, copied without arithmetic (`t.expand(B, 1)`, `.to(y.dtype)`): still "the time"
            if (name.split("::")[1] in self._TIME_COPIES and torch.is_tensor(out) and args and torch.is_tensor(args[0])
                    and id(args[0]) in self.raw_time and out.is_floating_point() and len(involved) == 1):
                self.raw_time.add(id(out))
            return out
        if timed and self.steps is None and func._schema.name == "aten::cat":
            return self.time_state_cat(func, args, kwargs)
        if timed and self.steps is None:
            raise DependsOnTime("drift or diffusion depends on t")
        if timed and any(id(a) in self.hidden or id(a) in self.nets or id(a) in self.time_state for a in involved):
            raise NotElementwise("the drift network depends on t")
        if any(id(a) in self.hidden for a in involved):
            self.check_kwargs(func._schema.name.split("::")[1], func._schema, kwargs)
            return self.perceptron_step(func, args, kwargs)
        if any(id(a) in self.time_state for a in involved):
            name = func._schema.name.split("::")[1]
            if name not in ("addmm", "mm", "linear") or func._schema.is_mutable:
                raise NotElementwise(f"{name} applied to cat([t, y]): only a first layer may take it")
            self.check_kwargs(name, func._schema, kwargs)
            return self.first_layer(name, args, func(*args, **kwargs))
        if any(self.net_of(a) is not None for a in involved):
            return self.output_step(func, args, kwargs)
        tracked = [a for a in involved if id(a) in self.forms]
        if not tracked:
            out = func(*args, **kwargs)
            if func._schema.name in ("aten::t", "aten::transpose") and torch.is_tensor(out) and out.dim() == 2 \
                    and args[0].dim() == 2:
                self.transposed[id(out)] = args[0]
                self.keep.append(out)
            # a y-independent value stretched over the probe's rows (`sigma.expand_as(y)`, `sigma.expand(B, d)`)
            if func._schema.name == "aten::expand" and torch.is_tensor(out) and tuple(out.shape) == (self.rows, self.d) \
                    and id(args[0]) not in self.time:
                src = args[0]
                if src.dim() <= 2 and (src.dim() < 2 or src.shape[0] == 1):
                    return self.track(out, self.constant_value(self.coefficient(src if src.dim() < 2 else src[0])))
            return out
        return self.tracked_op(func, args, kwargs)

    def constant_value(self, c):
        """The tracked value of something that does not depend on the state (`sigma.expand_as(y)`, `torch.ones_like(y)`)."""
        return _Form(rate=ZERO, shift=c)

    def tracked_op(self, func, args, kwargs):
        """An operator with at least one operand derived from the state: the single-function algebra (`_Form`)."""
        schema = func._schema
        name = schema.name.split("::")[1]
        if schema.is_mutable:
            raise NotElementwise(f"in-place {name} on a value derived from the state")
        self.check_kwargs(name, schema, kwargs)
        out = func(*args, **kwargs)
        if name in ("addmm", "mm", "linear"):
            return self.first_layer(name, args, out)
        x = self.form_of(args[0]) if args else None
        if name in self._LIKE or name in ("full_like", "empty_like", "new_zeros", "new_ones", "new_full", "new_empty"):
            if name in self._LIKE and self.state_shaped(out) and out.dtype == args[0].dtype:
                return self.track(out, _Form(rate=ZERO, shift=self._LIKE[name] or None))
            if name == "full_like" and self.state_shaped(out) and out.dtype == args[0].dtype \
                    and isinstance(args[1], (int, float)):
                return self.track(out, _Form(rate=ZERO, shift=args[1]))
            raise NotElementwise(f"{name} of the state")
        if name in self._SAME:
            if x is None or not torch.is_tensor(out) or out.shape != args[0].shape \
                    or out.dtype != args[0].dtype or out.device != args[0].device:
                raise NotElementwise(f"{name} changes the shape, dtype or device of a value derived from the state")
            return self.track(out, x)
        if name in self._UNARY and x is not None and len(args) == 1:
            return self.track(out, self.applied(self._UNARY[name], x))
        if name == "softplus" and x is not None:
            beta = args[1] if len(args) > 1 else kwargs.get("beta", 1)
            threshold = args[2] if len(args) > 2 else kwargs.get("threshold", 20)
            if beta != 1 or threshold != 20:
                raise NotElementwise("softplus with a non-default beta or threshold")
            return self.track(out, self.applied("softplus", x))
        if name == "neg" and x is not None:
            return self.track(out, self.negated(x))
        if name in ("mul", "add", "sub", "rsub", "div") and len(args) >= 2:
            a, b = args[0], args[1]
            fa, fb = self.form_of(a), self.form_of(b)
            alpha = kwargs.get("alpha", 1)
            if name == "rsub":                          # rsub(a, b) = b - alpha * a
                a, b, fa, fb, name = b, a, fb, fa, "sub"
            if alpha != 1:
                if not isinstance(alpha, (int, float)):
                    raise NotElementwise("a tensor-valued alpha")
                if fb is not None:
                    fb = self.scaled(fb, alpha, exact=False)
                else:
                    b = self.coefficient(b) * alpha
            if name == "mul":
                form = self.product(fa, fb) if fa is not None and fb is not None else \
                    self.scaled(fa, self.coefficient(b)) if fa is not None else self.scaled(fb, self.coefficient(a))
            elif name == "div":
                if fb is not None:
                    raise NotElementwise("a division by a function of the state")
                c = self.coefficient(b)
                form = self.scaled(fa, 1.0 / c, exact=False)
            elif name == "add":
                form = self.summed(fa, fb) if fa is not None and fb is not None else \
                    self.shifted(fa, self.coefficient(b)) if fa is not None else self.shifted(fb, self.coefficient(a))
            else:       # sub
                if fa is not None and fb is not None:
                    form = self.summed(fa, fb, sign=-1.0)
                elif fa is not None:
                    form = self.shifted(fa, -self.coefficient(b))
                else:
                    form = self.shifted(self.negated(fb), self.coefficient(a))
            return self.track(out, form)
        if name == "pow" and x is not None and len(args) == 2 and isinstance(args[1], (int, float)) and args[1] == 1:
            return self.track(out, x)
        if (name == "pow" and x is not None and len(args) == 2 and isinstance(args[1], (int, float))
                and args[1] in (2, 3)) or (name == "square" and x is not None):
            p = _as_poly(x)
            if p is None:
                raise NotElementwise(f"a power of {x.phi}")
            q = p
            for _ in range(1 if name == "square" else int(args[1]) - 1):
                q = _poly_product(q, p)
            return self.track(out, _poly_form(q))
        raise NotElementwise(f"operator {schema.name} on a value derived from the state")


# ---- expression programs: any elementwise code ---------------------------------------------------------------------------
class _Expr:
    """A node of the expression tree of a value derived from the state: op "y" (the state), "const" (`value`: a number or a
    per-channel tensor), a unary function or a binary operator (`args`). Duck-typed like `_Form` where the interpreter asks."""
    __slots__ = ("op", "args", "value", "trailing")
    phi, exact = "program", False

    def __init__(self, op, args=(), value=None, trailing=False):
        self.op, self.args, self.value, self.trailing = op, tuple(args), value, trailing

    def constant(self):
        return False

    def leaf(self):
        return self.op in ("y", "t", "const")

    def mentions_state(self):
        return self.op == "y" or any(a.mentions_state() for a in self.args)


_OPCODES = {"load": 0, "add": 1, "sub": 2, "rsub": 3, "mul": 4, "div": 5, "rdiv": 6, "neg": 16, "exp": 17, "log": 18, "sin": 19,
            "cos": 20, "tanh": 21, "sigmoid": 22, "softplus": 23, "sqrt": 24, "abs": 25, "relu": 26, "reciprocal": 27,
            "square": 28, "cube": 29, "dup": 30}          # include/torchsde_amd.h: tsde_trajectory_prog_diag
_SRC_STACK, _SRC_CONST, _SRC_STATE, _SRC_TIME = 0, 1, 2, 3
_REVERSED = {"add": "add", "mul": "mul", "sub": "rsub", "div": "rdiv"}
_STACK_DEPTH = 4


class _Program:
    """Postfix code for one expression tree: `words` (ints as the C ABI wants them) referring to rows of `consts`."""

    def __init__(self, consts):
        self.words, self.consts = [], consts

    def const_row(self, value):
        for k, have in enumerate(self.consts):
            if have is value or (not torch.is_tensor(value) and not torch.is_tensor(have) and have == value):
                return k
        if len(self.consts) >= 64:
            raise NotElementwise("more than 64 constants in drift and diffusion")
        self.consts.append(value)
        return len(self.consts) - 1

    def emit(self, op, src=_SRC_STACK, k=0):
        self.words.append(_OPCODES[op] | (src << 8) | (k << 16))

    @staticmethod
    def need(node):
        """Stack slots the evaluation of `node` occupies at its peak (Sethi-Ullman numbering; leaves are operands)."""
        if node.leaf():
            return 1
        if len(node.args) == 1:
            return _Program.need(node.args[0])
        a, b = node.args
        if b.leaf():
            return _Program.need(a)
        if a.leaf():
            return _Program.need(b)
        na, nb = _Program.need(a), _Program.need(b)
        return max(na, nb) + 1 if na == nb else max(na, nb)

    def source(self, leaf):
        if leaf.op == "y":
            return (_SRC_STATE, 0)
        if leaf.op == "t":
            return (_SRC_TIME, 0)
        return (_SRC_CONST, self.const_row(leaf.value))

    def compile(self, node):
        """Append the code that leaves the value of `node` on top of the stack."""
        if node.leaf():
            self.emit("load", *self.source(node))
        elif len(node.args) == 1:
            self.compile(node.args[0])
            self.emit(node.op)
        else:
            a, b = node.args
            if b.leaf():
                self.compile(a)
                self.emit(node.op, *self.source(b))
            elif a.leaf():
                self.compile(b)
                self.emit(_REVERSED[node.op], *self.source(a))
            elif self.need(a) >= self.need(b):
                self.compile(a)
                self.compile(b)
                self.emit(node.op)                      # (below the top) op (top)
            else:
                self.compile(b)
                self.compile(a)
                self.emit(_REVERSED[node.op])           # (top) op (below the top)


def _is_number(node, value):
    return node.op == "const" and not torch.is_tensor(node.value) and node.value == value


def _expr(op, *args):
    """A node with the obvious simplifications (the derivative trees are full of 0 and 1)."""
    if op == "mul":
        a, b = args
        if _is_number(a, 0) or _is_number(b, 0):
            return _Expr("const", value=0.0)
        if _is_number(a, 1):
            return b
        if _is_number(b, 1):
            return a
    if op == "add":
        a, b = args
        if _is_number(a, 0):
            return b
        if _is_number(b, 0):
            return a
    if op == "sub" and _is_number(args[1], 0):
        return args[0]
    if op == "neg" and args[0].op == "const" and not torch.is_tensor(args[0].value):
        return _Expr("const", value=-args[0].value)
    return _Expr(op, args)


def _number(v):
    return _Expr("const", value=float(v))


def _derivative(node):
    """d node / d y as a tree (what autograd computes for `g` in derivative-form Milstein, base_sde.py:147-152)."""
    op = node.op
    if op == "y":
        return _number(1)
    if op in ("const", "t"):
        return _number(0)
    if len(node.args) == 1:
        u = node.args[0]
        du = _derivative(u)
        if op == "neg":
            return _expr("neg", du)
        outer = {
            "exp": lambda: node,
            "log": lambda: _expr("reciprocal", u),
            "sin": lambda: _expr("cos", u),
            "cos": lambda: _expr("neg", _expr("sin", u)),
            "tanh": lambda: _expr("sub", _number(1), _expr("square", node)),
            "sigmoid": lambda: _expr("mul", node, _expr("sub", _number(1), node)),
            "softplus": lambda: _expr("sigmoid", u),
            "sqrt": lambda: _expr("div", _number(0.5), node),
            "reciprocal": lambda: _expr("neg", _expr("square", node)),
            "square": lambda: _expr("mul", _number(2), u),
            "cube": lambda: _expr("mul", _number(3), _expr("square", u)),
        }.get(op)
        if outer is None:
            raise NotElementwise(f"no derivative rule for {op} (Milstein needs the diffusion's derivative)")
        return _expr("mul", outer(), du)
    a, b = node.args
    da, db = _derivative(a), _derivative(b)
    if op == "add":
        return _expr("add", da, db)
    if op == "sub":
        return _expr("sub", da, db) if not _is_number(da, 0) else _expr("neg", db)
    if op == "mul":
        return _expr("add", _expr("mul", da, b), _expr("mul", a, db))
    if op == "div":
        if _is_number(db, 0):
            return _expr("div", da, b)
        return _expr("div", _expr("sub", _expr("mul", da, b), _expr("mul", a, db)), _expr("square", b))
    raise NotElementwise(f"no derivative rule for {op}")


class _TreeInterpreter(_Interpreter):
    """The same walk over the user's code, keeping the whole expression TREE of every value derived from the state instead
    of folding it into one function: anything built from + - * /, integer powers and the unary functions below."""
    _FUNCTIONS = {"exp": "exp", "log": "log", "sin": "sin", "cos": "cos", "tanh": "tanh", "sigmoid": "sigmoid", "sqrt": "sqrt",
                  "abs": "abs", "relu": "relu", "reciprocal": "reciprocal", "neg": "neg", "square": "square"}

    def __init__(self, y, t, rows, d):
        super().__init__(y, t, rows, d)
        self.forms[id(y)] = _Expr("y")
        # t is one more leaf: the kernel hands the programs the scheme's stage time, so arithmetic on t (`torch.cos(t) * y`,
        # `y / (2 + 2 * t)`) is followed like arithmetic on the state; what is NOT elementwise (`float(t)`, `cat` with the
        # state, comparisons) fails on its own operator
        self.forms[id(t)] = _Expr("t")
        self.time = set()

    def constant_value(self, c):
        return _Expr("const", value=c)

    def state_shaped(self, tensor):
        # state-shaped, the (rows, d, 1) of scalar noise's g, or -- functions of t alone -- anything per-channel
        return tuple(tensor.shape) in ((self.rows, self.d), (self.rows, self.d, 1), (), (1,), (self.d,), (1, self.d), (1, 1))

    def operand(self, x):
        node = self.form_of(x)
        return node if node is not None else _Expr("const", value=self.coefficient(x))

    def tracked_op(self, func, args, kwargs):
        schema = func._schema
        name = schema.name.split("::")[1]
        if schema.is_mutable:
            raise NotElementwise(f"in-place {name} on a value derived from the state")
        self.check_kwargs(name, schema, kwargs)
        out = func(*args, **kwargs)
        x = self.form_of(args[0]) if args else None
        if name in self._LIKE and out.dtype == args[0].dtype:
            return self.track(out, _Expr("const", value=self._LIKE[name]))
        if name == "full_like" and out.dtype == args[0].dtype and isinstance(args[1], (int, float)):
            return self.track(out, _Expr("const", value=float(args[1])))
        if name in ("expand", "view", "reshape", "_unsafe_view", "unsqueeze", "squeeze") and x is not None \
                and torch.is_tensor(out) and out.shape != args[0].shape and out.dim() <= 2 and args[0].dim() <= 2 and out.numel() in (args[0].numel(), args[0].numel() * self.rows):
            return self.track(out, x)            # a per-channel value (a function of t) reshaped or stretched over the rows
        if name in self._SAME and x is not None:
            if not torch.is_tensor(out) or out.shape != args[0].shape or out.dtype != args[0].dtype \
                    or out.device != args[0].device:
                raise NotElementwise(f"{name} changes the shape, dtype or device of a value derived from the state")
            return self.track(out, x)
        if name == "unsqueeze" and x is not None and tuple(out.shape) == (self.rows, self.d, 1) and not x.trailing:
            return self.track(out, _Expr(x.op, x.args, x.value, trailing=True))          # scalar noise: g of shape (B, d, 1)
        if x is not None and x.trailing:
            raise NotElementwise(f"{name} after the (rows, d, 1) reshape of the diffusion")
        if name in self._FUNCTIONS and x is not None and len(args) == 1:
            return self.track(out, _Expr(self._FUNCTIONS[name], (x,)))
        if name == "softplus" and x is not None:
            beta = args[1] if len(args) > 1 else kwargs.get("beta", 1)
            threshold = args[2] if len(args) > 2 else kwargs.get("threshold", 20)
            if beta != 1 or threshold != 20:
                raise NotElementwise("softplus with a non-default beta or threshold")
            return self.track(out, _Expr("softplus", (x,)))
        # single ATen operators that are compositions of the machine's functions (to rounding: torch evaluates silu as
        # x / (1 + exp(-x)), mish with its own softplus threshold -- the both-routes check of the first solve covers that)
        if name == "silu" and x is not None and len(args) == 1:
            return self.track(out, _Expr("mul", (x, _Expr("sigmoid", (x,)))))
        if name == "mish" and x is not None and len(args) == 1:
            return self.track(out, _Expr("mul", (x, _Expr("tanh", (_Expr("softplus", (x,)),)))))
        if name == "rsqrt" and x is not None and len(args) == 1:
            return self.track(out, _Expr("reciprocal", (_Expr("sqrt", (x,)),)))
        if name in ("clamp_min", "clamp") and x is not None and self.differentiable:
            # (torch's clamp passes the gradient AT the boundary, mask x >= min; the machine's relu has slope 0 there, like
            #  torch's relu: with a state that sits exactly on 0 the sensitivity kernel would differ from autograd)
            raise NotElementwise("clamp with autograd recording: its backward passes the gradient at the boundary, relu's does not")
        if name == "clamp_min" and x is not None and len(args) == 2 and isinstance(args[1], (int, float)) and args[1] == 0:
            return self.track(out, _Expr("relu", (x,)))
        if name == "clamp" and x is not None:
            low = args[1] if len(args) > 1 else kwargs.get("min")
            high = args[2] if len(args) > 2 else kwargs.get("max")
            if isinstance(low, (int, float)) and not isinstance(low, bool) and low == 0 and high is None:
                return self.track(out, _Expr("relu", (x,)))
            raise NotElementwise("clamp other than clamp(min=0)")
        if name == "pow" and x is not None and len(args) == 2 and isinstance(args[1], (int, float)):
            n = args[1]
            if n == 1:
                return self.track(out, x)
            if n == 2:
                return self.track(out, _Expr("square", (x,)))
            if n == 3:
                return self.track(out, _Expr("cube", (x,)))           # (torch evaluates x**3 as (x * x) * x)
            if n == 0.5:
                return self.track(out, _Expr("sqrt", (x,)))
            if n == -1:
                return self.track(out, _Expr("reciprocal", (x,)))
            if n == 4:
                return self.track(out, _Expr("square", (_Expr("square", (x,)),)))
            if n == -2:
                return self.track(out, _Expr("reciprocal", (_Expr("square", (x,)),)))
            if n == -0.5:
                return self.track(out, _Expr("reciprocal", (_Expr("sqrt", (x,)),)))
            raise NotElementwise(f"the power {n} of a function of the state")
        if name in ("mul", "add", "sub", "rsub", "div") and len(args) >= 2:
            a, b = self.operand(args[0]), self.operand(args[1])
            alpha = kwargs.get("alpha", 1)
            if name == "rsub":
                a, b, name = b, a, "sub"
            if alpha != 1:
                if not isinstance(alpha, (int, float)):
                    raise NotElementwise("a tensor-valued alpha")
                b = _Expr("mul", (b, _number(alpha)))
            return self.track(out, _Expr(name, (a, b)))
        raise NotElementwise(f"operator {schema.name} on a value derived from the state")


class RecognisedProgram:
    """Drift and diffusion of a diagonal- or scalar-noise SDE as expression programs (`tsde_trajectory_prog_diag`)."""
    perceptron = neural = timed = False
    exact = False

    def __init__(self, f, g, d, dtype, device, noise_type):
        self.d, self.dtype, self.device, self.noise_type = d, dtype, device, noise_type
        if f.trailing or g.trailing != (noise_type == "scalar"):
            raise NotElementwise(f"drift / diffusion of the wrong shape for {noise_type} noise")
        self.uses_time = any(self._mentions_time(tree) for tree in (f, g))
        consts = []
        self.programs = []
        for tree in (f, g, None):
            if tree is None:
                try:
                    tree = _derivative(g)
                except NotElementwise:
                    self.programs.append(None)           # (no derivative rule: every scheme but Milstein)
                    continue
            if _Program.need(tree) > _STACK_DEPTH:
                if len(self.programs) == 2:
                    self.programs.append(None)
                    continue
                raise NotElementwise("an expression that needs more than four intermediate values at once")
            prog = _Program(consts)
            prog.compile(tree)
            self.programs.append(tuple(prog.words))
        if sum(len(w) for w in self.programs[:2]) > 96:
            raise NotElementwise("drift and diffusion of more than 96 operations")
        if self.programs[2] is not None and sum(len(w) for w in self.programs) > 96:
            self.programs[2] = None                      # (the derivative does not fit: every scheme but Milstein)
        self.consts = consts

    @staticmethod
    def _mentions_time(node):
        return node.op == "t" or any(RecognisedProgram._mentions_time(a) for a in node.args)

    def structure(self):
        """Key of the trust verdict: the programs themselves (constants by position only: values are live)."""
        return (("program", self.noise_type) + tuple(self.programs), ("consts", len(self.consts)))

    def affine_leaves(self):
        return None

    def const_table(self):
        if not self.consts:
            return torch.zeros(1, self.d, dtype=self.dtype, device=self.device)
        rows = []
        for c in self.consts:
            if torch.is_tensor(c):
                if c.dtype != self.dtype and c.dim() > 0:
                    raise NotElementwise(f"a constant of dtype {c.dtype} with a state of dtype {self.dtype}")
                rows.append(c.detach().to(device=self.device, dtype=self.dtype).reshape(-1).expand(self.d))
            else:
                rows.append(_constant_vector(float(c), self.d, self.dtype, self.device))
        return torch.stack(rows).contiguous()

    def trainable_rows(self, users_tensors=None):
        """Rows of the constant table that need a gradient: tensors with requires_grad that the user's code handed to an
        operator as they stand (parameters, or values autograd watched them derive) with one element or d elements. None
        when such a constant has another shape or there are more than four of them (the kernel carries four tangents)."""
        rows = []
        for k, c in enumerate(self.consts):
            if torch.is_tensor(c) and c.requires_grad:
                if c.numel() not in (1, self.d) or c.dtype != self.dtype or c.device != self.device:
                    return None
                rows.append(k)
        return rows if len(rows) <= _native.TRAJ_SENS - 1 else None

    def spec(self, milstein=False):
        f, g, dg = self.programs
        if milstein and dg is None:
            raise NotElementwise("Milstein: the diffusion's derivative has no program")
        return ("program_diagonal", f, g, dg if (milstein and dg) else (), self.const_table(), self.noise_type == "scalar")


def recognise_program(sde, t, y0, noise_type, rows=None, differentiable=False):
    """`recognise` for code the single-function forms cannot hold: the expression trees of `sde.f_and_g` on the probe ->
    `RecognisedProgram`, or NotElementwise (anything that is not elementwise arithmetic of the state and constants; any
    use of t)."""
    rows = 2 if rows is None else int(rows)
    if rows == y0.shape[0]:
        rows += 1
    d = y0.shape[1]
    probe = y0.detach()[:1].expand(rows, d).clone() if y0.shape[0] > 0 else torch.zeros(rows, d, dtype=y0.dtype, device=y0.device)
    t_probe = t.detach().clone()
    if differentiable:
        probe.requires_grad_(True)               # (so that a stop-gradient in the user's code shows: `check_stop_gradient`)
    interp = _TreeInterpreter(probe, t_probe, rows, d)
    interp.differentiable = bool(differentiable)
    try:
        # `differentiable`: autograd watches the user's own constant arithmetic (`-self.p ** 2`), so that a constant which is
        # such a tensor carries its graph back to the parameters (cf. `recognise`)
        with (torch.enable_grad() if differentiable else torch.no_grad()), interp:
            f, g = sde.f_and_g(t_probe, probe)
    except NotElementwise:
        raise
    except Exception as e:
        raise NotElementwise(f"{type(e).__name__}: {e}") from None
    trees = []
    for name, value in (("drift", f), ("diffusion", g)):
        tree = interp.form_of(value)
        if not isinstance(tree, _Expr):
            raise NotElementwise(f"the {name} is not a tracked function of the state")
        interp.check_stop_gradient((value,), where=f"the {name}")
        trees.append(tree)
    found = RecognisedProgram(trees[0], trees[1], d, y0.dtype, y0.device, noise_type)
    found._alive = interp.keep
    return found


class RecognisedAdditive:
    """An additive-noise SDE (base_sde.py:101-102: g depends on t only) for `tsde_trajectory_prog_additive`: the drift as an
    expression program, the diffusion as the table of its (d, m) matrix at the scheme's stage times."""
    perceptron = neural = timed = False
    exact = False
    noise_type = "additive"

    def __init__(self, f, table, m, time_dependent, d, dtype, device):
        self.d, self.m, self.dtype, self.device, self.time_dependent = d, m, dtype, device, bool(time_dependent)
        self.table = table          # (m, d), or (K, m, d): one matrix (transposed) per stage time, in the order of `times`
        self.net, self.program, self.consts = None, None, []
        if isinstance(f, _Perceptron):
            # the drift of the reference's NeuralAdditive (tests/problems.py:203-217): a perceptron of cat([t, y])
            if dtype != torch.float32 or d > 64 or not f.classic():
                raise NotElementwise("a drift network outside the neural-SDE kernel's shapes (float32, d up to 64, two layers)")
            if f.out != d or f.final is not None or f.scale != 1.0 or f.shape[1:] != (d,) or f.w1.shape[0] > 128:
                raise NotElementwise("a drift network that does not map to the state channels (or is wider than 128)")
            if any(t is not None and (t.dtype != torch.float32 or t.device != device) for t in (f.w1, f.b1, f.w2, f.b2, f.wt)):
                raise NotElementwise("network weights of another dtype or device than the state")
            self.net = f
            return
        if f.trailing:
            raise NotElementwise("a drift of the wrong shape")
        if _Program.need(f) > _STACK_DEPTH:
            raise NotElementwise("an expression that needs more than four intermediate values at once")
        self.consts = []
        prog = _Program(self.consts)
        prog.compile(f)
        if len(prog.words) > 96:
            raise NotElementwise("a drift of more than 96 operations")
        self.program = tuple(prog.words)

    def structure(self):
        if self.net is not None:
            f = self.net
            return (("perceptron", "additive", f.act, tuple(f.w1.shape), f.wt is not None, f.b1 is not None, f.b2 is not None),
                    ("consts", 0), ("g", self.m, self.time_dependent))
        return (("program", "additive", self.program), ("consts", len(self.consts)), ("g", self.m, self.time_dependent))

    def affine_leaves(self):
        return None

    const_table = RecognisedProgram.const_table

    def spec(self):
        if self.net is not None:
            from . import kernels as K
            f = self.net
            hidden = f.w1.shape[0]
            b1 = f.b1.detach() if f.b1 is not None else _constant_vector(0.0, hidden, self.dtype, self.device)
            b2 = f.b2.detach() if f.b2 is not None else _constant_vector(0.0, f.out, self.dtype, self.device)
            net = K.NeuralNet(f.w1.detach().t(), None if f.wt is None else f.wt.detach(), b1, f.w2.detach().t(), b2,
                              Recognised._ACTIVATIONS[f.act])
            return ("neural_additive", net, None, self.table, self.m)
        return ("program_additive", self.program, self.const_table(), self.table, self.m)


def recognise_additive(sde, t, y0, times, rows=None, check_rows=False):
    """Additive noise (the reference's ExAdditive, tests/problems.py:106-132): the drift's expression tree as for
    `recognise_program`; the diffusion must not touch the state (it may use its SHAPE: `.repeat(y.size(0), 1, m)`) and is
    tabulated: the user's `g` evaluated for every entry of `times` (the (K,) stage times of the solve) in ONE batched call
    (`torch.vmap` over t), or once if it does not use t. `check_rows`: also confirm that every probe row got the same
    matrix (one device synchronisation; the verifying solve asks for it)."""
    rows = 2 if rows is None else int(rows)
    if rows == y0.shape[0]:
        rows += 1
    d = y0.shape[1]
    probe = y0.detach()[:1].expand(rows, d).clone() if y0.shape[0] > 0 else torch.zeros(rows, d, dtype=y0.dtype, device=y0.device)
    t_probe = t.detach().clone()
    drift = _TreeInterpreter(probe, t_probe, rows, d)
    diffusion = _Interpreter(probe, t_probe, rows, d)
    tree = None
    try:
        with torch.no_grad():
            try:
                with drift:
                    f = sde.f(t_probe, probe)
                tree = drift.form_of(f)
                if not isinstance(tree, _Expr):
                    raise NotElementwise("the drift is not a tracked function of the state")
            except NotElementwise as first:
                # not elementwise code: a perceptron of (t, y)? (the single-function interpreter follows networks)
                drift = _Interpreter(probe, t_probe, rows, d)
                try:
                    with drift:
                        f = sde.f(t_probe, probe)
                except NotElementwise as second:
                    raise NotElementwise(f"{first}; as a network: {second}") from None
                tree = drift.net_of(f)
                if tree is None:
                    raise NotElementwise(f"{first}; and the drift is not a two-layer network either") from None
            with diffusion:
                g = sde.g(t_probe, probe)
    except NotElementwise:
        raise
    except Exception as e:
        raise NotElementwise(f"{type(e).__name__}: {e}") from None
    if not torch.is_tensor(g) or g.dim() != 3 or tuple(g.shape[:2]) != (rows, d) or g.dtype != y0.dtype or g.device != y0.device:
        raise NotElementwise(f"the diffusion of an additive-noise SDE must have shape (rows, d, m), got "
                             f"{tuple(getattr(g, 'shape', ()))}")
    if any(id(g) in book for book in (diffusion.forms, diffusion.hidden, diffusion.nets, diffusion.time_state)):
        raise NotElementwise("the diffusion of an additive-noise SDE is computed from the state")
    m = int(g.shape[2])
    if not 1 <= m <= 16:
        raise NotElementwise(f"{m} Brownian channels (the additive-noise kernel takes up to 16)")
    time_dependent = id(g) in diffusion.time
    if time_dependent:
        if times is None:
            raise NotElementwise("no stage times for this scheme")
        if times.numel() * rows * d * m > 2 ** 27:
            raise NotElementwise(f"the diffusion's table over {times.numel()} stage times would take "
                                 f"{times.numel() * d * m * y0.element_size() >> 20} MiB")
        try:
            with torch.no_grad():
                g = torch.vmap(lambda tt: sde.g(tt, probe))(times.detach().to(t_probe.dtype))
        except Exception as e:
            raise NotElementwise(f"the diffusion cannot be evaluated for all stage times at once ({type(e).__name__}: {e})") \
                from None
        if tuple(g.shape) != (times.numel(), rows, d, m) or g.dtype != y0.dtype:
            raise NotElementwise(f"the diffusion evaluated for all stage times has shape {tuple(g.shape)}")
        # (to rounding: a network of t goes through a batched matrix product whose rows need not agree in the last bit)
        tight = dict(rtol=1e-5, atol=1e-7) if g.dtype == torch.float32 else dict(rtol=1e-12, atol=1e-14)
        if check_rows and not torch.allclose(g[:, 0], g[:, -1], **tight):
            raise NotElementwise("the diffusion differs between batch rows")
        table = g[:, 0].transpose(1, 2).contiguous()                 # (K, m, d)
    else:
        if check_rows and not torch.equal(g[0], g[-1]):
            raise NotElementwise("the diffusion differs between batch rows")
        table = g[0].t().contiguous()                                # (m, d)
    found = RecognisedAdditive(tree, table, m, time_dependent, d, y0.dtype, y0.device)
    found._alive = drift.keep + diffusion.keep
    return found


class Recognised:
    """What the interpretation found: the function codes and, per function, the four coefficients (None = neutral)."""

    def __init__(self, f, g, d, dtype, device):
        self.f, self.g, self.d, self.dtype, self.device = f, g, d, dtype, device
        self.exact = f.exact and g.exact
        if isinstance(g, _Perceptron) and not isinstance(f, _Perceptron):
            raise NotElementwise("a network diffusion beside a drift that is not a network")

    @property
    def perceptron(self):
        """Perceptron drift with an elementwise diagonal diffusion: `tsde_trajectory_mlp_diag` and its adjoint."""
        return isinstance(self.f, _Perceptron) and not isinstance(self.g, _Perceptron)

    @property
    def neural(self):
        """Drift AND diffusion are perceptrons (of y, or of [t, y]): `tsde_trajectory_mlp_general`."""
        return isinstance(self.f, _Perceptron) and isinstance(self.g, _Perceptron)

    def structure(self):
        """The part of the result that does not change when parameter VALUES change: key of the trust verdict."""
        def shape(form):
            if isinstance(form, _Perceptron):
                return ("perceptron", form.act, tuple(form.w1.shape), form.b1 is None, form.b2 is None, form.out,
                        form.wt is not None, form.final, form.scale, form.shape[1:], len(form.mids), form.act_scale,
                        tuple(b is None for _, b in form.mids))
            return (form.phi, form.constant()) + tuple(
                None if c is None else "number" if isinstance(c, (int, float))
                else "table" if torch.is_tensor(c) and c.dim() == 3 else "tensor"
                for c in (form.scale, form.rate, form.shift, form.offset))
        return shape(self.f), shape(self.g)

    def _vector(self, c, neutral):
        """One coefficient as a contiguous (d,) tensor of the state dtype."""
        if c is None or isinstance(c, (bool, int, float)):
            return _constant_vector(neutral if c is None else float(c), self.d, self.dtype, self.device)
        if c.dtype != self.dtype and c.dim() > 0:
            raise NotElementwise(f"a coefficient of dtype {c.dtype} with a state of dtype {self.dtype}")
        if c.dtype != self.dtype and not c.is_floating_point():
            raise NotElementwise(f"a 0-d coefficient of dtype {c.dtype}")
        if c.dtype != self.dtype or c.device != self.device:      # a 0-d tensor takes part like a Python number
            c = c.to(device=self.device, dtype=self.dtype)
        if c.dim() == 3:            # (K, 1, 1) or (K, 1, d): a function of t, one row per step
            return c.detach().reshape(c.shape[0], -1).expand(c.shape[0], self.d).contiguous()
        return c.detach().reshape(-1).expand(self.d).contiguous()

    @staticmethod
    def _as_tables(vectors):
        """If any coefficient is a (K, d) table, every one becomes one (the `_timed` kernels take all or none)."""
        steps = [v.shape[0] for v in vectors if v.dim() == 2]
        if not steps:
            return vectors
        return tuple(v if v.dim() == 2 else v.expand(steps[0], v.shape[0]).contiguous() for v in vectors)

    @property
    def timed(self):
        if self.neural:
            return False
        forms = [self.g] if self.perceptron else [self.f, self.g]
        return any(torch.is_tensor(c) and c.dim() == 3 for v in forms for c in (v.scale, v.rate, v.shift, v.offset))

    def _four(self, form):
        if form.phi == "poly3":     # (c3, c2, c1, c0): every missing coefficient is 0
            return tuple(self._vector(c, 0.0) for c in (form.scale, form.rate, form.shift, form.offset))
        if form.constant():         # the value is `shift (+ offset)`: rate 0, identity
            return (self._vector(None, 1.0), self._vector(None, 0.0), self._vector(_add(form.shift, form.offset), 0.0),
                    self._vector(None, 0.0))
        return (self._vector(form.scale, 1.0), self._vector(form.rate, 1.0), self._vector(form.shift, 0.0),
                self._vector(form.offset, 0.0))

    _ACTIVATIONS = {"tanh": 0, "softplus": 1}              # closed_form.MLPDriftDiagonalSDE._ACTIVATIONS

    def perceptron_diffusion(self):
        """(kind code, amplitude, rate, shift) of a diffusion the perceptron kernels take: `rate * y + shift` (affine) or
        `amplitude * sigmoid(rate * y + shift)` with a fixed number as amplitude; else NotElementwise."""
        g = self.g
        if isinstance(g, _Perceptron):
            raise NotElementwise("a perceptron diffusion")
        if self.timed:
            raise NotElementwise("a time-dependent diffusion beside a perceptron drift")
        if g.constant():
            return _native.DIFF_AFFINE, 1.0, 0.0, _add(g.shift, g.offset)
        if g.phi == "identity" and g.scale is None and g.offset is None:
            return _native.DIFF_AFFINE, 1.0, g.rate, g.shift
        if g.phi == "sigmoid" and g.offset is None and (g.scale is None or isinstance(g.scale, (int, float))):
            return _native.DIFF_SIGMOID, 1.0 if g.scale is None else float(g.scale), g.rate, g.shift
        raise NotElementwise(f"a {g.phi} diffusion beside a perceptron drift")

    def perceptron_supported(self, float32_only=True):
        """The shape limits of the perceptron kernels (closed_form.MLPDriftDiagonalSDE.closed_form)."""
        f = self.f
        hidden = f.w1.shape[0]
        tensors = [t for t in (f.w1, f.b1, f.w2, f.b2) if t is not None]
        if not f.plain() or f.out != self.d:
            return False            # (a net that takes t, or is followed by a factor: not what these kernels evaluate)
        return (self.dtype == torch.float32 and all(t.dtype == torch.float32 and t.device == self.device for t in tensors)
                and self.d % 4 == 0 and self.d <= 128 and hidden <= (256 if self.d <= 64 else 128))

    def perceptron_spec(self):
        """("mlp_diagonal", W1 (d, hidden) input-major, b1, W2 (hidden, d), b2, rate (d,), shift (d,), activation code,
        (diffusion kind, amplitude)): what `kernels.trajectory_mlp_diag` takes (closed_form.py)."""
        f = self.f
        if not f.plain() or f.out != self.d:
            raise NotElementwise("a drift network that takes t (or is followed by a factor) beside an elementwise diffusion: "
                                 "the perceptron-drift kernels have no time input, the neural-SDE kernel wants a diffusion "
                                 "network")
        if not self.perceptron_supported():
            raise NotElementwise("a perceptron drift outside the kernels' shapes")
        kind, amplitude, rate, shift = self.perceptron_diffusion()
        hidden = f.w1.shape[0]
        b1 = f.b1.detach().contiguous() if f.b1 is not None else _constant_vector(0.0, hidden, self.dtype, self.device)
        b2 = f.b2.detach().contiguous() if f.b2 is not None else _constant_vector(0.0, self.d, self.dtype, self.device)
        return ("mlp_diagonal", f.w1.detach().t().contiguous(), b1, f.w2.detach().t().contiguous(), b2,
                self._vector(rate, 1.0), self._vector(shift, 0.0), self._ACTIVATIONS[f.act], (kind, amplitude))

    _NOISE_CODES = {"diagonal": _native.NOISE_DIAGONAL, "scalar": _native.NOISE_SCALAR, "general": _native.NOISE_GENERAL}

    def neural_spec(self, noise_type):
        """("neural", drift NeuralNet, diffusion NeuralNet, noise code, m): what `kernels.trajectory_mlp_general` takes, for
        a module whose drift and diffusion are both perceptrons (the reference's Neural* problems, tests/problems.py:135-252).
        The diffusion's output must have the shape the noise type prescribes (sdeint.py:179-197): (rows, d) diagonal,
        (rows, d, 1) scalar, (rows, d, m) general."""
        from . import kernels as K
        f, g, d = self.f, self.g, self.d
        if self.dtype != torch.float32:
            raise NotElementwise("the neural-SDE kernel is float32")
        if not (f.classic() and g.classic()):
            raise NotElementwise("networks deeper than two layers, or with functions other than tanh / softplus (+ closing "
                                 "sigmoid): the reversible-Heun kernels take those, the Euler / midpoint / SRK kernel does not")
        if f.out != d or f.final is not None or f.shape[1:] != (d,):
            raise NotElementwise("a drift network that does not map to the state channels")
        if noise_type == "diagonal":
            want, m = (d,), d
        elif noise_type == "scalar":
            want, m = (d, 1), 1
        elif noise_type == "general":
            if len(g.shape) != 3 or g.shape[1] != d:
                raise NotElementwise(f"a general-noise diffusion of shape {g.shape}")
            want, m = (d, g.shape[2]), g.shape[2]
        else:
            raise NotElementwise(f"{noise_type} noise")
        if g.shape[1:] != want or g.out != (d if noise_type != "general" else d * m):
            raise NotElementwise(f"a diffusion network of shape {g.shape} for {noise_type} noise")
        nets = []
        for net in (f, g):
            tensors = [t for t in (net.w1, net.b1, net.w2, net.b2, net.wt) if t is not None]
            if any(t.dtype != torch.float32 or t.device != self.device for t in tensors):
                raise NotElementwise("network weights of another dtype or device than the state")
            hidden = net.w1.shape[0]
            scale = float(net.scale)
            w2, b2 = net.w2.detach().t(), (net.b2.detach() if net.b2 is not None else
                                           _constant_vector(0.0, net.out, self.dtype, self.device))
            fold = net.final is None and scale != 1.0          # a plain factor after a linear layer: into that layer
            if fold:
                w2, b2, scale = w2 * scale, b2 * scale, 1.0
            b1 = net.b1.detach() if net.b1 is not None else _constant_vector(0.0, hidden, self.dtype, self.device)
            nets.append(K.NeuralNet(net.w1.detach().t(), None if net.wt is None else net.wt.detach(), b1, w2, b2,
                                    self._ACTIVATIONS[net.act],
                                    _native.FINAL_SIGMOID if net.final == "sigmoid" else _native.FINAL_NONE, scale))
        noise = self._NOISE_CODES[noise_type]
        need = K.mlp_general_lds(d, m, nets[0].hidden, nets[1].hidden, nets[1].out, noise)
        if need <= 0 or need > 160 * 1024:
            raise NotElementwise(f"networks outside the neural-SDE kernel's shapes (d = {d}, m = {m}, hidden "
                                 f"{nets[0].hidden} / {nets[1].hidden}: {need} bytes of LDS)")
        return ("neural", nets[0], nets[1], noise, m)

    _DEEP_ACTS = {"tanh": _native.ACT_TANH, "softplus": _native.ACT_SOFTPLUS, "silu": _native.ACT_SILU}
    _DEEP_FINALS = {None: _native.FINAL_NONE, "sigmoid": _native.FINAL_SIGMOID, "tanh": _native.FINAL_TANH}

    def deep_spec(self, noise_type):
        """("neural_rheun", drift DeepNet, diffusion DeepNet, noise code, m): what the reversible-Heun kernels take
        (neural_rheun.py) -- perceptrons of two to four Linear layers, tanh / softplus / (a factor times) silu between them, an
        optional closing sigmoid or tanh and a numeric factor: the reference's Neural* problems AND the generator of
        examples/sde_gan.py:50-66,77-101. The DeepNets hold the USER's own tensors (`nn.Linear.weight` as it stands, time column
        included), so that the backward kernels' gradients land on them."""
        from . import neural_rheun
        f, g, d = self.f, self.g, self.d
        if not self.neural:
            raise NotElementwise("drift and diffusion are not both networks")
        if self.dtype != torch.float32:
            raise NotElementwise("the reversible-Heun kernels are float32")
        if f.out != d or f.shape[1:] != (d,):
            raise NotElementwise("a drift network that does not map to the state channels")
        if noise_type == "diagonal":
            want, m = (d,), d
        elif noise_type == "scalar":
            want, m = (d, 1), 1
        elif noise_type == "general":
            if len(g.shape) != 3 or g.shape[1] != d:
                raise NotElementwise(f"a general-noise diffusion of shape {g.shape}")
            want, m = (d, g.shape[2]), g.shape[2]
        else:
            raise NotElementwise(f"{noise_type} noise")
        if g.shape[1:] != want or g.out != (d if noise_type != "general" else d * m):
            raise NotElementwise(f"a diffusion network of shape {g.shape} for {noise_type} noise")
        nets = []
        for net in (f, g):
            if net.final not in self._DEEP_FINALS:
                raise NotElementwise(f"a network closed by {net.final}")
            if len(net.mids) > neural_rheun.MAX_MID:
                raise NotElementwise(f"a network of more than {neural_rheun.MAX_MID + 2} Linear layers")
            linears = [(net.w_full, net.b1)] + list(net.mids) + [(net.w2, net.b2)]
            if any(t is not None and (t.dtype != torch.float32 or t.device != self.device) for pair in linears for t in pair):
                raise NotElementwise("network weights of another dtype or device than the state")
            if any(tuple(w.shape) != (net.w1.shape[0], net.w1.shape[0]) for w, _ in net.mids):
                raise NotElementwise("hidden layers of different widths")
            nets.append(neural_rheun.DeepNet(linears, self._DEEP_ACTS[net.act], net.act_scale, self._DEEP_FINALS[net.final],
                                             float(net.scale), time_input=net.wt is not None))
        noise = self._NOISE_CODES[noise_type]
        need = neural_rheun.lds_bytes(d, m, nets[0], nets[1], noise)
        if need <= 0 or need > 160 * 1024:
            raise NotElementwise(f"networks outside the reversible-Heun kernels' shapes (d = {d}, m = {m}, hidden "
                                 f"{nets[0].hidden} / {nets[1].hidden}, {nets[0].n_mid + 2} / {nets[1].n_mid + 2} layers: "
                                 f"{need} bytes of LDS)")
        return ("neural_rheun", nets[0], nets[1], noise, m)

    def perceptron_parameters(self):
        """(lin1.weight, lin1.bias, lin2.weight, lin2.bias, rate, shift) as the tensors the user's module holds -- what
        `sdeint_adjoint` through `tsde_adjoint_mlp_diag` returns gradients for -- or None when a coefficient of the
        diffusion is not such a tensor as it stands (derived from parameters by arithmetic: the trace ran without
        autograd, so its gradient could not be passed on) or a layer has no bias."""
        f = self.f
        _, _, rate, shift = self.perceptron_diffusion()
        if f.b1 is None or f.b2 is None:
            return None
        own = [f.w1, f.b1, f.w2, f.b2]
        for c, neutral in ((rate, 1.0), (shift, 0.0)):
            if c is None or isinstance(c, (bool, int, float)):
                own.append(torch.tensor(float(neutral if c is None else c), dtype=self.dtype, device=self.device))
            elif c.dim() == 1 and c.numel() in (1, self.d) or c.dim() == 0:
                own.append(c)
            else:
                return None
        return own

    def affine_leaves(self):
        """For a solve that autograd records: (drift rate, drift shift, diffusion rate, diffusion shift) as tensors the
        sensitivity kernel's gradients can be handed to (`kernels.trajectory_affine_diag_differentiable`), or None.
        Only plain `rate * y + shift` forms qualify, and only when every tensor coefficient is the user's OWN tensor --
        seen as an operand of their code, so a parameter or something autograd saw them compute -- of shape (d,) or one
        element: a coefficient folded inside the interpretation (`mu - 0.5 * sigma ** 2` assembled from two terms) has
        no graph behind it."""
        if self.perceptron or self.timed or not all(
                v.phi == "identity" and v.scale is None and v.offset is None and not v.constant() for v in (self.f, self.g)):
            return None
        out = []
        for c, neutral in ((self.f.rate, 1.0), (self.f.shift, 0.0), (self.g.rate, 1.0), (self.g.shift, 0.0)):
            if c is None or isinstance(c, (bool, int, float)):
                out.append(torch.tensor(float(neutral if c is None else c), dtype=self.dtype, device=self.device))
            elif (id(c) in self.users_tensors and c.dtype == self.dtype and c.device == self.device
                  and (c.numel() == 1 or tuple(c.shape) == (self.d,))):
                out.append(c)
            else:
                return None
        return out

    def spec(self):
        """The `closed_form()` tuple the trajectory launchers take (closed_form.py): affine kernel when both functions
        are plain `rate * y + shift`, else the expression kernel; perceptron drift: `perceptron_spec`."""
        if self.perceptron:
            return self.perceptron_spec()
        if self.neural:
            raise NotElementwise("drift and diffusion networks: `neural_spec(noise_type)`")
        f4, g4 = self._four(self.f), self._four(self.g)
        plain = all(v.phi == "identity" and v.scale is None and v.offset is None for v in (self.f, self.g))
        if plain:
            return ("affine_diagonal",) + self._as_tables((f4[1], f4[2], g4[1], g4[2]))
        return ("elementwise_diagonal", _native.FN_CODES[self.f.phi], _native.FN_CODES[self.g.phi]) + self._as_tables(f4 + g4)


_VECTORS = {}


def _constant_vector(value, d, dtype, device):
    key = (value, d, dtype, str(device))
    hit = _VECTORS.get(key)
    if hit is None:
        if len(_VECTORS) > 256:
            _VECTORS.clear()
        hit = _VECTORS[key] = torch.full((d,), value, dtype=dtype, device=device)
    return hit


def recognise(sde, t, y0, differentiable=False, times=None, rows=None):
    """Interpret ``sde.f_and_g`` (a ForwardSDE: whichever of f / g / f_and_g the user defined) on a probe of the state's
    width; returns `Recognised` or raises `NotElementwise`. Launches a handful of tiny kernels, never synchronises.
    `rows`: the probe's height (default 2). Code that derives a coefficient from the batch size (`y / y.shape[0]`) is
    seen by interpreting at two heights and comparing the coefficients (the trust check of `_integrate_recognised`)."""
    rows = 2 if rows is None else int(rows)
    if rows == y0.shape[0]:
        rows += 1                               # a per-ROW constant of the real batch cannot broadcast against the probe
    d = y0.shape[1]
    probe = y0.detach()[:1].expand(rows, d).clone() if y0.shape[0] > 0 else torch.zeros(rows, d, dtype=y0.dtype,
                                                                                       device=y0.device)
    # `times`: the (K,) tensor of every step's start time -> t is handed to the user's code as a (K, 1, 1) tensor and
    # coefficients that depend on t come back as one row per step (module docstring)
    t_probe = t.detach().clone() if times is None else times.detach().reshape(-1, 1, 1).clone()
    if differentiable:
        probe.requires_grad_(True)               # (so that a stop-gradient in the user's code shows: `check_stop_gradient`)
    interp = _Interpreter(probe, t_probe, rows, d, steps=None if times is None else int(times.numel()))
    interp.differentiable = bool(differentiable)
    try:
        # `differentiable`: autograd watches what the user's code computes from its parameters on the way (`-self.theta`,
        # `self.sigma ** 2`), so that a coefficient which is such a tensor carries its graph (see `affine_leaves`)
        with (torch.enable_grad() if differentiable else torch.no_grad()), interp:
            f, g = sde.f_and_g(t_probe, probe)
    except NotElementwise:
        raise
    except Exception as e:        # the user's code failed on the probe (or on the vector of step times) (a per-row buffer of the real batch, say)
        raise NotElementwise(f"{type(e).__name__}: {e}") from None
    forms = []
    for name, value in (("drift", f), ("diffusion", g)):
        form = interp.form_of(value)
        if form is None:
            form = interp.net_of(value)
        if form is None:
            raise NotElementwise(f"the {name} is not a tracked function of the state")
        interp.check_stop_gradient((value,), where=f"the {name}")
        forms.append(form)
    found = Recognised(forms[0], forms[1], d, y0.dtype, y0.device)
    found.users_tensors = interp.seen
    found._alive = interp.keep        # (the ids above stay meaningful for as long as this object lives)
    return found


def describe(sde):
    """What the recognised route has decided about `sde` so far, one line per form / refusal: the answer to "why is my
    solve (not) one kernel launch?" (cf. `graph.describe_cache` for the stepwise route's launch graphs)."""
    base = sde
    while hasattr(base, "_base_sde"):
        base = base._base_sde
    book = getattr(base, "_tsde_recognised", None)
    if not book:
        return ["nothing recorded: no solve of this object has reached the recognised route (see the conditions in "
                "solvers.BaseSDESolver._integrate_recognised: diagonal noise, fixed step, this package's BrownianInterval, "
                "a CUDA state of at least 8 rows)"]
    lines = []
    for key, verdict in book["trusted"].items():
        structure, _, solver, sde_type, d, dtype, batch = key[:7]
        kind = ("perceptron drift" if structure[0][0] == "perceptron"
                else f"expression program, {structure[0][1]} noise" if structure[0][0] == "program"
                else f"f: {structure[0][0]}, g: {structure[1][0]}")
        timed = any("table" in part for part in structure if isinstance(part, tuple))
        route = ("trajectory kernel" + (" with per-stage-time coefficient rows" if timed else "")
                 + (" (sensitivity kernel: autograd)" if key[7:] == ("autograd",) else ""))
        lines.append(f"[{solver}, {sde_type}, batch = {batch}, d = {d}, {dtype}] {kind}: "
                     + (route if verdict is True else f"stays stepwise: {verdict}"))
    for (_, _, solver), reason in book["refused"].items():
        lines.append(f"[{solver}] stays stepwise: {reason}")
    return lines

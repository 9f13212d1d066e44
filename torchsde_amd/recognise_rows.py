"""Small ROW-COUPLED systems -- channels that read each other -- as one trajectory-kernel launch.

The hand-written SDEs of the reference's examples are not all elementwise: `StochasticLorenz`
(examples/latent_sde_lorenz.py:56-86) splits the state into its columns, does arithmetic among them and concatenates the result

    x1, x2, x3 = torch.split(y, (1, 1, 1), dim=1)
    f1 = a1 * (x2 - x1);  f2 = a2 * x1 - x2 - x1 * x3;  f3 = x1 * x2 - a3 * x3
    return torch.cat([f1, f2, f3], dim=1)

-- and so do Van der Pol, FitzHugh-Nagumo, SIR ... The elementwise interpreters of recognise.py end at the `split`. This one
follows COLUMNS: every value derived from the state is a list of per-column expression trees over the d state channels, t and
scalar constants; `split` / `chunk` / `unbind` / `y[:, c]` / `y[:, a:b]` pick columns, + - * /, integer powers and the unary
functions act column by column, `cat` / `stack` along dim 1 assemble the result. What comes out is d scalar expressions for the
drift and d for the diffusion (diagonal noise), which torchsde_amd/specialise.py turns into a model for the program kernel with
W = d: ONE LANE OWNS A WHOLE ROW of the batch (d <= 8 state values in registers) for the whole solve. There is no interpreter
for such systems: the route exists once the generated unit is compiled (4 s, in the background; until then the solve is
stepwise), and is trusted like every recognised route only after its first solve reproduced the stepwise one.

Schemes: those of the program kernel that need no derivative of g -- Euler, midpoint, Heun, Euler-Heun, SRK.
"""
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from .recognise import NotElementwise, _Expr

MAX_D = 8
_UNARY = {"exp", "log", "sin", "cos", "tanh", "sigmoid", "sqrt", "abs", "relu", "reciprocal", "neg"}
_VIEWS = {"alias", "detach", "clone", "contiguous", "lift_fresh", "positive", "_to_copy"}


def _const(value):
    return _Expr("const", value=value)


class _Columns(TorchDispatchMode):
    """`cols[id(tensor)]`: the column expressions of a (rows, k) or (rows,) value derived from the state."""

    def __init__(self, y, t, rows, d):
        super().__init__()
        self.rows, self.d = rows, d
        self.cols = {id(y): [_Expr("y", value=c) for c in range(d)]}
        self.keep = [y, t]
        self.t_id = id(t)
        self.flat = set()           # ids of tracked values of shape (rows,) (a column without its unit axis)
        self.scalar_like = set()    # ids of 0-d functions of t (they broadcast against either kind)
        self.made = {id(y), id(t)}  # ids of tensors whose storage was allocated during the interpretation
        self.differentiable = False

    def track(self, tensor, cols, flat=False):
        want = (self.rows,) if flat else (self.rows, len(cols))
        if tuple(tensor.shape) != want:
            raise NotElementwise(f"a value derived from the state has shape {tuple(tensor.shape)}, not {want}")
        self.cols[id(tensor)] = list(cols)
        if flat:
            self.flat.add(id(tensor))
        self.keep.append(tensor)
        return tensor

    def columns_of(self, x, k):
        """The k column expressions of operand x: a tracked value (k columns, or one that broadcasts), t, or a constant."""
        if torch.is_tensor(x) and id(x) in self.cols:
            c = self.cols[id(x)]
            if len(c) == k:
                return c
            if len(c) == 1:
                return c * k
            raise NotElementwise("operands with different numbers of columns")
        if torch.is_tensor(x) and id(x) == self.t_id:
            return [_Expr("t")] * k
        if isinstance(x, (bool, int, float)):
            return [_const(float(x))] * k
        if torch.is_tensor(x):
            if x.numel() == 1:
                return [_const(x.reshape(()))] * k
            if x.dim() <= 2 and x.shape[-1] == k and x.numel() == k:
                flat = x.reshape(-1)
                return [_const(flat[j]) for j in range(k)]
            raise NotElementwise(f"an operand of shape {tuple(x.shape)} beside columns of the state")
        raise NotElementwise(f"an operand of type {type(x).__name__}")

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        self.check_side_effects(func, args, kwargs)
        out = self.follow(func, args, kwargs)
        returns = func._schema.returns
        outs = out if isinstance(out, (list, tuple)) else (out,)
        for k, o in enumerate(outs):
            if torch.is_tensor(o) and id(o) not in self.made:
                aliasing = k < len(returns) and returns[k].alias_info is not None
                if not aliasing or (args and torch.is_tensor(args[0]) and id(args[0]) in self.made):
                    self.made.add(id(o))
                    self.keep.append(o)
        return out

    def check_side_effects(self, func, args, kwargs):
        """As recognise._Interpreter.check_side_effects: the code runs once per solve here, once per step in the reference
        (base_solver.py:114-149) -- random draws and in-place writes to tensors that existed before the call end the
        interpretation."""
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

    def follow(self, func, args, kwargs):
        schema = func._schema
        name = schema.name.split("::")[1]
        flat_args = list(args) + list(kwargs.values())
        tensors = [a for a in flat_args if torch.is_tensor(a)]
        for a in flat_args:
            if isinstance(a, (list, tuple)):
                tensors.extend(x for x in a if torch.is_tensor(x))
        for a in tensors:
            self.keep.append(a)
        tracked = [a for a in tensors if id(a) in self.cols]
        if self.differentiable:
            for a in tracked:
                if a.is_floating_point() and not a.requires_grad and any(c.mentions_state() for c in self.cols[id(a)]):
                    raise NotElementwise("a stop-gradient on a value derived from the state")
        uses_t = any(id(a) == self.t_id for a in tensors)
        if not tracked and not uses_t:
            return func(*args, **kwargs)
        if uses_t and not tracked:
            if name in ("_local_scalar_dense", "item"):
                raise NotElementwise("the code reads t on the host")
            # arithmetic among t and constants: a (0-d) function of t, followed as a one-column value
            out = func(*args, **kwargs)
            if not torch.is_tensor(out) or out.numel() != 1:
                raise NotElementwise(f"{name} makes something other than a number out of t")
            expr = self.elementwise(name, args, kwargs, 1, out)
            self.cols[id(out)] = expr
            self.keep.append(out)
            self.scalar_like.add(id(out))
            return out
        if schema.is_mutable:
            raise NotElementwise(f"in-place {name} on a value derived from the state")
        out = func(*args, **kwargs)
        src = args[0] if args and torch.is_tensor(args[0]) and id(args[0]) in self.cols else None
        if name in _VIEWS and src is not None and torch.is_tensor(out) and out.shape == src.shape and out.dtype == src.dtype:
            return self.track(out, self.cols[id(src)], flat=id(src) in self.flat)
        if name in ("split_with_sizes", "split", "chunk", "unbind", "tensor_split") and src is not None:
            dim = args[2] if len(args) > 2 else kwargs.get("dim", 0)
            if name == "unbind":
                dim = args[1] if len(args) > 1 else kwargs.get("dim", 0)
            if id(src) in self.flat or dim not in (1, -1):
                raise NotElementwise(f"{name} of the state along the batch")
            cols, at, pieces = self.cols[id(src)], 0, []
            for piece in out:
                if name == "unbind":
                    pieces.append(self.track(piece, cols[at:at + 1], flat=True))
                    at += 1
                else:
                    width = piece.shape[1]
                    pieces.append(self.track(piece, cols[at:at + width]))
                    at += width
            return type(out)(pieces) if isinstance(out, tuple) else pieces
        if name == "select" and src is not None and id(src) not in self.flat:
            dim, index = args[1], args[2]
            if dim not in (1, -1):
                raise NotElementwise("a row of the batch is picked out")
            return self.track(out, [self.cols[id(src)][index]], flat=True)
        if name in ("slice", "narrow") and src is not None:
            dim = args[1] if len(args) > 1 else kwargs.get("dim", 0)
            if dim in (1, -1) and id(src) not in self.flat:
                cols = self.cols[id(src)]
                if name == "slice":
                    start = args[2] if len(args) > 2 and args[2] is not None else 0
                    end = args[3] if len(args) > 3 and args[3] is not None else len(cols)
                    step = args[4] if len(args) > 4 else 1
                    picked = cols[slice(start, min(end, len(cols)), step)]
                else:
                    picked = cols[args[2]:args[2] + args[3]]
                return self.track(out, picked)
            if tuple(out.shape) == tuple(src.shape):           # `y[:]`: the whole batch
                return self.track(out, self.cols[id(src)], flat=id(src) in self.flat)
            raise NotElementwise("rows of the batch are picked out")
        if name in ("unsqueeze", "squeeze", "view", "reshape", "_unsafe_view", "_reshape_alias") and src is not None:
            cols = self.cols[id(src)]
            if tuple(out.shape) == (self.rows, len(cols)):
                return self.track(out, cols)
            if len(cols) == 1 and tuple(out.shape) == (self.rows,):
                return self.track(out, cols, flat=True)
            raise NotElementwise(f"{name} gives a value derived from the state shape {tuple(out.shape)}")
        if name in ("cat", "stack"):
            pieces = args[0]
            dim = args[1] if len(args) > 1 else kwargs.get("dim", 0)
            if dim not in (1, -1):
                raise NotElementwise(f"{name} along the batch")
            cols = []
            for piece in pieces:
                if id(piece) not in self.cols or id(piece) in self.scalar_like:
                    raise NotElementwise(f"{name} with a block that is not made of columns of the state")
                if (name == "stack") != (id(piece) in self.flat):
                    raise NotElementwise(f"{name} of columns with / without their unit axis")
                cols.extend(self.cols[id(piece)])
            return self.track(out, cols)
        if name in ("zeros_like", "ones_like") and src is not None:
            value = 0.0 if name == "zeros_like" else 1.0
            return self.track(out, [_const(value)] * len(self.cols[id(src)]), flat=id(src) in self.flat)
        # elementwise arithmetic, column by column
        shaped = [a for a in tracked if id(a) not in self.scalar_like]
        if not shaped:               # functions of t among themselves
            expr = self.elementwise(name, args, kwargs, 1, out)
            self.cols[id(out)] = expr
            self.keep.append(out)
            self.scalar_like.add(id(out))
            return out
        k = max(len(self.cols[id(a)]) for a in shaped)
        flat = all(id(a) in self.flat for a in shaped)
        if any((id(a) in self.flat) != flat for a in shaped):
            raise NotElementwise("columns with and without their unit axis in one operation")
        return self.track(out, self.elementwise(name, args, kwargs, k, out), flat=flat)

    def elementwise(self, name, args, kwargs, k, out):
        if name in _UNARY and len(args) == 1:
            return [_Expr(name, (c,)) for c in self.columns_of(args[0], k)]
        if name == "softplus":
            beta = args[1] if len(args) > 1 else kwargs.get("beta", 1)
            threshold = args[2] if len(args) > 2 else kwargs.get("threshold", 20)
            if beta != 1 or threshold != 20:
                raise NotElementwise("softplus with a non-default beta or threshold")
            return [_Expr("softplus", (c,)) for c in self.columns_of(args[0], k)]
        if name == "pow" and len(args) == 2 and isinstance(args[1], (int, float)):
            n = args[1]
            table = {1: None, 2: "square", 3: "cube", 0.5: "sqrt", -1: "reciprocal"}
            if n not in table:
                raise NotElementwise(f"the power {n}")
            cols = self.columns_of(args[0], k)
            return cols if table[n] is None else [_Expr(table[n], (c,)) for c in cols]
        if name in ("mul", "add", "sub", "rsub", "div") and len(args) >= 2:
            for key, value in kwargs.items():
                if key != "alpha" and value is not None:
                    raise NotElementwise(f"{name} with {key}={value!r}")
            a, b = self.columns_of(args[0], k), self.columns_of(args[1], k)
            alpha = kwargs.get("alpha", 1)
            if name == "rsub":
                a, b, name = b, a, "sub"
            if alpha != 1:
                if not isinstance(alpha, (int, float)):
                    raise NotElementwise("a tensor-valued alpha")
                b = [_Expr("mul", (c, _const(float(alpha)))) for c in b]
            return [_Expr(name, (x, y)) for x, y in zip(a, b)]
        raise NotElementwise(f"operator {name} on columns of the state")


class RecognisedRows:
    """d drift and d diffusion expressions over the row's d channels, t and scalar constants."""
    perceptron = neural = timed = False
    exact = False

    def __init__(self, f, g, d, dtype, device):
        self.f, self.g, self.d, self.dtype, self.device = list(f), list(g), d, dtype, device
        self.consts = []            # scalar constants in order of first use: numbers, or 0-d tensors (live values)
        self.statements, self.outputs = self._linearise()

    def _linearise(self):
        """Every distinct node once, in evaluation order: [(name, op, operand names)], and the 2 d output names."""
        names, statements = {}, []

        def visit(node):
            if id(node) in names:
                return names[id(node)]
            if node.op == "y":
                name = f"x.v[{node.value}]"
            elif node.op == "t":
                name = "time"
            elif node.op == "const":
                if torch.is_tensor(node.value):
                    for k, have in enumerate(self.consts):
                        if have is node.value:
                            break
                    else:
                        self.consts.append(node.value)
                        k = len(self.consts) - 1
                    name = f"c[{k}]"
                else:
                    name = f"(T){float(node.value)!r}"
            else:
                operands = [visit(a) for a in node.args]
                name = f"n{len(statements)}"
                statements.append((name, node.op, operands))
            names[id(node)] = name
            return name
        outputs = [visit(n) for n in self.f] + [visit(n) for n in self.g]
        if len(statements) > 512:
            raise NotElementwise("a system of more than 512 operations")
        return statements, outputs

    def structure(self):
        return (("rows", self.d, tuple((op, tuple(operands)) for _, op, operands in self.statements), tuple(self.outputs)),
                ("consts", len(self.consts)))

    def affine_leaves(self):
        return None

    def const_table(self):
        if not self.consts:
            return torch.zeros(1, dtype=self.dtype, device=self.device)
        return torch.stack([c.detach().to(device=self.device, dtype=self.dtype).reshape(()) for c in self.consts]).contiguous()

    def spec(self):
        return ("program_rows", self.structure(), self.const_table(), self.d)


def recognise_rows(sde, t, y0, rows=None, differentiable=False):
    """`RecognisedRows` for a diagonal-noise SDE whose f and g do arithmetic among the columns of the state, or NotElementwise."""
    rows = 2 if rows is None else int(rows)
    if rows == y0.shape[0]:
        rows += 1
    d = y0.shape[1]
    if d > MAX_D:
        raise NotElementwise(f"a row-coupled system of more than {MAX_D} channels")
    probe = y0.detach()[:1].expand(rows, d).clone() if y0.shape[0] > 0 else torch.zeros(rows, d, dtype=y0.dtype, device=y0.device)
    t_probe = t.detach().clone()
    interp = _Columns(probe, t_probe, rows, d)
    try:
        with torch.no_grad(), interp:
            f, g = sde.f_and_g(t_probe, probe)
    except NotElementwise:
        raise
    except Exception as e:
        raise NotElementwise(f"{type(e).__name__}: {e}") from None
    out = []
    for name, value in (("drift", f), ("diffusion", g)):
        cols = interp.cols.get(id(value)) if torch.is_tensor(value) else None
        if cols is None or id(value) in interp.flat or len(cols) != d or tuple(value.shape) != (rows, d):
            raise NotElementwise(f"the {name} is not a (rows, d) value assembled from columns of the state")
        out.append(cols)
    found = RecognisedRows(out[0], out[1], d, y0.dtype, y0.device)
    found._alive = interp.keep
    return found

"""SDEs whose drift and diffusion are given in closed form, so the whole stepping loop can stay on the chip.

The reference's SDE contract is a pair of Python callables ``f(t, y)``/``g(t, y)`` (torchsde/_core/base_sde.py:
24-64), which forces one trip through HBM per call and per solver stage. A closed-form SDE still implements that
contract with plain torch ops -- it is a valid SDE for the reference, for ``sdeint_adjoint`` and for every solver
here -- and additionally publishes its coefficients, which lets ``sdeint`` run all fixed steps of a forward solve
in ONE kernel launch (``tsde_trajectory_affine_diag``: state in registers, increments from the counter RNG,
HBM touched only for y0 and the requested outputs). The arithmetic is the same chain of single-rounded
operations either way, so both routes return bit-identical trajectories.
"""
import torch
from torch import nn


def publishes_its_own_dynamics(sde):
    """True when the `closed_form()` an SDE object publishes is a statement of the `f` / `g` it would be integrated
    with: the same class defines `closed_form`, `f` and `g` (a subclass that overrides the drift or the diffusion --
    say, makes it time-dependent -- without restating `closed_form` still inherits the parent's coefficients), and
    nothing on the class or the instance adds another drift / diffusion provider (`f_and_g`, `g_prod`,
    `f_and_g_prod`), which `ForwardSDE` would prefer over `f` and `g` (base_sde.py:51-73)."""
    def owner(name):
        if name in getattr(sde, "__dict__", {}):
            return "instance"
        for klass in type(sde).__mro__:
            if name in klass.__dict__:
                return klass
        return None
    home = owner("closed_form")
    if home is None or home == "instance":
        return False
    if owner("f") is not home or owner("g") is not home or owner("closed_form_parameters") is not home:
        return False
    return all(owner(name) is None for name in ("f_and_g", "g_prod", "f_and_g_prod"))


class AffineDiagonalSDE(nn.Module):
    """Diagonal-noise SDE with per-channel affine coefficients:

        f(t, y) = drift_rate * y + drift_shift        g(t, y) = diff_rate * y + diff_shift

    Geometric Brownian motion (shift = 0), Ornstein-Uhlenbeck / Vasicek (diff_rate = 0) and their mixtures.
    Each coefficient is a scalar or a length-``d`` tensor and is registered as a parameter.
    """
    noise_type = "diagonal"

    def __init__(self, drift_rate, drift_shift, diff_rate, diff_shift, sde_type="ito", dtype=None, device=None):
        super().__init__()
        if sde_type not in ("ito", "stratonovich"):
            raise ValueError(f"Expected sde_type 'ito' or 'stratonovich', got {sde_type!r}.")
        self.sde_type = sde_type
        for name, value in (("drift_rate", drift_rate), ("drift_shift", drift_shift), ("diff_rate", diff_rate),
                            ("diff_shift", diff_shift)):
            value = torch.as_tensor(value, dtype=dtype, device=device)
            if not value.is_floating_point():
                value = value.to(torch.get_default_dtype())
            if value.dim() > 1:
                raise ValueError(f"`{name}` must be a scalar or a 1-D tensor over the state channels.")
            setattr(self, name, nn.Parameter(value.detach().clone()))

    def f(self, t, y):
        return self.drift_rate * y + self.drift_shift

    def g(self, t, y):
        return self.diff_rate * y + self.diff_shift

    def closed_form_parameters(self):
        """The four coefficient tensors, in the order the kernels take them."""
        return self.drift_rate, self.drift_shift, self.diff_rate, self.diff_shift

    def closed_form(self, d, dtype, device):
        """Coefficients as contiguous ``(d,)`` tensors, or None if they cannot be served in `dtype` as they are
        (then the stepwise path is used: torch's type promotion would change the arithmetic)."""
        out = []
        for p in (self.drift_rate, self.drift_shift, self.diff_rate, self.diff_shift):
            if p.dtype != dtype or p.device != device or (p.dim() == 1 and p.numel() not in (1, d)):
                return None
            out.append(p.detach().reshape(-1).expand(d).contiguous())
        return ("affine_diagonal",) + tuple(out)


class ElementwiseDiagonalSDE(nn.Module):
    """Diagonal-noise SDE whose drift and diffusion are elementwise expressions, per state channel:

        f(t, y) = f_scale * phi_f(f_rate * y + f_shift) + f_offset
        g(t, y) = g_scale * phi_g(g_rate * y + g_shift) + g_offset

    with ``phi`` one of ``identity, exp, sigmoid, tanh, softplus, sin, cos`` -- e.g. the SDE the reference's own
    benchmark integrates (benchmarks/brownian.py:131-139), ``f = y, g = exp(-y)``:
    ``ElementwiseDiagonalSDE("identity", "exp", diffusion_coefficients=(1., -1., 0., 0.))``; geometric Brownian
    motion; bounded (sigmoid / tanh) diffusions and drifts. Every coefficient is a scalar or a length-``d`` tensor and
    is registered as a parameter. An ordinary module for the reference, for autograd and for ``sdeint_adjoint``;
    forward solves without autograd run all their steps in one launch of ``tsde_trajectory_expr_diag`` (Euler,
    Milstein, midpoint, SRK), the kernel evaluating ``phi`` with the formulas torch uses for it.
    """
    noise_type = "diagonal"
    _FUNCTIONS = {"identity": lambda u: u, "exp": torch.exp, "sigmoid": torch.sigmoid, "tanh": torch.tanh,
                  "softplus": nn.functional.softplus, "sin": torch.sin, "cos": torch.cos}
    _NAMES = ("f_scale", "f_rate", "f_shift", "f_offset", "g_scale", "g_rate", "g_shift", "g_offset")

    def __init__(self, drift="identity", diffusion="identity", drift_coefficients=(1.0, 1.0, 0.0, 0.0),
                 diffusion_coefficients=(1.0, 1.0, 0.0, 0.0), sde_type="ito", dtype=None, device=None):
        super().__init__()
        if sde_type not in ("ito", "stratonovich"):
            raise ValueError(f"Expected sde_type 'ito' or 'stratonovich', got {sde_type!r}.")
        for name in (drift, diffusion):
            if name not in self._FUNCTIONS:
                raise ValueError(f"Expected a function in {sorted(self._FUNCTIONS)}, got {name!r}.")
        if len(drift_coefficients) != 4 or len(diffusion_coefficients) != 4:
            raise ValueError("coefficients are (scale, rate, shift, offset)")
        self.sde_type, self.drift, self.diffusion = sde_type, drift, diffusion
        for name, value in zip(self._NAMES, tuple(drift_coefficients) + tuple(diffusion_coefficients)):
            value = torch.as_tensor(value, dtype=dtype, device=device)
            if not value.is_floating_point():
                value = value.to(torch.get_default_dtype())
            if value.dim() > 1:
                raise ValueError(f"`{name}` must be a scalar or a 1-D tensor over the state channels.")
            setattr(self, name, nn.Parameter(value.detach().clone()))

    def f(self, t, y):
        return self.f_scale * self._FUNCTIONS[self.drift](self.f_rate * y + self.f_shift) + self.f_offset

    def g(self, t, y):
        return self.g_scale * self._FUNCTIONS[self.diffusion](self.g_rate * y + self.g_shift) + self.g_offset

    def closed_form_parameters(self):
        return tuple(getattr(self, name) for name in self._NAMES)

    def closed_form(self, d, dtype, device):
        """("elementwise_diagonal", drift code, diffusion code, eight contiguous (d,) coefficient tensors), or None
        if the coefficients cannot be served in `dtype` as they are."""
        from . import _native
        out = []
        for p in self.closed_form_parameters():
            if p.dtype != dtype or p.device != device or (p.dim() == 1 and p.numel() not in (1, d)):
                return None
            out.append(p.detach().reshape(-1).expand(d).contiguous())
        return ("elementwise_diagonal", _native.FN_CODES[self.drift], _native.FN_CODES[self.diffusion]) + tuple(out)


class MLPDriftDiagonalSDE(nn.Module):
    """Diagonal-noise neural SDE with a two-layer perceptron drift shared by the batch and an elementwise diffusion:

        f(t, y) = lin2(act(lin1(y)))          g(t, y) = diff_rate * y + diff_shift                         ("affine")
                                              g(t, y) = diff_scale * sigmoid(diff_rate * y + diff_shift)   ("sigmoid":
                                              the per-channel diffusion of latent-SDE models, bounded and positive;
                                              `diff_scale` is a fixed number, not a parameter)

    An ordinary module for every solver, for autograd and for ``sdeint_adjoint`` (train it as usual). For SAMPLING --
    forward solves without autograd, Euler, Milstein or midpoint, float32, ``d`` a multiple of 4, ``d, hidden <= 128``
    (``hidden <= 256`` for ``d <= 64``: both weight matrices live in the LDS of a compute unit) -- ``sdeint`` runs the whole solve in one launch of ``tsde_trajectory_mlp_diag``: the state stays in registers, the
    weights in LDS, both layers on the f32 matrix cores. For TRAINING through ``sdeint`` (autograd on, Euler -- with the
    affine diffusion also Milstein --, ``hidden`` a multiple of 4) the backward pass is
    ``tsde_trajectory_mlp_diag_backward`` + ``tsde_gram_partials``: the gradients back-propagation through the stepwise solver gives, without a tape (the forward launch keeps the
    state of every step in HBM: steps x batch x d floats). Results agree with the stepwise path up to the summation
    order of the matrix products (same Brownian path).
    """
    noise_type = "diagonal"
    _ACTIVATIONS = {"tanh": (0, torch.tanh), "softplus": (1, nn.functional.softplus)}

    def __init__(self, d, hidden, activation="softplus", diff_rate=0.0, diff_shift=0.1, sde_type="ito", dtype=None,
                 device=None, diffusion="affine", diff_scale=1.0):
        super().__init__()
        if diffusion not in ("affine", "sigmoid"):
            raise ValueError(f"Expected diffusion 'affine' or 'sigmoid', got {diffusion!r}.")
        self.diffusion, self.diff_scale = diffusion, float(diff_scale)
        if sde_type not in ("ito", "stratonovich"):
            raise ValueError(f"Expected sde_type 'ito' or 'stratonovich', got {sde_type!r}.")
        if activation not in self._ACTIVATIONS:
            raise ValueError(f"Expected activation in {sorted(self._ACTIVATIONS)}, got {activation!r}.")
        self.sde_type, self.activation = sde_type, activation
        self.lin1 = nn.Linear(d, hidden, dtype=dtype, device=device)
        self.lin2 = nn.Linear(hidden, d, dtype=dtype, device=device)
        for name, value in (("diff_rate", diff_rate), ("diff_shift", diff_shift)):
            value = torch.as_tensor(value, dtype=self.lin1.weight.dtype, device=device)
            if value.dim() > 1:
                raise ValueError(f"`{name}` must be a scalar or a 1-D tensor over the state channels.")
            setattr(self, name, nn.Parameter(value.detach().clone()))

    def f(self, t, y):
        return self.lin2(self._ACTIVATIONS[self.activation][1](self.lin1(y)))

    def g(self, t, y):
        u = self.diff_rate * y + self.diff_shift
        return self.diff_scale * torch.sigmoid(u) if self.diffusion == "sigmoid" else u

    def closed_form_parameters(self):
        """The six parameters the differentiable trajectory path returns gradients for."""
        return (self.lin1.weight, self.lin1.bias, self.lin2.weight, self.lin2.bias, self.diff_rate, self.diff_shift)

    def closed_form(self, d, dtype, device):
        """("mlp_diagonal", W1 (d, hidden) input-major, b1, W2 (hidden, d), b2, diff_rate (d,), diff_shift (d,), act,
        (diffusion kind, diff_scale)) for
        the sampling kernel, or None when it does not apply (then the stepwise path runs)."""
        hidden = self.lin1.out_features
        params = list(self.parameters())
        if (dtype != torch.float32 or any(p.dtype != dtype or p.device != device for p in params)
                or self.lin1.in_features != d or d % 4 != 0 or d > 128 or hidden > (256 if d <= 64 else 128)
                or self.lin1.bias is None or self.lin2.bias is None):
            return None
        coefs = []
        for p in (self.diff_rate, self.diff_shift):
            if p.dim() == 1 and p.numel() not in (1, d):
                return None
            coefs.append(p.detach().reshape(-1).expand(d).contiguous())
        return ("mlp_diagonal", self.lin1.weight.detach().t().contiguous(), self.lin1.bias.detach().contiguous(),
                self.lin2.weight.detach().t().contiguous(), self.lin2.bias.detach().contiguous(), coefs[0], coefs[1],
                self._ACTIVATIONS[self.activation][0], (1 if self.diffusion == "sigmoid" else 0, self.diff_scale))

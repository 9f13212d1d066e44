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

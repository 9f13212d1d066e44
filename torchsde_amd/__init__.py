"""torchsde_amd -- the MI355X-native time-stepping hot path of torchsde, behind torchsde's own API."""
from .brownian import (BaseBrownian, BrownianInterval, BrownianPath, BrownianTree, ReverseBrownian,
                       brownian_interval_like)
from .adjoint import sdeint_adjoint
from .closed_form import AffineDiagonalSDE, ElementwiseDiagonalSDE, MLPDriftDiagonalSDE
from .integrate import sdeint
from .sde import BaseSDE, SDEIto, SDEStratonovich
from . import types  # noqa: F401

__version__ = "0.1.0"

"""String enums of the solver contract.

The values are API: they are the strings users pass as ``method=``, ``noise_type``, ``sde_type`` and
``levy_area_approximation=`` (reference: torchsde/settings.py:16-61). Each enum supports attribute
access (``METHODS.euler``), membership (``'euler' in METHODS``) and ``.all()``.
"""


class _StrEnumMeta(type):
    """Metaclass turning a class of string constants into a small immutable enum."""

    def _values(cls):
        return tuple(v for k, v in vars(cls).items() if not k.startswith("_") and isinstance(v, str))

    def all(cls):
        return sorted(cls._values())

    def __contains__(cls, item):
        return item in cls._values()

    def __iter__(cls):
        return iter(cls.all())

    def __str__(cls):
        return str(cls.all())

    __repr__ = __str__


class METHODS(metaclass=_StrEnumMeta):
    euler = "euler"
    milstein = "milstein"
    srk = "srk"
    midpoint = "midpoint"
    heun = "heun"
    euler_heun = "euler_heun"
    log_ode_midpoint = "log_ode"
    reversible_heun = "reversible_heun"
    adjoint_reversible_heun = "adjoint_reversible_heun"


class NOISE_TYPES(metaclass=_StrEnumMeta):  # noqa: N801
    diagonal = "diagonal"
    scalar = "scalar"
    additive = "additive"
    general = "general"


class SDE_TYPES(metaclass=_StrEnumMeta):  # noqa: N801
    ito = "ito"
    stratonovich = "stratonovich"


class LEVY_AREA_APPROXIMATIONS(metaclass=_StrEnumMeta):  # noqa: N801
    none = "none"              # increments only
    space_time = "space-time"  # + exact space-time Levy area H (U)
    davie = "davie"            # + Davie's approximation of the Levy area A
    foster = "foster"          # + Foster's correction of Davie's approximation


class METHOD_OPTIONS(metaclass=_StrEnumMeta):  # noqa: N801
    grad_free = "grad_free"


# Methods whose per-step update is a hand-written HIP kernel in this package (SURVEY.md section 8 a + f).
NATIVE_METHODS = (METHODS.euler, METHODS.milstein, METHODS.srk, METHODS.midpoint, METHODS.heun, METHODS.euler_heun,
                  METHODS.log_ode_midpoint, METHODS.reversible_heun, METHODS.adjoint_reversible_heun)

"""The reference's method-validity matrix (reference tests/test_sdeint.py:101-157 `test_sdeint_run_shape_method`):
every (noise type x sde type x method x Levy approximation x adaptive) combination either runs and returns
(T, batch, d) or raises ValueError exactly when the reference's rules say so (failure rules :124-136,
compatibility table SURVEY.md appendix A.4)."""

import pytest
import torch

from workloads import problems

pytestmark = pytest.mark.gpu
DEV = "cuda"

METHODS = ["euler", "milstein", "srk", "midpoint", "heun", "euler_heun", "log_ode", "reversible_heun"]
ITO_METHODS = {"euler", "milstein", "srk"}
STRAT_METHODS = {"milstein", "midpoint", "heun", "euler_heun", "log_ode", "reversible_heun"}
NO_GENERAL = {"milstein", "srk"}
PROBLEMS = {"diagonal": "gbm", "scalar": "scalar", "additive": "additive", "general": "general"}


def _should_fail(noise, sde_type, method, levy):
    if sde_type == "ito" and method not in ITO_METHODS:
        return True
    if sde_type == "stratonovich" and method not in STRAT_METHODS:
        return True
    if noise == "general" and method in NO_GENERAL:
        return True
    if method == "srk" and levy == "none":
        return True
    if method == "log_ode" and levy in ("none", "space-time"):
        return True
    return False


@pytest.mark.parametrize("adaptive", [False, True])
@pytest.mark.parametrize("levy", ["none", "space-time", "davie", "foster"])
@pytest.mark.parametrize("sde_type", ["ito", "stratonovich"])
@pytest.mark.parametrize("noise", list(PROBLEMS))
def test_method_matrix(noise, sde_type, levy, adaptive):
    import torchsde_amd
    B, d = 4, 4
    m = {"diagonal": d, "scalar": 1, "additive": 3, "general": 4}[noise]
    tag = "ito" if sde_type == "ito" else "strat"
    sde = problems.make(f"{PROBLEMS[noise]}_{tag}", d=d, m=m).to(DEV)
    y0 = torch.full((B, d), 0.1, device=DEV)
    ts = torch.tensor([0.0, 0.2, 0.5], device=DEV)
    for method in METHODS:
        bm = torchsde_amd.BrownianInterval(0.0, 0.5, size=(B, m), device=DEV, dtype=torch.float32, entropy=3,
                                           levy_area_approximation=levy)
        kw = dict(bm=bm, method=method, dt=0.1, adaptive=adaptive, rtol=1e-2, atol=1e-2)
        if _should_fail(noise, sde_type, method, levy):
            with pytest.raises(ValueError):
                with torch.no_grad():
                    torchsde_amd.sdeint(sde, y0, ts, **kw)
            continue
        with torch.no_grad():
            ys = torchsde_amd.sdeint(sde, y0, ts, **kw)
        assert ys.shape == (3, B, d), (method, ys.shape)
        assert torch.isfinite(ys).all(), method
        assert torch.equal(ys[0], y0)


ALL_METHODS = ["blah", "euler", "milstein", "milstein_grad_free", "srk", "euler_heun", "heun", "midpoint", "log_ode"]


@pytest.mark.parametrize("logqp", [False, True])
@pytest.mark.parametrize("adaptive", [False, True])
@pytest.mark.parametrize("sde_type", ["ito", "stratonovich"])
@pytest.mark.parametrize("noise", list(PROBLEMS))
def test_method_matrix_without_a_brownian_motion_with_logqp_and_other_drifts(noise, sde_type, adaptive, logqp):
    """The rest of `test_sdeint_run_shape_method` (reference tests/test_sdeint.py:101-216): `bm=None` (sdeint builds the
    BrownianInterval the method needs, sdeint.py:246-270), an unknown method name, derivative-free Milstein, `logqp=True`
    (shape (T - 1, batch) of the log-ratio) and `names={"drift": "h"}`."""
    import warnings
    import torchsde_amd
    B, d, T = 4, 4, 3
    m = {"diagonal": d, "scalar": 1, "additive": 3, "general": 4}[noise]
    tag = "ito" if sde_type == "ito" else "strat"
    sde = problems.make(f"{PROBLEMS[noise]}_{tag}", d=d, m=m).to(DEV)
    if not hasattr(sde, "h"):
        sde.h = lambda t, y: -0.5 * y
    y0 = torch.ones(B, d, device=DEV)
    ts = torch.linspace(0.0, 0.5, T, device=DEV)
    for name in ALL_METHODS:
        method, options = ("milstein", {"grad_free": True}) if name == "milstein_grad_free" else (name, {})
        # with bm=None the solver gets the Levy area it needs, so only the method / type rules can fail
        fail = name == "blah" or _should_fail(noise, sde_type, method, "foster")
        for names in (None, {"drift": "h"}):
            kw = dict(method=method, dt=0.1, adaptive=adaptive, rtol=1e-2, atol=1e-2, logqp=logqp, options=options,
                      names=names)
            with torch.no_grad(), warnings.catch_warnings():
                warnings.simplefilter("ignore")
                if fail:
                    with pytest.raises(ValueError):
                        torchsde_amd.sdeint(sde, y0, ts, **kw)
                    continue
                ans = torchsde_amd.sdeint(sde, y0, ts, **kw)
            if logqp:
                ans, log_ratio = ans
                assert log_ratio.shape == (T - 1, B), (name, log_ratio.shape)
            assert ans.shape == (T, B, d) and torch.isfinite(ans).all(), name

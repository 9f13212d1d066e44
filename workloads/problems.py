"""Test SDEs (one per noise type), defined without importing either torchsde or torchsde_amd so that the
same objects can be handed to the reference (golden generation), to the oracle and to the HIP path.

They cover the families the reference's own tests use (reference tests/problems.py): geometric Brownian
motion for diagonal noise (closed-form solution available), a trigonometric scalar-noise SDE, a
time-dependent additive-noise SDE, small MLPs for general/diagonal noise, and the README quick example.
"""
import math

import torch
from torch import nn


def _sigmoid_randn(gen, *shape):
    return torch.sigmoid(torch.randn(*shape, generator=gen, dtype=torch.float64))


class GBMDiag(nn.Module):
    """dy = mu*y dt + sigma*y dW (Ito) / with the -sigma^2 y/2 correction (Stratonovich); non-exploding."""
    noise_type = "diagonal"

    def __init__(self, d, sde_type="ito", seed=0, dtype=torch.float32):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        sigma = _sigmoid_randn(gen, d)
        mu = -sigma ** 2 - _sigmoid_randn(gen, d)
        self.mu = nn.Parameter(mu.to(dtype))
        self.sigma = nn.Parameter(sigma.to(dtype))
        self.sde_type = sde_type

    def f(self, t, y):
        if self.sde_type == "ito":
            return self.mu * y
        return self.mu * y - .5 * (self.sigma ** 2) * y

    def g(self, t, y):
        return self.sigma * y

    def h(self, t, y):
        """A prior drift for `logqp=True` (KL between this SDE and the one with drift h; sdeint.py:142-144)."""
        return -0.5 * y + 0.05

    def exact(self, y0, t, W_t):
        """Closed form y0 * exp((mu - sigma^2/2) t + sigma W_t) on the same Brownian path."""
        return y0 * torch.exp((self.mu - 0.5 * self.sigma ** 2) * t + self.sigma * W_t)


class ScalarTrig(nn.Module):
    """Scalar noise: dy = -p^2 sin(y) cos^3(y) dt + p cos^2(y) dW (Ito form)."""
    noise_type = "scalar"

    def __init__(self, d, sde_type="ito", seed=1, dtype=torch.float32):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.p = nn.Parameter(_sigmoid_randn(gen, d).to(dtype))
        self.sde_type = sde_type

    def f(self, t, y):
        if self.sde_type == "ito":
            return -self.p ** 2. * torch.sin(y) * torch.cos(y) ** 3.
        return torch.zeros_like(y)

    def g(self, t, y):
        return (self.p * torch.cos(y) ** 2).unsqueeze(dim=-1)


class AdditiveDecay(nn.Module):
    """Additive noise with time-dependent diffusion: g does not depend on y."""
    noise_type = "additive"

    def __init__(self, d, m, sde_type="ito", seed=2, dtype=torch.float32):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.a = nn.Parameter(_sigmoid_randn(gen, d).to(dtype))
        self.b = nn.Parameter(_sigmoid_randn(gen, d).to(dtype))
        self.m = m
        self.sde_type = sde_type

    def f(self, t, y):
        return self.b / torch.sqrt(1. + t) - y / (2. + 2. * t)

    def g(self, t, y):
        fill = self.a * self.b / torch.sqrt(1. + t)
        return fill.unsqueeze(0).unsqueeze(-1).repeat(y.size(0), 1, self.m)


class AdditiveShared(nn.Module):
    """Additive noise the way the reference's examples return it (examples/cont_ddpm.py-style constant diffusion): ONE
    (d, m) matrix for every batch row, handed back as `sigma.expand(B, d, m)` -- no per-row copies. Linear mean-reverting
    drift."""
    noise_type = "additive"

    def __init__(self, d, m, sde_type="ito", seed=4, dtype=torch.float32):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.rate = nn.Parameter(_sigmoid_randn(gen, d).to(dtype))
        self.sigma = nn.Parameter((0.5 * torch.randn(d, m, generator=gen) / m ** 0.5).to(dtype))
        self.sde_type = sde_type

    def f(self, t, y):
        return -self.rate * y

    def g(self, t, y):
        return self.sigma.expand(y.size(0), -1, -1)


def _mlp(gen, sizes, dtype, final=None):
    layers = []
    for i, (a, b) in enumerate(zip(sizes[:-1], sizes[1:])):
        lin = nn.Linear(a, b)
        with torch.no_grad():
            lin.weight.copy_((torch.randn(b, a, generator=gen, dtype=torch.float64) / math.sqrt(a)).to(dtype))
            lin.bias.copy_((0.1 * torch.randn(b, generator=gen, dtype=torch.float64)).to(dtype))
        layers.append(lin.to(dtype))
        if i < len(sizes) - 2:
            layers.append(nn.Softplus())
    if final is not None:
        layers.append(final)
    return nn.Sequential(*layers)


class MLPGeneral(nn.Module):
    """General noise: drift and diffusion are small time-dependent MLPs; g has shape (B, d, m)."""
    noise_type = "general"

    def __init__(self, d, m, sde_type="ito", seed=3, hidden=8, dtype=torch.float32):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.d, self.m, self.sde_type = d, m, sde_type
        self.f_net = _mlp(gen, (d + 1, hidden, d), dtype)
        self.g_net = _mlp(gen, (d + 1, hidden, d * m), dtype, final=nn.Sigmoid())

    def _ty(self, t, y):
        return torch.cat([t.expand(y.size(0), 1).to(y.dtype), y], dim=1)

    def f(self, t, y):
        return self.f_net(self._ty(t, y))

    def g(self, t, y):
        return self.g_net(self._ty(t, y)).reshape(y.size(0), self.d, self.m)

    def h(self, t, y):
        """Prior drift for `logqp=True` (general noise: u = pinv(g) (f - h), base_sde.py:283-287)."""
        return -y


class MLPNetDiag(nn.Module):
    """Diagonal noise with drift AND diffusion nets of (t, y), the shape of the reference's NeuralDiagonal
    (tests/problems.py:135-162): g = 0.1 * sigmoid-closed net. (Such a g mixes channels, so it is "diagonal" only in
    the sense of the contract -- (B, d) times dW (B, d) -- exactly like the reference's problem.)"""
    noise_type = "diagonal"

    def __init__(self, d, sde_type="ito", seed=5, hidden=8, dtype=torch.float32):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.sde_type = sde_type
        self.f_net = _mlp(gen, (d + 1, hidden, d), dtype)
        self.g_net = _mlp(gen, (d + 1, hidden, d), dtype, final=nn.Sigmoid())

    def _ty(self, t, y):
        return torch.cat([t.expand(y.size(0), 1).to(y.dtype), y], dim=1)

    def f(self, t, y):
        return self.f_net(self._ty(t, y))

    def g(self, t, y):
        return 0.1 * self.g_net(self._ty(t, y))


class MLPNetScalar(MLPNetDiag):
    """... and the reference's NeuralScalar (tests/problems.py:165-192): one Brownian channel, g of shape (B, d, 1)."""
    noise_type = "scalar"

    def g(self, t, y):
        return 0.1 * self.g_net(self._ty(t, y)).unsqueeze(-1)


class MLPNetAdditive(nn.Module):
    """... and the reference's NeuralAdditive (tests/problems.py:195-224): the drift a net of (t, y), the diffusion a net of
    t alone reshaped to (B, d, m)."""
    noise_type = "additive"

    def __init__(self, d, m, sde_type="ito", seed=6, hidden=8, dtype=torch.float32):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.d, self.m, self.sde_type = d, m, sde_type
        self.f_net = _mlp(gen, (d + 1, hidden, d), dtype)
        self.g_net = _mlp(gen, (1, hidden, d * m), dtype, final=nn.Sigmoid())

    def f(self, t, y):
        return self.f_net(torch.cat([t.expand(y.size(0), 1).to(y.dtype), y], dim=1))

    def g(self, t, y):
        return self.g_net(t.expand(y.size(0), 1).to(y.dtype)).view(y.size(0), self.d, self.m)


class MLPDiag(nn.Module):
    """Diagonal noise with an elementwise diffusion g_i(y_i) (a valid diagonal SDE for Milstein/adjoint)."""
    noise_type = "diagonal"

    def __init__(self, d, sde_type="ito", seed=4, hidden=8, dtype=torch.float32):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.sde_type = sde_type
        self.f_net = _mlp(gen, (d + 1, hidden, d), dtype)
        self.w = nn.Parameter(torch.randn(d, generator=gen, dtype=torch.float64).to(dtype))
        self.b = nn.Parameter((0.1 * torch.randn(d, generator=gen, dtype=torch.float64)).to(dtype))

    def f(self, t, y):
        return self.f_net(torch.cat([t.expand(y.size(0), 1).to(y.dtype), y], dim=1))

    def g(self, t, y):
        return 0.1 * torch.sigmoid(self.w * y + self.b)

    def h(self, t, y):
        """Prior drift for `logqp=True`."""
        return 0.5 * torch.tanh(y) - 0.1 * torch.cos(t)

    def prior(self, t, y):
        """A second prior drift, reached through `names={"prior_drift": "prior"}`."""
        return -y * torch.sigmoid(self.b)


class LatentDiag(nn.Module):
    """Latent-SDE-style diagonal Ito SDE (BASELINE configs[4]; cf. reference examples/latent_sde_lorenz.py:122-148):
    drift Linear(d,d)-Softplus-Linear(d,d), elementwise diffusion 0.1 * sigmoid(w*y + b)."""
    noise_type, sde_type = "diagonal", "ito"

    def __init__(self, d, seed=0):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.net = nn.Sequential(nn.Linear(d, d), nn.Softplus(), nn.Linear(d, d))
        with torch.no_grad():
            for p in self.net.parameters():
                p.copy_(torch.randn(p.shape, generator=gen) / (d ** 0.5))
        self.w = nn.Parameter(torch.randn(d, generator=gen))
        self.b = nn.Parameter(0.1 * torch.randn(d, generator=gen))

    def f(self, t, y):
        return self.net(y)

    def g(self, t, y):
        return 0.1 * torch.sigmoid(self.w * y + self.b)


class LatentDiagStrat(LatentDiag):
    """The same dynamics read as a Stratonovich SDE: what `method="reversible_heun"` (reversible_heun.py:48-73, Stratonovich
    only) integrates -- the reference's recommended training pair at the configs[4] shape."""
    sde_type = "stratonovich"


class LatentDiagLogqp(LatentDiag):
    """... with a prior drift `h` (an Ornstein-Uhlenbeck pull, cf. examples/latent_sde.py:119-121), so that
    `sdeint(..., logqp=True)` (base_sde.py:240-306) can integrate the KL column u = (f - h) / g beside the state."""

    def __init__(self, d, seed=0):
        super().__init__(d, seed)
        self.theta = nn.Parameter(torch.full((d,), 0.5))

    def h(self, t, y):
        return -self.theta * y


class StochasticLorenz:
    """The stochastic Lorenz system of the reference's examples/latent_sde_lorenz.py:56-86 with that example's constants,
    written the way that example writes it -- the state's columns taken apart with `torch.split` and put back together with
    `torch.cat` -- because that is the kind of user code the column interpreter (recognise_rows.py) has to follow: a
    row-coupled system with diagonal noise, no parameters, not an nn.Module."""
    noise_type = "diagonal"
    sde_type = "ito"

    def __init__(self, drift_constants=(10.0, 28.0, 8.0 / 3.0), noise_constants=(0.1, 0.28, 0.3)):
        self.sigma, self.rho, self.beta = drift_constants
        self.noise_constants = noise_constants

    @staticmethod
    def _columns(y):
        return torch.split(y, split_size_or_sections=(1, 1, 1), dim=1)

    def f(self, t, y):
        x, yy, z = self._columns(y)
        dx = self.sigma * (yy - x)
        dy = self.rho * x - yy - x * z
        dz = x * yy - self.beta * z
        return torch.cat([dx, dy, dz], dim=1)

    def g(self, t, y):
        scaled = [column * c for column, c in zip(self._columns(y), self.noise_constants)]
        return torch.cat(scaled, dim=1)

    def to(self, device):
        return self

    def parameters(self):
        return iter(())


class LipSwish(nn.Module):
    """examples/sde_gan.py:44-47."""

    def forward(self, x):
        return 0.909 * torch.nn.functional.silu(x)


def sde_gan_mlp(in_size, out_size, mlp_size, num_layers, tanh):
    """The layer list of the MLP of examples/sde_gan.py:50-66 (restated)."""
    model = [nn.Linear(in_size, mlp_size), LipSwish()]
    for _ in range(num_layers - 1):
        model += [nn.Linear(mlp_size, mlp_size), LipSwish()]
    model.append(nn.Linear(mlp_size, out_size))
    if tanh:
        model.append(nn.Tanh())
    return nn.Sequential(*model)


class SdeGanGenerator(nn.Module):
    """The generator SDE of the reference's examples/sde_gan.py:77-101 (restated): Stratonovich, general noise, drift and
    diffusion LipSwish MLPs of cat([t, x]) closed by tanh. The example's sizes: hidden 16, noise 3, mlp 16, one hidden layer
    (examples/sde_gan.py:336-340), batch 1024, 64 output times a unit step apart, `reversible_heun` + `adjoint_reversible_heun`
    (:129-130)."""
    sde_type, noise_type = "stratonovich", "general"

    def __init__(self, noise_size=3, hidden_size=16, mlp_size=16, num_layers=1, seed=0):
        super().__init__()
        self._noise_size, self._hidden_size = noise_size, hidden_size
        state = torch.random.get_rng_state()
        torch.manual_seed(seed)
        self._drift = sde_gan_mlp(1 + hidden_size, hidden_size, mlp_size, num_layers, tanh=True)
        self._diffusion = sde_gan_mlp(1 + hidden_size, hidden_size * noise_size, mlp_size, num_layers, tanh=True)
        torch.random.set_rng_state(state)

    def f_and_g(self, t, x):
        t = t.expand(x.size(0), 1)
        tx = torch.cat([t, x], dim=1)
        return self._drift(tx), self._diffusion(tx).view(x.size(0), self._hidden_size, self._noise_size)


class ExpDiffusion(nn.Module):
    """The SDE the reference's own benchmark integrates (benchmarks/brownian.py:131-139): f = y, g = exp(-y),
    diagonal Ito noise, no parameters."""
    noise_type, sde_type = "diagonal", "ito"

    def f(self, t, y):
        return y

    def g(self, t, y):
        return torch.exp(-y)


class ScheduledDiag(nn.Module):
    """Time enters through a schedule only: dy = -1/2 beta(t) y dt + sqrt(beta(t)) sigma dW, beta(t) = b0 + t (b1 - b0) --
    the forward SDE of a variance-preserving diffusion model, per channel (cf. the reference's examples/cont_ddpm.py)."""
    noise_type, sde_type = "diagonal", "ito"

    def __init__(self, d, seed=3):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.b0 = nn.Parameter(torch.tensor(0.1))
        self.b1 = nn.Parameter(torch.tensor(4.0))
        self.sigma = nn.Parameter(_sigmoid_randn(gen, d).to(torch.float32))

    def beta(self, t):
        return self.b0 + t * (self.b1 - self.b0)

    def f(self, t, y):
        return -0.5 * self.beta(t) * y

    def g(self, t, y):
        return torch.sqrt(self.beta(t)) * self.sigma * torch.ones_like(y)


class DoubleWell(nn.Module):
    """dy = (y - y^3) dt + sigma (1 + y^2 / 2) dW, per channel: a drift that is a SUM of functions of the state and a
    quadratic diffusion -- polynomials (the textbook bistable system)."""
    noise_type, sde_type = "diagonal", "ito"

    def __init__(self, d, seed=5):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.sigma = nn.Parameter((0.3 * _sigmoid_randn(gen, d)).to(torch.float32))

    def f(self, t, y):
        return y - y ** 3

    def g(self, t, y):
        return self.sigma * (1.0 + 0.5 * y * y)


class Logistic(nn.Module):
    """dy = r y (1 - y / K) dt + sigma y dW per channel: a drift that is a PRODUCT of two affine functions of the state
    (stochastic logistic growth); Ito or Stratonovich."""
    noise_type = "diagonal"

    def __init__(self, d, sde_type="ito", seed=6):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.sde_type = sde_type
        self.r = nn.Parameter((0.5 + _sigmoid_randn(gen, d)).to(torch.float32))
        self.K = nn.Parameter((1.0 + _sigmoid_randn(gen, d)).to(torch.float32))
        self.sigma = nn.Parameter((0.4 * _sigmoid_randn(gen, d)).to(torch.float32))

    def f(self, t, y):
        return self.r * y * (1.0 - y / self.K)

    def g(self, t, y):
        return self.sigma * y


class ReadmeSDE(nn.Module):
    """The README quick example: general Ito noise, linear drift, linear diffusion reshaped to (B, d, m)."""
    noise_type = "general"
    sde_type = "ito"

    def __init__(self, d=3, m=2, seed=5, dtype=torch.float32):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        self.d, self.m = d, m
        self.mu = _mlp(gen, (d, d), dtype)
        self.sigma = _mlp(gen, (d, d * m), dtype)

    def f(self, t, y):
        return self.mu(y)

    def g(self, t, y):
        return self.sigma(y).view(y.size(0), self.d, self.m)


# ---- the same diagonal SDE exposed through the other provider combinations (base_sde.py:51-73) -----------
class GBMViaFAndG(GBMDiag):
    def f_and_g(self, t, y):
        return GBMDiag.f(self, t, y), GBMDiag.g(self, t, y)


class GBMViaGProd(GBMDiag):
    def g_prod(self, t, y, v):
        return GBMDiag.g(self, t, y) * v


class GBMViaFAndGProd(GBMDiag):
    def f_and_g_prod(self, t, y, v):
        return GBMDiag.f(self, t, y), GBMDiag.g(self, t, y) * v


def make(name, dtype=torch.float32, **kw):
    table = {
        "gbm_ito": lambda: GBMDiag(kw.get("d", 4), "ito", dtype=dtype),
        "gbm_strat": lambda: GBMDiag(kw.get("d", 4), "stratonovich", dtype=dtype),
        "scalar_ito": lambda: ScalarTrig(kw.get("d", 4), "ito", dtype=dtype),
        "scalar_strat": lambda: ScalarTrig(kw.get("d", 4), "stratonovich", dtype=dtype),
        "additive_ito": lambda: AdditiveDecay(kw.get("d", 4), kw.get("m", 3), "ito", dtype=dtype),
        "additive_strat": lambda: AdditiveDecay(kw.get("d", 4), kw.get("m", 3), "stratonovich", dtype=dtype),
        "additive_shared_ito": lambda: AdditiveShared(kw.get("d", 4), kw.get("m", 4), "ito", dtype=dtype),
        "general_ito": lambda: MLPGeneral(kw.get("d", 4), kw.get("m", 4), "ito", dtype=dtype),
        "general_strat": lambda: MLPGeneral(kw.get("d", 4), kw.get("m", 4), "stratonovich", dtype=dtype),
        "general_odd_ito": lambda: MLPGeneral(kw.get("d", 3), kw.get("m", 5), "ito", dtype=dtype),
        "netdiag_ito": lambda: MLPNetDiag(kw.get("d", 4), "ito", hidden=kw.get("hidden", 8), dtype=dtype),
        "netdiag_strat": lambda: MLPNetDiag(kw.get("d", 4), "stratonovich", hidden=kw.get("hidden", 8), dtype=dtype),
        "netscalar_ito": lambda: MLPNetScalar(kw.get("d", 4), "ito", hidden=kw.get("hidden", 8), dtype=dtype),
        "netscalar_strat": lambda: MLPNetScalar(kw.get("d", 4), "stratonovich", hidden=kw.get("hidden", 8), dtype=dtype),
        "netadditive_ito": lambda: MLPNetAdditive(kw.get("d", 4), kw.get("m", 3), "ito", hidden=kw.get("hidden", 8), dtype=dtype),
        "netadditive_strat": lambda: MLPNetAdditive(kw.get("d", 4), kw.get("m", 3), "stratonovich", hidden=kw.get("hidden", 8),
                                                    dtype=dtype),
        "mlpdiag_ito": lambda: MLPDiag(kw.get("d", 4), "ito", dtype=dtype),
        "mlpdiag_strat": lambda: MLPDiag(kw.get("d", 4), "stratonovich", dtype=dtype),
        "readme": lambda: ReadmeSDE(dtype=dtype),
    }
    return table[name]()


class PartialDependence(nn.Module):
    """Diagonal Ito SDEs whose drift/diffusion ignore part of their inputs, with a mix of trainable, frozen and unused
    parameters (the situations of the reference's BasicSDE1-4, tests/problems.py:258-328):

      kind "state"     f, g depend on t, y, a trainable and a frozen parameter
      kind "params"    f, g depend on the parameters only (not on y or t)
      kind "frozen"    like "params", but every parameter that is used is frozen
      kind "constant"  f, g are constants
    An auxiliary drift `h` exists for `names={"drift": "h"}`.
    """
    noise_type, sde_type = "diagonal", "ito"

    def __init__(self, d=10, kind="state", seed=11, dtype=torch.float32):
        super().__init__()
        gen = torch.Generator().manual_seed(seed)
        used_trainable = kind != "frozen"
        self.kind = kind
        self.scale = nn.Parameter(torch.randn(1, d, generator=gen).to(dtype), requires_grad=used_trainable)
        self.frozen = nn.Parameter(torch.randn(1, d, generator=gen).to(dtype), requires_grad=False)
        self.spare_frozen = nn.Parameter(torch.randn(1, d, generator=gen).to(dtype), requires_grad=False)
        self.spare_trainable = nn.Parameter(torch.randn(1, d, generator=gen).to(dtype), requires_grad=True)

    def f(self, t, y):
        if self.kind == "state":
            return 0.2 * self.scale * torch.sin(y) + 0.1 * torch.cos(y * y) + torch.cos(t) + 0.1 * self.frozen * y
        if self.kind == "constant":
            return torch.full_like(y, 0.1)
        return 0.2 * self.scale + 0.1 * self.frozen + torch.zeros_like(y)

    def g(self, t, y):
        if self.kind == "state":
            return (torch.sigmoid(0.3 * self.scale * torch.cos(y) + torch.sin(t)) + torch.sigmoid(self.frozen * y)
                    + 0.1)
        if self.kind == "constant":
            return torch.full_like(y, 0.6)
        return torch.sigmoid(0.3 * self.scale) + torch.sigmoid(self.frozen) + torch.zeros_like(y) + 0.1

    def h(self, t, y):
        return torch.sigmoid(y)

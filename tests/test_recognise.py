"""recognise.py on CPU tensors (the interpretation is plain torch: only the solvers' use of it needs the GPU): every form
it follows evaluates to what the user's code computes, the coefficients are the live values, and everything it must
not follow is refused with the reason."""
import pytest
import torch
import torch.nn.functional as F
from torch import nn

from torchsde_amd import recognise
from torchsde_amd.sde import ForwardSDE
from workloads import problems

D = 6
PHI = {"identity": lambda u: u, "exp": torch.exp, "sigmoid": torch.sigmoid, "tanh": torch.tanh, "softplus": F.softplus,
       "sin": torch.sin, "cos": torch.cos}


def _value(form, y):
    """scale * phi(rate * y + shift) + offset with None as the neutral element (what the kernels evaluate)."""
    if form.phi == "poly3":
        c3, c2, c1, c0 = (0.0 if c is None else c for c in (form.scale, form.rate, form.shift, form.offset))
        return ((c3 * y + c2) * y + c1) * y + c0
    if form.constant():
        c = recognise._add(form.shift, form.offset)
        return torch.zeros_like(y) + (0.0 if c is None else c)
    one = lambda c, n: n if c is None else c      # noqa: E731
    return one(form.scale, 1.0) * PHI[form.phi](one(form.rate, 1.0) * y + one(form.shift, 0.0)) + one(form.offset, 0.0)


class _M(nn.Module):
    noise_type, sde_type = "diagonal", "ito"

    def __init__(self, f, g):
        super().__init__()
        gen = torch.Generator().manual_seed(1)
        self.mu = nn.Parameter(torch.randn(D, generator=gen))
        self.sigma = nn.Parameter(torch.rand(D, generator=gen))
        self.w = nn.Parameter(torch.randn(1, D, generator=gen))
        self.b = torch.randn(D, generator=gen)
        self._f, self._g = f, g

    def f(self, t, y):
        return self._f(self, t, y)

    def g(self, t, y):
        return self._g(self, t, y)


FITS = {
    "gbm": (lambda s, t, y: s.mu * y, lambda s, t, y: y * s.sigma, "affine_diagonal", True),
    "reference benchmark": (lambda s, t, y: y, lambda s, t, y: torch.exp(-y), "elementwise_diagonal", True),
    "ornstein-uhlenbeck": (lambda s, t, y: s.sigma * (s.mu - y), lambda s, t, y: s.sigma.expand_as(y), "affine_diagonal", False),
    "latent diffusion": (lambda s, t, y: -0.5 * y + s.b, lambda s, t, y: 0.1 * torch.sigmoid(s.w * y + s.b),
                         "elementwise_diagonal", True),
    "ones_like": (lambda s, t, y: torch.tanh(y) * 2 - 1, lambda s, t, y: torch.ones_like(y) * s.sigma, "elementwise_diagonal", True),
    "stratonovich gbm": (lambda s, t, y: s.mu * y - 0.5 * s.sigma ** 2 * y, lambda s, t, y: s.sigma * y, "affine_diagonal", False),
    "softplus / sin": (lambda s, t, y: F.softplus(y / 2.0), lambda s, t, y: 0.3 - torch.sin(y), "elementwise_diagonal", False),
    "constants": (lambda s, t, y: torch.zeros_like(y), lambda s, t, y: torch.full_like(y, 0.3), "affine_diagonal", True),
    "negation": (lambda s, t, y: -(s.mu * y + 1.0), lambda s, t, y: (1.0 - y) * 0.5, "affine_diagonal", False),
    "cos, alpha": (lambda s, t, y: torch.add(s.b, y, alpha=2.0), lambda s, t, y: torch.cos(y * s.mu) / 4, "elementwise_diagonal", False),
}


@pytest.mark.parametrize("name", sorted(FITS))
def test_followed_forms_evaluate_to_the_users_code(name):
    f, g, kernel, exact = FITS[name]
    sde = _M(f, g)
    y, t = torch.randn(16, D), torch.tensor(0.3)
    found = recognise.recognise(ForwardSDE(sde), t, y)
    torch.testing.assert_close(_value(found.f, y), sde.f(t, y), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(_value(found.g, y), sde.g(t, y), rtol=1e-6, atol=1e-6)
    spec = found.spec()
    assert spec[0] == kernel and found.exact == exact
    assert all(c.shape == (D,) and c.is_contiguous() and c.dtype == y.dtype for c in spec[1:] if torch.is_tensor(c))
    with torch.no_grad():                      # the next interpretation reads the new parameter values
        sde.mu.mul_(2.0)
        sde.sigma.add_(0.1)
    again = recognise.recognise(ForwardSDE(sde), t, y)
    torch.testing.assert_close(_value(again.f, y), sde.f(t, y), rtol=1e-6, atol=1e-6)
    assert again.structure() == found.structure()


REFUSED = {
    "depends on t": (lambda s, t, y: torch.sin(t) * y, "depends on t"),
    "reads t on the host": (lambda s, t, y: float(t) * y, "reads t on the host"),
    "matrix product": (lambda s, t, y: y @ torch.eye(D), "drift is not"),
    "nested functions": (lambda s, t, y: torch.tanh(torch.exp(y)), "nested"),
    "per-row constant": (lambda s, t, y: y * torch.ones(16, 1), "RuntimeError"),
    "reduction over the batch": (lambda s, t, y: y - y.mean(0), "aten::mean"),
    "in-place update": (lambda s, t, y: y.mul_(2), "in-place"),
    "two functions summed": (lambda s, t, y: torch.tanh(y) + y, "sum of two different"),
    "unsupported function": (lambda s, t, y: torch.relu(y), "aten::relu"),
    "slice of the state": (lambda s, t, y: y[:, :1] * s.mu, "aten::slice"),
    "constant with the probe's rows": (lambda s, t, y: y + torch.zeros(y.shape), "not one value per channel"),
    # kwargs the handlers do not model change what the operator computes
    "floor division": (lambda s, t, y: torch.div(y, 2, rounding_mode="floor"), "rounding_mode"),
    "trunc division": (lambda s, t, y: torch.div(y, s.mu, rounding_mode="trunc"), "rounding_mode"),
    # ... and calls that leave something behind run once per solve here, once per step in the reference
    "random draw": (lambda s, t, y: y + torch.randn_like(s.b), "draws random numbers"),
    "dropout": (lambda s, t, y: F.dropout(y, 0.5, training=True), "empty_like of the state|draws random numbers"),
    "in-place write to a buffer": (lambda s, t, y: y * s.b.add_(1.0), "existed before f and g were called"),
    "in-place write through a view": (lambda s, t, y: y * s.b[:].mul_(2.0), "existed before f and g were called"),
}


@pytest.mark.parametrize("name", sorted(REFUSED))
def test_everything_else_is_refused_with_its_reason(name):
    f, reason = REFUSED[name]
    sde = _M(f, lambda s, t, y: y)
    with pytest.raises(recognise.NotElementwise, match=reason):
        recognise.recognise(ForwardSDE(sde), torch.tensor(0.3), torch.randn(16, D))


def test_in_place_arithmetic_on_tensors_the_code_made_itself_is_followed():
    """`out = torch.zeros(d); out += mu` inside f is the user's way of writing a sum: nothing outlives the call."""
    def f(s, t, y):
        rate = torch.zeros(D)
        rate += s.mu.detach()
        rate.mul_(2.0)
        return rate * y
    sde = _M(f, lambda s, t, y: y * s.sigma)
    found = recognise.recognise(ForwardSDE(sde), torch.tensor(0.3), torch.randn(16, D))
    torch.testing.assert_close(found.spec()[1], 2.0 * sde.mu.detach())


def test_true_division_with_an_explicit_none_rounding_mode_is_followed():
    sde = _M(lambda s, t, y: torch.div(y, 4.0, rounding_mode=None), lambda s, t, y: y * s.sigma)
    found = recognise.recognise(ForwardSDE(sde), torch.tensor(0.3), torch.randn(16, D))
    torch.testing.assert_close(found.spec()[1], torch.full((D,), 0.25))


def test_two_probe_heights_expose_code_that_uses_the_batch_size():
    """The interpretation runs on a probe of a few rows; whatever the code derives from `y.shape[0]` is evaluated for the
    probe. Two heights give two sets of coefficients -- what the trust check of `_integrate_recognised` compares."""
    sde = _M(lambda s, t, y: -y / y.shape[0], lambda s, t, y: y * s.sigma)
    y, t = torch.randn(16, D), torch.tensor(0.3)
    a = recognise.recognise(ForwardSDE(sde), t, y).spec()
    b = recognise.recognise(ForwardSDE(sde), t, y, rows=5).spec()
    assert not torch.equal(a[1], b[1])
    plain = _M(lambda s, t, y: s.mu * y, lambda s, t, y: y * s.sigma)
    a = recognise.recognise(ForwardSDE(plain), t, y).spec()
    b = recognise.recognise(ForwardSDE(plain), t, y, rows=5).spec()
    assert all(torch.equal(u, v) for u, v in zip(a[1:], b[1:]))


def test_perceptron_drift_hands_back_the_users_own_parameters():
    sde = problems.LatentDiag(8)
    y = torch.randn(16, 8)
    found = recognise.recognise(ForwardSDE(sde), torch.tensor(0.0), y)
    assert found.perceptron and found.structure()[0][:2] == ("perceptron", "softplus")
    kind, w1t, b1, w2t, b2, rate, shift, act, (diffusion, amplitude) = found.spec()
    assert kind == "mlp_diagonal" and act == 1 and diffusion == 1 and amplitude == 0.1
    torch.testing.assert_close(F.softplus(y @ w1t + b1) @ w2t + b2, sde.f(None, y))
    torch.testing.assert_close(amplitude * torch.sigmoid(rate * y + shift), sde.g(None, y))
    own = found.perceptron_parameters()
    params = list(sde.parameters())
    assert len(own) == 6 and all(any(o is p for p in params) for o in own)

    class Residual(problems.LatentDiag):
        def f(self, t, y):
            return self.net(y) - y

    with pytest.raises(recognise.NotElementwise, match="output of the drift network"):
        recognise.recognise(ForwardSDE(Residual(8)), torch.tensor(0.0), y)
    deep = problems.LatentDiag(8)
    deep.net = nn.Sequential(nn.Linear(8, 8), nn.Tanh(), nn.Linear(8, 8), nn.Tanh(), nn.Linear(8, 8))
    # a three-layer drift is FOLLOWED (its middle layer kept: the reversible-Heun kernels take such nets, recognise.deep_spec) --
    # but the two-layer perceptron-drift kernels do not evaluate it
    found = recognise.recognise(ForwardSDE(deep), torch.tensor(0.0), y)
    assert found.perceptron and len(found.f.mids) == 1 and found.f.mids[0][0] is deep.net[2].weight and not found.f.classic()
    with pytest.raises(recognise.NotElementwise):
        found.spec()
    # a drift net that takes t (`cat([t, y])`, like the reference's NeuralDiagonal) beside an ELEMENTWISE diffusion: followed,
    # but neither kernel family evaluates it (the perceptron-drift kernels have no time input, the neural-SDE kernel wants
    # a diffusion net)
    mlp_with_time = problems.make("mlpdiag_ito", d=8)
    found = recognise.recognise(ForwardSDE(mlp_with_time), torch.tensor(0.0), y)
    assert found.perceptron and found.f.wt is not None
    with pytest.raises(recognise.NotElementwise, match="takes t"):
        found.spec()


def test_affine_leaves_are_the_users_tensors_or_nothing():
    """Training route: coefficients must be tensors the user's code produced (a graph behind them), not folded here."""
    gbm = problems.make("gbm_ito", d=D)
    found = recognise.recognise(ForwardSDE(gbm), torch.tensor(0.0), torch.randn(16, D), differentiable=True)
    leaves = found.affine_leaves()
    assert leaves[0] is gbm.mu and leaves[2] is gbm.sigma and leaves[1].item() == 0.0 and leaves[3].item() == 0.0
    derived = _M(lambda s, t, y: -s.mu * y, lambda s, t, y: s.sigma.exp() * y)
    found = recognise.recognise(ForwardSDE(derived), torch.tensor(0.0), torch.randn(16, D), differentiable=True)
    leaves = found.affine_leaves()
    assert leaves[0].grad_fn is not None and leaves[2].grad_fn is not None          # NegBackward / ExpBackward
    leaves[0].sum().backward()
    assert torch.equal(derived.mu.grad, -torch.ones(D))
    folded = problems.make("gbm_strat", d=D)
    found = recognise.recognise(ForwardSDE(folded), torch.tensor(0.0), torch.randn(16, D), differentiable=True)
    assert found.affine_leaves() is None                     # mu - 0.5 * sigma^2 was assembled inside the interpretation
    nonlinear = _M(lambda s, t, y: y, lambda s, t, y: torch.exp(-y))
    found = recognise.recognise(ForwardSDE(nonlinear), torch.tensor(0.0), torch.randn(16, D), differentiable=True)
    assert found.affine_leaves() is None


def _random_expression(rng, consts):
    """A random chain of the operators the interpretation follows, as a Python function of y: affine steps, optionally one
    function, more affine steps -- with per-channel tensors, (1, d) tensors, 0-d tensors and numbers as operands, on
    either side. Returns (callable, description)."""
    steps = []

    def operand():
        kind = rng.integers(0, 4)
        if kind == 0:
            return float(rng.uniform(-1.5, 1.5))
        return consts[int(rng.integers(0, len(consts)))]

    def affine_step():
        op = int(rng.integers(0, 8))
        c = operand()
        return {0: (lambda v: v * c, "v*c"), 1: (lambda v: c * v, "c*v"), 2: (lambda v: v + c, "v+c"),
                3: (lambda v: c - v, "c-v"), 4: (lambda v: v - c, "v-c"), 5: (lambda v: -v, "-v"),
                6: (lambda v: v / (2.0 + abs(c) if isinstance(c, float) else 2.0 + c.abs()), "v/c"),
                7: (lambda v: torch.add(v, c, alpha=0.5), "add(alpha)")}[op]
    for _ in range(int(rng.integers(0, 4))):
        steps.append(affine_step())
    if rng.random() < 0.7:
        name = ["exp", "sigmoid", "tanh", "softplus", "sin", "cos"][int(rng.integers(0, 6))]
        fn = {"exp": lambda v: torch.exp(0.3 * v), "softplus": F.softplus}.get(name, getattr(torch, name, None))
        steps.append((fn, name))
        for _ in range(int(rng.integers(0, 3))):
            op = int(rng.integers(0, 4))
            c = operand()
            steps.append({0: (lambda v: v * c, "v*c"), 1: (lambda v: v + c, "v+c"), 2: (lambda v: -v, "-v"),
                          3: (lambda v: c - v, "c-v")}[op])

    def run(y):
        v = y
        for fn, _ in steps:
            v = fn(v)
        return v
    return run, " ; ".join(d for _, d in steps)


def test_random_expression_chains_are_followed_exactly():
    """300 random chains: whatever the interpretation accepts evaluates (scale * phi(rate * y + shift) + offset) to what
    the chain computes; none of these chains may be refused."""
    import numpy as np
    rng = np.random.default_rng(7)
    gen = torch.Generator().manual_seed(7)
    consts = [torch.randn(D, generator=gen), torch.rand(1, D, generator=gen) + 0.5, torch.tensor(0.7),
              nn.Parameter(torch.randn(D, generator=gen)), torch.rand(1, generator=gen) + 0.2]
    y = 0.5 * torch.randn(16, D, generator=gen)
    for i in range(300):
        f, what_f = _random_expression(rng, consts)
        g, what_g = _random_expression(rng, consts)
        sde = _M(lambda s, t, v, f=f: f(v), lambda s, t, v, g=g: g(v))
        found = recognise.recognise(ForwardSDE(sde), torch.tensor(0.1), y)
        for form, fn, what in ((found.f, f, what_f), (found.g, g, what_g)):
            want = fn(y)
            torch.testing.assert_close(_value(form, y), want.detach(), rtol=2e-5, atol=2e-5, msg=f"chain {i}: {what}")
        spec = found.spec()          # ... and the kernels' coefficient vectors say the same
        if spec[0] == "affine_diagonal":
            torch.testing.assert_close(spec[1] * y + spec[2], f(y).detach(), rtol=2e-5, atol=2e-5)
            torch.testing.assert_close(spec[3] * y + spec[4], g(y).detach(), rtol=2e-5, atol=2e-5)
        else:
            names = {v: k for k, v in __import__("torchsde_amd")._native.FN_CODES.items()}
            for fn, code, c4 in ((f, spec[1], spec[3:7]), (g, spec[2], spec[7:11])):
                got = c4[0] * PHI[names[code]](c4[1] * y + c4[2]) + c4[3]
                torch.testing.assert_close(got, fn(y).detach(), rtol=2e-5, atol=2e-5)


class _Scheduled(nn.Module):
    """Coefficients that depend on t through a schedule (a variance-preserving diffusion's forward SDE, Hull-White-style
    mean reversion ...): t only ever broadcasts."""
    noise_type, sde_type = "diagonal", "ito"

    def __init__(self):
        super().__init__()
        self.b0 = nn.Parameter(torch.tensor(0.1))
        self.b1 = nn.Parameter(torch.tensor(2.0))
        self.w = nn.Parameter(torch.rand(D) + 0.5)

    def beta(self, t):
        return self.b0 + t * (self.b1 - self.b0)

    def f(self, t, y):
        return -0.5 * self.beta(t) * y + torch.sin(t) * self.w

    def g(self, t, y):
        return torch.sqrt(self.beta(t)) * 0.3 * torch.sigmoid(y * torch.exp(-t))


def test_time_dependent_coefficients_come_back_as_one_row_per_step():
    sde = _Scheduled()
    y = torch.randn(16, D)
    with pytest.raises(recognise.DependsOnTime):
        recognise.recognise(ForwardSDE(sde), torch.tensor(0.3), y)
    times = torch.linspace(0, 1, 11)[:-1]
    found = recognise.recognise(ForwardSDE(sde), torch.tensor(0.0), y, times=times)
    assert found.timed and found.affine_leaves() is None
    kind, fk, gk, *coefs = found.spec()
    assert kind == "elementwise_diagonal" and all(c.shape == (10, D) and c.is_contiguous() for c in coefs)
    names = {v: k for k, v in __import__("torchsde_amd")._native.FN_CODES.items()}
    for k in (0, 4, 9):
        f = coefs[0][k] * PHI[names[fk]](coefs[1][k] * y + coefs[2][k]) + coefs[3][k]
        g = coefs[4][k] * PHI[names[gk]](coefs[5][k] * y + coefs[6][k]) + coefs[7][k]
        torch.testing.assert_close(f, sde.f(times[k], y), rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(g, sde.g(times[k], y), rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name,f", [
    ("branch on t", lambda s, t, y: y if t > 0.5 else -y),
    ("t on the host", lambda s, t, y: float(t) * y),
    ("t reshaped", lambda s, t, y: t.reshape(-1)[:1] * y),
    ("t concatenated with the state", lambda s, t, y: torch.cat([t.expand(y.shape[0], 1), y], 1)[:, 1:]),
    ("reduction over t", lambda s, t, y: t.sum() * y),
    ("t squeezed", lambda s, t, y: t.squeeze() * y),
])
def test_other_uses_of_t_end_the_interpretation(name, f):
    sde = _M(f, lambda s, t, y: y)
    times = torch.linspace(0, 1, 11)[:-1]
    with pytest.raises(recognise.NotElementwise):
        try:
            recognise.recognise(ForwardSDE(sde), torch.tensor(0.3), torch.randn(16, D))
        except recognise.DependsOnTime:
            recognise.recognise(ForwardSDE(sde), torch.tensor(0.3), torch.randn(16, D), times=times)


POLYNOMIALS = {
    "double well": (lambda s, t, y: y - y ** 3, lambda s, t, y: s.sigma * torch.ones_like(y)),
    "logistic growth": (lambda s, t, y: s.sigma * y * (1.0 - y / 2.5), lambda s, t, y: 0.2 * y),
    "quadratic diffusion": (lambda s, t, y: -y, lambda s, t, y: s.sigma * y * y + 0.1),
    "square of an affine": (lambda s, t, y: (s.mu * y + s.b) ** 2, lambda s, t, y: torch.square(y) - y),
    "cubic by products": (lambda s, t, y: (y - 1.0) * (y + s.b) * (0.5 * y - s.mu), lambda s, t, y: -(y * y) * s.w),
    "sum of polynomials": (lambda s, t, y: (y ** 2 - s.b) + (y ** 3) * 0.1 - 2 * (y * y), lambda s, t, y: 1.0 - y ** 2),
}


@pytest.mark.parametrize("name", sorted(POLYNOMIALS))
def test_polynomials_of_the_state_up_to_degree_three(name):
    """Sums and products of affine functions of the state (double-well and logistic drifts, quadratic diffusions) come
    back as cubics (TSDE_FN_POLY3: the expression kernel with ((c3 y + c2) y + c1) y + c0 in place of phi)."""
    f, g = POLYNOMIALS[name]
    sde = _M(f, g)
    y, t = 0.7 * torch.randn(16, D), torch.tensor(0.3)
    found = recognise.recognise(ForwardSDE(sde), t, y)
    torch.testing.assert_close(_value(found.f, y), sde.f(t, y).detach(), rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(_value(found.g, y), sde.g(t, y).detach(), rtol=2e-5, atol=2e-5)
    kind, fk, gk, *coefs = found.spec()
    assert kind == "elementwise_diagonal" and 7 in (fk, gk) and not found.exact and found.affine_leaves() is None
    for code, c4, fn in ((fk, coefs[:4], f), (gk, coefs[4:], g)):
        if code == 7:
            got = ((c4[0] * y + c4[1]) * y + c4[2]) * y + c4[3]
            torch.testing.assert_close(got, fn(sde, t, y).detach(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("f", [lambda s, t, y: y ** 4, lambda s, t, y: (y * y) * (y * y), lambda s, t, y: torch.exp(y * y),
                               lambda s, t, y: torch.tanh(y) * y, lambda s, t, y: (y ** 2) * torch.sin(y)])
def test_higher_degrees_and_functions_of_polynomials_are_refused(f):
    with pytest.raises(recognise.NotElementwise):
        recognise.recognise(ForwardSDE(_M(f, lambda s, t, y: y)), torch.tensor(0.3), torch.randn(16, D))


# ---- drift AND diffusion networks of (t, y): the reference's Neural* problems ------------------------------------------
def _evaluate(net, t, y):
    """What tsde_trajectory_mlp_general computes for one `kernels.NeuralNet` (include/torchsde_amd.h: tsde_mlp_t)."""
    w1, w1t, b1, w2, b2 = net.tensors
    pre = y @ w1 + b1 + (0.0 if w1t is None else w1t * t)
    hid = F.softplus(pre) if net.activation == 1 else torch.tanh(pre)
    z = hid @ w2 + b2
    return net.scale * (torch.sigmoid(z) if net.final == 1 else z)


@pytest.mark.parametrize("name,noise,d,m", [("general_ito", "general", 8, 4), ("general_strat", "general", 8, 16),
                                            ("netdiag_ito", "diagonal", 8, 8), ("netscalar_ito", "scalar", 8, 1)])
def test_drift_and_diffusion_networks_with_time_as_an_input(name, noise, d, m):
    """`torch.cat([t.expand(B, 1), y], 1)` into `nn.Sequential(Linear, Softplus, Linear[, Sigmoid])`, a numeric factor, the
    reshape to (B, d, m): followed, and the spec evaluates to what the module computes (tests/problems.py:135-252)."""
    from torchsde_amd import _native
    if not _native.is_built():
        pytest.skip("needs the built library (shape limits come from the C ABI)")
    sde = problems.make(name, d=d, m=m) if noise == "general" else problems.make(name, d=d)
    y, t = 0.5 * torch.randn(16, d), torch.tensor(0.37)
    found = recognise.recognise(ForwardSDE(sde), t, y)
    assert found.neural and not found.perceptron and not found.timed and found.affine_leaves() is None
    kind, fnet, gnet, code, m_found = found.neural_spec(noise)
    assert kind == "neural" and m_found == m and code == {"diagonal": 0, "scalar": 1, "general": 2}[noise]
    assert fnet.tensors[1] is not None and gnet.tensors[1] is not None            # both nets see t
    with torch.no_grad():
        torch.testing.assert_close(_evaluate(fnet, t, y), sde.f(t, y), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(_evaluate(gnet, t, y).reshape(sde.g(t, y).shape), sde.g(t, y), rtol=1e-5, atol=1e-6)
    # two probe heights, same networks (the trust check's comparison)
    again = recognise.recognise(ForwardSDE(sde), t, y, rows=5).neural_spec(noise)
    assert again[1] == fnet and again[2] == gnet and again[3:] == (code, m)
    with pytest.raises(recognise.NotElementwise):
        found.spec()


def test_networks_the_neural_kernel_does_not_take():
    from torchsde_amd import _native
    if not _native.is_built():
        pytest.skip("needs the built library")
    y, t = torch.randn(16, 8), torch.tensor(0.1)
    odd = problems.make("general_odd_ito")                     # d = 3, m = 5: runs in the m = 8 tile width, element-wise rows
    assert recognise.recognise(ForwardSDE(odd), t, torch.randn(16, 3)).neural_spec("general")[3:] == (_native.NOISE_GENERAL, 5)
    wide = problems.MLPGeneral(68, 4, "ito", hidden=8)         # more than 64 state channels / 33 Brownian channels: no kernel
    with pytest.raises(recognise.NotElementwise, match="outside the neural-SDE kernel's shapes"):
        recognise.recognise(ForwardSDE(wide), t, torch.randn(16, 68)).neural_spec("general")
    many = problems.MLPGeneral(8, 33, "ito", hidden=8)
    with pytest.raises(recognise.NotElementwise, match="outside the neural-SDE kernel's shapes"):
        recognise.recognise(ForwardSDE(many), t, y).neural_spec("general")

    class TimeTwice(problems.MLPNetDiag):                      # arithmetic on t before the cat: not "the time" any more
        def _ty(self, t, y):
            return torch.cat([(2 * t).expand(y.size(0), 1), y], dim=1)
    with pytest.raises(recognise.NotElementwise, match="cat of t"):
        recognise.recognise(ForwardSDE(TimeTwice(8)), t, y)

    class Wrong(problems.MLPNetDiag):                          # [y, t] instead of [t, y]
        def _ty(self, t, y):
            return torch.cat([y, t.expand(y.size(0), 1)], dim=1)
    with pytest.raises(recognise.NotElementwise, match="cat of t"):
        recognise.recognise(ForwardSDE(Wrong(8)), t, y)

    class Squared(problems.MLPNetDiag):                        # something after the net that is not a factor or a reshape
        def g(self, t, y):
            return self.g_net(self._ty(t, y)) ** 2
    with pytest.raises(recognise.NotElementwise, match="output of the drift network"):
        recognise.recognise(ForwardSDE(Squared(8)), t, y)


# ---- expression programs: any elementwise code --------------------------------------------------------------------------
def _run_program(words, consts, y, t=None):
    """The stack machine of csrc/trajectory.hip (ProgModel::run) in torch: what the kernel evaluates per element."""
    stack = []
    unary = {16: torch.neg, 17: torch.exp, 18: torch.log, 19: torch.sin, 20: torch.cos, 21: torch.tanh, 22: torch.sigmoid,
             23: F.softplus, 24: torch.sqrt, 25: torch.abs, 26: torch.relu, 27: torch.reciprocal, 28: lambda v: v * v,
             29: lambda v: (v * v) * v}
    for w in words:
        op, src, k = w & 0xFF, (w >> 8) & 0xFF, w >> 16
        if op < 16:
            if src == 0:
                b, a = stack.pop(), stack.pop()
            else:
                b = consts[k].expand_as(y) if src == 1 else (y if src == 2 else t.expand_as(y))
                a = stack.pop() if op != 0 else None
            if op == 0:
                stack.append(b)
            else:
                stack.append({1: a + b, 2: a - b, 3: b - a, 4: a * b, 5: a / b, 6: b / a}[op])
        elif op == 30:
            stack.append(stack[-1])
        else:
            stack.append(unary[op](stack.pop()))
        assert len(stack) <= 4
    assert len(stack) == 1
    return stack[0]


PROGRAMS = {
    # the SDE of the reference's ExScalar test problem (tests/problems.py:75-103): the same formulas as user code
    "ex_scalar": ("scalar", lambda s, t, y: -s.mu ** 2. * torch.sin(y) * torch.cos(y) ** 3.,
                  lambda s, t, y: (s.sigma * torch.cos(y) ** 2).unsqueeze(dim=-1)),
    "two functions summed": ("diagonal", lambda s, t, y: torch.tanh(y) + y, lambda s, t, y: 0.3 * torch.sigmoid(y) + s.sigma),
    "quartic well": ("diagonal", lambda s, t, y: y - y ** 4 * s.mu, lambda s, t, y: s.sigma * torch.sqrt(1.0 + y * y)),
    "rational": ("diagonal", lambda s, t, y: -y / (1.0 + y ** 2), lambda s, t, y: s.sigma / (2.0 + torch.cos(y))),
    "deep tree": ("diagonal", lambda s, t, y: (torch.sin(y) * s.mu + torch.cos(y)) * (torch.exp(-y * y) + s.b * y),
                  lambda s, t, y: (y * s.sigma + 0.1) * (torch.tanh(y) - 2.0)),
    "softplus and abs": ("diagonal", lambda s, t, y: F.softplus(y) - torch.abs(y) * s.mu, lambda s, t, y: torch.relu(y) + 0.2),
    # single ATen operators that are compositions of the machine's functions
    "silu, mish, rsqrt, clamp": ("diagonal", lambda s, t, y: F.silu(y) - F.mish(y * s.mu) + torch.clamp(y, min=0),
                                 lambda s, t, y: s.sigma * torch.rsqrt(1.0 + y * y) + (2.0 + y * y) ** -2 + (1.5 + torch.sin(y)) ** -0.5),
    # t as one more leaf (the reference's ExAdditive drift, tests/problems.py:119-121, beside a state-dependent diffusion)
    "time in the arithmetic": ("diagonal", lambda s, t, y: s.b / torch.sqrt(1. + t) - y / (2. + 2. * t) + torch.tanh(y) * torch.cos(t),
                               lambda s, t, y: (s.sigma / torch.sqrt(1. + t)).expand_as(y) * torch.sigmoid(y)),
}


@pytest.mark.parametrize("name", sorted(PROGRAMS))
def test_expression_programs_evaluate_to_the_users_code(name):
    """Code the single-function forms refuse comes back as postfix programs over a four-deep stack; running them (and the
    derivative program Milstein uses) reproduces f, g and dg/dy."""
    noise, f, g = PROGRAMS[name]
    sde = _M(f, g)
    sde.noise_type = noise
    y, t = 0.8 * torch.randn(16, D), torch.tensor(0.3)
    with pytest.raises(recognise.NotElementwise):
        recognise.recognise(ForwardSDE(sde), t, y).spec()
    found = recognise.recognise_program(ForwardSDE(sde), t, y, noise)
    kind, fw, gw, dgw, table, scalar = found.spec(milstein=name not in ("softplus and abs",))
    assert kind == "program_diagonal" and scalar == (noise == "scalar") and table.shape[1] == D
    with torch.no_grad():
        torch.testing.assert_close(_run_program(fw, table, y, t), sde.f(t, y), rtol=1e-5, atol=1e-6)
        want_g = sde.g(t, y)
        torch.testing.assert_close(_run_program(gw, table, y, t), want_g.reshape(16, D), rtol=1e-5, atol=1e-6)
    if dgw:
        yy = y.clone().requires_grad_(True)
        dg, = torch.autograd.grad(sde.g(t, yy).sum(), yy)            # elementwise g: the gradient of the sum is g'
        torch.testing.assert_close(_run_program(dgw, table, y, t), dg, rtol=1e-4, atol=1e-5)
    again = recognise.recognise_program(ForwardSDE(sde), t, y, noise, rows=5)
    assert again.structure() == found.structure()


def test_programs_refuse_what_is_not_elementwise_arithmetic():
    y, t = torch.randn(16, D), torch.tensor(0.3)
    for f, reason in ((lambda s, t, y: y @ torch.eye(D), "aten::mm"), (lambda s, t, y: y * float(t), "aten::_local_scalar_dense"),
                      (lambda s, t, y: y - y.mean(0), "aten::mean"), (lambda s, t, y: y ** 2.5, "power"),
                      (lambda s, t, y: y + torch.randn(D), "random"), (lambda s, t, y: y * s.b.add_(1.0), "existed before")):
        with pytest.raises(recognise.NotElementwise, match=reason):
            recognise.recognise_program(ForwardSDE(_M(f, lambda s, t, y: y)), t, y, "diagonal")
    # five values alive at once: beyond the four-deep stack
    wide = lambda s, t, y: ((torch.sin(y) * torch.cos(y)) + (torch.exp(y) * torch.tanh(y))) * \
        ((torch.sin(2 * y) * torch.cos(3 * y)) + (torch.exp(-y) * torch.tanh(2 * y)))          # noqa: E731
    found = recognise.recognise_program(ForwardSDE(_M(wide, lambda s, t, y: y)), t, y, "diagonal")      # depth 4: fits
    assert found.programs[0]
    wider = lambda s, t, y: wide(s, t, y) * (wide(s, t, 2 * y) + wide(s, t, 3 * y) * wide(s, t, 4 * y))       # noqa: E731
    with pytest.raises(recognise.NotElementwise, match="more than four"):
        recognise.recognise_program(ForwardSDE(_M(wider, lambda s, t, y: y)), t, y, "diagonal")


# ---- additive noise: the drift a program, the diffusion a table over the stage times ---------------------------------------
class _NetOfTimeDiffusion(nn.Module):
    """g of the reference's NeuralAdditive (tests/problems.py:208-220)."""
    noise_type, sde_type = "additive", "ito"

    def __init__(self, d, m):
        super().__init__()
        self.d, self.m = d, m
        self.g_net = nn.Sequential(nn.Linear(1, 8), nn.Softplus(), nn.Linear(8, d * m), nn.Sigmoid())

    def f(self, t, y):
        return -torch.sin(y) * t

    def g(self, t, y):
        return self.g_net(t.expand(y.size(0), 1)).view(y.size(0), self.d, self.m)


@pytest.mark.parametrize("make,m,timed", [(lambda: problems.AdditiveDecay(D, 3), 3, True),
                                          (lambda: problems.AdditiveShared(D, 4), 4, False),
                                          (lambda: _NetOfTimeDiffusion(D, 5), 5, True)])
def test_additive_noise_drift_program_and_diffusion_table(make, m, timed):
    """The reference's ExAdditive, a constant matrix `sigma.expand(B, d, m)`, a network of t: the table holds g(t)^T for every
    stage time -- one batched call of the user's g -- and equals g called time by time; the drift program evaluates to f."""
    sde = ForwardSDE(make())
    y, t = 0.8 * torch.randn(16, D), torch.tensor(0.3)
    times = torch.linspace(0.0, 1.0, 13)
    found = recognise.recognise_additive(sde, t, y, times, check_rows=True)
    kind, fw, consts, table, m_found = found.spec()
    assert kind == "program_additive" and m_found == m and found.time_dependent == timed
    with torch.no_grad():
        if timed:
            want = torch.stack([sde.g(tt, y[:1])[0].t() for tt in times])
            assert table.shape == (13, m, D)
        else:
            want = sde.g(t, y[:1])[0].t()
            assert table.shape == (m, D)
        torch.testing.assert_close(table, want, rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(_run_program(fw, consts, y, t), sde.f(t, y), rtol=1e-5, atol=1e-6)
    again = recognise.recognise_additive(sde, t, y, times, rows=5, check_rows=True)
    assert again.structure() == found.structure()
    torch.testing.assert_close(again.table, table, rtol=1e-6, atol=1e-7)


def test_additive_noise_refusals():
    y, t, times = torch.randn(16, D), torch.tensor(0.3), torch.linspace(0.0, 1.0, 5)

    class FromState(problems.AdditiveShared):
        def g(self, t, y):
            return (self.sigma * y.mean()).expand(y.size(0), -1, -1)

    class HostTime(problems.AdditiveShared):
        def g(self, t, y):
            return self.sigma.expand(y.size(0), -1, -1) * (1.0 if t > 0.5 else 2.0)

    class PerRow(problems.AdditiveShared):
        def g(self, t, y):
            return self.sigma.expand(y.size(0), -1, -1) * torch.arange(y.size(0), dtype=y.dtype).reshape(-1, 1, 1)

    class Random(problems.AdditiveShared):
        def g(self, t, y):
            return self.sigma.expand(y.size(0), -1, -1) + torch.randn(D, 4)

    class MatrixDrift(problems.AdditiveShared):
        def f(self, t, y):
            return y @ torch.eye(D)

    for cls, m, reason in ((FromState, 4, "aten::mean"), (HostTime, 4, "reads t on the host"), (PerRow, 4, "differs between batch rows"),
                           (Random, 4, "random"), (MatrixDrift, 4, "aten::mm"), (problems.AdditiveShared, 32, "up to 16")):
        with pytest.raises(recognise.NotElementwise, match=reason):
            recognise.recognise_additive(ForwardSDE(cls(D, m)), t, y, times, check_rows=True)


def test_additive_noise_with_a_network_drift():
    """The reference's NeuralAdditive (tests/problems.py:195-224): the drift is followed as a perceptron of cat([t, y]) (time
    column split off), the diffusion tabulated."""
    sde = ForwardSDE(problems.MLPNetAdditive(8, 3, hidden=16))
    y, t, times = torch.randn(16, 8), torch.tensor(0.3), torch.linspace(0.0, 1.0, 7)
    found = recognise.recognise_additive(sde, t, y, times, check_rows=True)
    kind, net, _, table, m = found.spec()
    assert kind == "neural_additive" and m == 3 and table.shape == (7, 3, 8) and found.structure()[0][0] == "perceptron"
    w1, w1t, b1, w2, b2 = net.tensors
    assert w1.shape == (8, 16) and w1t.shape == (16,) and w2.shape == (16, 8)
    with torch.no_grad():
        hidden = torch.nn.functional.softplus(y @ w1 + w1t * t + b1)
        torch.testing.assert_close(hidden @ w2 + b2, sde.f(t, y), rtol=1e-5, atol=1e-6)
    again = recognise.recognise_additive(sde, t, y, times, rows=5, check_rows=True)
    assert again.structure() == found.structure() and again.spec()[1] == net


# ---- random expression trees through the program compiler -------------------------------------------------------------
def _random_tree(rng, depth):
    """(python source of an expression of y, t, s.mu, s.sigma, s.b and numbers): smooth on y in [-1, 1], denominators and
    arguments of log / sqrt kept positive."""
    leaves = ["y", "y", "y", "s.mu", "s.sigma", "s.b", "t", "torch.tensor(0.5)", "torch.tensor(2.0)", "torch.tensor(-1.5)"]
    if depth == 0 or rng.random() < 0.15:
        return rng.choice(leaves)
    kind = rng.random()
    if kind < 0.35:
        fn = rng.choice(["torch.sin", "torch.cos", "torch.tanh", "torch.sigmoid", "torch.exp", "F.softplus", "torch.square",
                         "torch.neg"])
        inner = _random_tree(rng, depth - 1)
        if fn == "torch.exp":
            inner = f"torch.tanh({inner})"                # (bounded argument)
        return f"{fn}({inner})"
    if kind < 0.45:
        inner = _random_tree(rng, depth - 1)
        return rng.choice([f"torch.sqrt(1.5 + torch.tanh({inner}))", f"torch.log(2.5 + torch.sin({inner}))",
                           f"({inner}) ** 2", f"({inner}) ** 3", f"torch.reciprocal(2.0 + torch.cos({inner}))"])
    a, b = _random_tree(rng, depth - 1), _random_tree(rng, depth - 1)
    op = rng.choice(["+", "-", "*", "/"])
    if op == "/":
        b = f"(2.5 + torch.sin({b}))"
    return f"({a} {op} {b})"


@pytest.mark.parametrize("seed", range(120))
def test_random_expression_trees_compile_to_programs_that_evaluate_to_the_code(seed):
    """120 seeded random expression trees (depth <= 5; unary functions, powers, the four operations with operands in either
    order, constants, parameters, t) as drift and diffusion of a user module: whatever the interpreter accepts must come back
    as programs whose evaluation on the four-deep stack (the kernel's machine, restated in `_run_program`) equals the user's
    code, and whose derivative program equals autograd's dg/dy. Trees that need more than four live values or 96 words are
    refused with that reason -- never evaluated wrongly."""
    import random
    rng = random.Random(seed)
    src_f, src_g = _random_tree(rng, 5 if seed % 3 == 0 else 4), _random_tree(rng, 4 if seed % 3 == 0 else 3)
    if "y" not in src_f:
        src_f = f"({src_f}) * y"
    if "y" not in src_g:
        src_g = f"({src_g}) + torch.sin(y)"
    env = {"torch": torch, "F": F}
    f = eval(f"lambda s, t, y: {src_f}", env)
    g = eval(f"lambda s, t, y: {src_g}", env)
    sde = _M(f, g)
    y, t = 2.0 * torch.rand(16, D) - 1.0, torch.tensor(0.3)
    try:
        found = recognise.recognise_program(ForwardSDE(sde), t, y, "diagonal")
    except recognise.NotElementwise as e:
        assert any(reason in str(e) for reason in ("more than four", "more than 96", "more than 64")), (src_f, src_g, str(e))
        return
    fw, gw, dgw = found.programs
    table = found.const_table()
    with torch.no_grad():
        torch.testing.assert_close(_run_program(fw, table, y, t), sde.f(t, y).expand_as(y), rtol=2e-5, atol=2e-6, msg=src_f)
        torch.testing.assert_close(_run_program(gw, table, y, t), sde.g(t, y).expand_as(y), rtol=2e-5, atol=2e-6, msg=src_g)
    if dgw:
        yy = y.clone().requires_grad_(True)
        dg, = torch.autograd.grad(sde.g(t, yy).expand_as(yy).sum(), yy)
        torch.testing.assert_close(_run_program(dgw, table, y, t), dg, rtol=2e-4, atol=2e-5, msg=src_g)


def test_additive_noise_table_size_is_bounded():
    """A time-dependent diffusion is tabulated over EVERY stage time of the solve: a solve of millions of steps keeps the
    stepwise route instead of allocating gigabytes."""
    sde = ForwardSDE(problems.AdditiveDecay(64, 16))
    with pytest.raises(recognise.NotElementwise, match="MiB"):
        recognise.recognise_additive(sde, torch.tensor(0.0), torch.randn(16, 64), torch.linspace(0.0, 1.0, 2 ** 17))
    found = recognise.recognise_additive(ForwardSDE(problems.AdditiveShared(64, 16)), torch.tensor(0.0), torch.randn(16, 64),
                                         torch.linspace(0.0, 1.0, 2 ** 17))
    assert found.table.shape == (16, 64)                    # (a constant matrix needs no table over time)


# ---- random single-function forms through the coefficient algebra -----------------------------------------------------
def _random_affine(rng, inner):
    for _ in range(rng.randint(0, 3)):
        c = rng.choice(["s.mu", "s.sigma", "s.b", "2.0", "-0.5", "s.w"])
        inner = rng.choice([f"({inner} * {c})", f"({c} * {inner})", f"({inner} + {c})", f"({c} - {inner})", f"({inner} - {c})",
                            f"(-{inner})", f"({inner} / (1.5 + s.sigma))", f"torch.add({c}, {inner}, alpha=2)"])
    return inner


@pytest.mark.parametrize("seed", range(60))
def test_random_single_function_forms_fold_to_the_right_coefficients(seed):
    """Sixty seeded compositions of per-channel affine operations around at most one function of the state (or products of
    affine values up to a cubic): the five coefficients `recognise` folds them into evaluate to what the code computes.
    Whatever leaves the form is refused, never folded wrongly."""
    import random
    rng = random.Random(1000 + seed)
    def one():
        kind = rng.random()
        inner = _random_affine(rng, "y")
        if kind < 0.5:
            fn = rng.choice(["torch.exp", "torch.sigmoid", "torch.tanh", "F.softplus", "torch.sin", "torch.cos"])
            return _random_affine(rng, f"{fn}({inner})")
        if kind < 0.8:
            return rng.choice([f"({inner}) * ({_random_affine(rng, 'y')})", f"({inner}) ** 2", f"({inner}) ** 3",
                               f"({inner}) * ({_random_affine(rng, 'y')}) + ({_random_affine(rng, 'y')})"])
        return inner
    src_f, src_g = one(), one()
    env = {"torch": torch, "F": F}
    sde = _M(eval(f"lambda s, t, y: {src_f}", env), eval(f"lambda s, t, y: {src_g}", env))
    y, t = 0.6 * torch.randn(16, D), torch.tensor(0.3)
    try:
        found = recognise.recognise(ForwardSDE(sde), t, y)
    except recognise.NotElementwise:
        return
    with torch.no_grad():
        for got, want, src in ((_value(found.f, y), sde.f(t, y).expand_as(y), src_f),
                               (_value(found.g, y), sde.g(t, y).expand_as(y), src_g)):
            # (an expanded cubic cancels near its roots: the error scales with the size of its terms, not of its value)
            torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-6 * max(1.0, want.abs().max().item()), msg=src)


# ---- stop-gradients ---------------------------------------------------------------------------------------
class _StopGrad(nn.Module):
    noise_type, sde_type = "diagonal", "ito"

    def __init__(self, kind):
        super().__init__()
        self.kind = kind
        self.mu = nn.Parameter(torch.full((4,), 0.3))
        self.sigma = nn.Parameter(torch.full((4,), 0.2))

    def f(self, t, y):
        if self.kind == "detach":
            return y.detach() * self.mu
        if self.kind == "no_grad":
            with torch.no_grad():
                z = y * 2
            return z * self.mu
        if self.kind == "mixed":
            return y.detach() * self.mu + y * self.sigma
        if self.kind == "function_detached":
            return torch.tanh(y).detach() * self.mu + y
        if self.kind == "parameter_detached":
            return y * self.mu.detach()
        if self.kind == "clamp":
            return torch.clamp(y, min=0) * self.mu
        return y * self.mu

    def g(self, t, y):
        if self.kind == "data":
            return self.sigma * y.data
        return self.sigma * y


@pytest.mark.parametrize("kind", ["detach", "no_grad", "data", "mixed", "function_detached"])
def test_a_stop_gradient_on_the_state_ends_the_differentiable_interpretation(kind):
    """The reference differentiates the user's code as written (autograd through base_solver.py:143-149): `y.detach()`,
    `y.data` and state arithmetic under `torch.no_grad()` cut the gradient there. The kernels' sensitivities would run straight
    through, so with `differentiable=True` both interpreters refuse; without gradients a stop-gradient is the identity."""
    from torchsde_amd import recognise
    from torchsde_amd.sde import ForwardSDE
    sde = ForwardSDE(_StopGrad(kind))
    y0, t = torch.full((16, 4), 0.1), torch.tensor(0.0)
    for interpret in (recognise.recognise, lambda *a, **k: recognise.recognise_program(*a, "diagonal", **k)):
        with pytest.raises(recognise.NotElementwise, match="stop-gradient"):
            interpret(sde, t, y0, differentiable=True)
    assert recognise.recognise_program(sde, t, y0, "diagonal") is not None          # values only: followed


def test_what_is_not_a_stop_gradient_on_the_state_is_still_followed():
    from torchsde_amd import recognise
    from torchsde_amd.sde import ForwardSDE
    y0, t = torch.full((16, 4), 0.1), torch.tensor(0.0)
    for kind in ("plain", "parameter_detached"):
        sde = ForwardSDE(_StopGrad(kind))
        found = recognise.recognise(sde, t, y0, differentiable=True)
        leaves = found.affine_leaves()
        assert leaves is not None
        # the detached parameter is a coefficient without a graph: no gradient reaches `mu`, as in the reference
        assert leaves[0].requires_grad == (kind == "plain")
        assert recognise.recognise_program(sde, t, y0, "diagonal", differentiable=True) is not None
    # constants made from the state's shape carry no gradient and need none
    class Ones(_StopGrad):
        def g(self, t, y):
            return torch.ones_like(y) * self.sigma
    assert recognise.recognise(ForwardSDE(Ones("plain")), t, y0, differentiable=True) is not None


def test_clamp_is_refused_when_gradients_flow():
    """torch's clamp passes the gradient at the boundary (x >= min), the machine's relu does not."""
    from torchsde_amd import recognise
    from torchsde_amd.sde import ForwardSDE
    sde = ForwardSDE(_StopGrad("clamp"))
    y0, t = torch.zeros(16, 4), torch.tensor(0.0)
    assert recognise.recognise_program(sde, t, y0, "diagonal") is not None
    with pytest.raises(recognise.NotElementwise, match="clamp"):
        recognise.recognise_program(sde, t, y0, "diagonal", differentiable=True)


# ---- deeper networks, LipSwish, closing tanh: the generator of the reference's examples/sde_gan.py -----------------------------
_LipSwish = problems.LipSwish
_sde_gan_mlp = problems.sde_gan_mlp


class _GeneratorFunc(problems.SdeGanGenerator):
    """examples/sde_gan.py:77-101 (workloads.problems.SdeGanGenerator) with the example's argument order."""

    def __init__(self, noise_size, hidden_size, mlp_size, num_layers):
        super().__init__(noise_size, hidden_size, mlp_size, num_layers, seed=int(torch.randint(0, 2 ** 31, (1,))))


@pytest.mark.parametrize("num_layers", [1, 2, 3])
def test_the_sde_gan_generator_is_followed_at_every_depth(num_layers):
    sde = _GeneratorFunc(3, 16, 16, num_layers)
    y = torch.randn(12, 16)
    found = recognise.recognise(ForwardSDE(sde), torch.tensor(0.3), y, differentiable=True)
    assert found.neural
    for net, module in ((found.f, sde._drift), (found.g, sde._diffusion)):
        linears = [m for m in module if isinstance(m, nn.Linear)]
        assert net.act == "silu" and net.act_scale == 0.909 and net.final == "tanh" and net.wt is not None
        assert net.w_full is linears[0].weight and net.b1 is linears[0].bias
        assert [w for w, _ in net.mids] == [m.weight for m in linears[1:-1]] or all(
            a is b.weight for (a, _), b in zip(net.mids, linears[1:-1]))
        assert net.w2 is linears[-1].weight and net.b2 is linears[-1].bias
        assert not net.classic()
    assert found.g.shape == (2, 16, 3)
    with pytest.raises(recognise.NotElementwise, match="deeper than two layers"):
        found.neural_spec("general")             # (the Euler / midpoint kernel does not take it)


def test_hidden_layers_must_share_activation_width_and_factor():
    class Mixed(nn.Module):
        sde_type, noise_type = "stratonovich", "diagonal"

        def __init__(self, net):
            super().__init__()
            self.net = net
            self.g_net = nn.Sequential(nn.Linear(8, 8), nn.Tanh(), nn.Linear(8, 8))

        def f(self, t, y):
            return self.net(y)

        def g(self, t, y):
            return self.g_net(y)
    y = torch.randn(12, 8)
    for net in (nn.Sequential(nn.Linear(8, 8), nn.Tanh(), nn.Linear(8, 8), nn.Softplus(), nn.Linear(8, 8)),
                nn.Sequential(nn.Linear(8, 8), nn.Tanh(), nn.Linear(8, 12), nn.Tanh(), nn.Linear(12, 8)),
                nn.Sequential(nn.Linear(8, 8), _LipSwish(), nn.Linear(8, 8), nn.SiLU(), nn.Linear(8, 8))):
        with pytest.raises(recognise.NotElementwise):
            recognise.recognise(ForwardSDE(Mixed(net)), torch.tensor(0.0), y)
    ok = nn.Sequential(nn.Linear(8, 8), nn.Tanh(), nn.Linear(8, 8), nn.Tanh(), nn.Linear(8, 8), nn.Tanh(), nn.Linear(8, 8))
    found = recognise.recognise(ForwardSDE(Mixed(ok)), torch.tensor(0.0), y)
    assert len(found.f.mids) == 2 and found.f.final is None

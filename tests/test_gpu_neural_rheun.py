"""The reversible Heun pair on the matrix cores (csrc/tsde_neural_rheun.h, torchsde_amd/neural_rheun.py; ``-m gpu``).

Kernel level, through the C ABI: `tsde_rheun_mlp_forward` / `_backward` against a float64 torch statement of
methods/reversible_heun.py:48-73 on the SAME increments (the Brownian motion's own, queried interval by interval), values and --
by autograd through that statement -- the gradients with respect to y0 and every weight: what `sdeint_adjoint(method=
"reversible_heun", adjoint_method="adjoint_reversible_heun")` returns in the reference up to rounding (the pair's backward pass is
exact, reversible_heun.py:76-144)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
DT = 2.0 ** -6


def _linears(gen, sizes, bias=True):
    out = []
    for a, b in zip(sizes[:-1], sizes[1:]):
        w = (torch.randn(b, a, generator=gen) / a ** 0.5).to(DEV).requires_grad_(True)
        out.append((w, (0.1 * torch.randn(b, generator=gen)).to(DEV).requires_grad_(True) if bias else None))
    return out


def _nets(d, m, hidden, depth, act, final_f, final_g, noise, seed, time_input=True):
    from torchsde_amd import _native
    from torchsde_amd.neural_rheun import DeepNet
    gen = torch.Generator().manual_seed(seed)
    out = d * m if noise == "general" else d
    inp = d + (1 if time_input else 0)
    act_code = {"tanh": _native.ACT_TANH, "softplus": _native.ACT_SOFTPLUS, "lipswish": _native.ACT_SILU}[act]
    finals = {None: _native.FINAL_NONE, "sigmoid": _native.FINAL_SIGMOID, "tanh": _native.FINAL_TANH}
    scale = 0.909 if act == "lipswish" else 1.0
    f = DeepNet(_linears(gen, [inp] + [hidden] * (depth - 1) + [d]), act_code, scale, finals[final_f], 1.0, time_input)
    g = DeepNet(_linears(gen, [inp] + [hidden] * (depth - 1) + [out]), act_code, scale, finals[final_g], 0.3, time_input)
    return f, g


def _schedule(bm, steps, outputs, B):
    from torchsde_amd import kernels as K
    grid = np.arange(steps + 1) * DT
    bm.adopt_grid(grid)
    cells = np.asarray(bm.match_grid(grid), dtype=np.int64)
    rows = np.zeros((steps, 8))
    rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3] = DT, DT / 2, 1 / DT, np.sqrt(DT)
    rows[:, 4], rows[:, 5], rows[:, 6], rows[:, 7] = np.sqrt(DT), np.sqrt(DT / 12), DT, grid[:-1]
    return K.TrajectorySchedule(rows, cells, outputs, [(0.0, 1.0)] * len(outputs), torch.device(DEV), torch.float32), grid


def _reference(f, g, y0, bm, grid, outputs, noise, d, m):
    """methods/reversible_heun.py:48-73 in float64 on the generator's increments, recorded by autograd."""
    f64 = lambda net: net.rebuilt([t.double() for t in net.parameters()])                 # noqa: E731
    fd, gd = f64(f), f64(g)

    def fg(t, z):
        tt = torch.tensor(t, dtype=torch.float64, device=DEV)
        gv = gd(tt, z)
        return fd(tt, z), (gv.reshape(z.shape[0], d, m) if noise == "general" else gv)

    def prod(gv, w):
        return torch.bmm(gv, w.unsqueeze(-1)).squeeze(-1) if noise == "general" else gv * w
    y = y0.double()
    z = y
    f0, g0 = fg(grid[0], z)
    ys = [y]
    for k in range(len(grid) - 1):
        dW = bm(float(grid[k]), float(grid[k + 1])).double()
        if noise == "scalar":
            dW = dW.expand(-1, d)
        z1 = 2 * y - z + f0 * DT + prod(g0, dW)
        f1, g1 = fg(grid[k + 1], z1)
        y = y + (f0 + f1) * (0.5 * DT) + prod(g0 + g1, 0.5 * dW)
        z, f0, g0 = z1, f1, g1
        if k + 1 in outputs:
            ys.append(y)
    return torch.stack(ys), z


CASES = [
    # d, m, hidden, depth, act, final_f, final_g, noise, B
    (8, 3, 8, 2, "lipswish", "tanh", "tanh", "general", 40),          # examples/sde_gan.py:50-66,93-94 (num_layers = 1)
    (16, 3, 16, 3, "lipswish", "tanh", "tanh", "general", 33),         # ... num_layers = 2, the example's sizes
    (8, 16, 8, 2, "tanh", None, "sigmoid", "general", 24),             # tests/problems.py:226-252 (NeuralGeneral)
    (20, 7, 24, 4, "softplus", None, None, "general", 17),             # four Linear layers, channel counts between the tiles
    (32, 16, 64, 2, "softplus", None, "sigmoid", "general", 48),       # the BASELINE configs[2] shape
    (8, 8, 8, 2, "tanh", None, "sigmoid", "diagonal", 40),             # NeuralDiagonal
    (12, 12, 20, 3, "softplus", "tanh", None, "diagonal", 19),
    (6, 1, 8, 2, "tanh", None, "sigmoid", "scalar", 21),               # NeuralScalar
    (3, 2, 8, 2, "tanh", None, None, "general", 9),                    # rows that are not 16-byte groups
    # more than 32 state channels: the 64-channel instantiations (run-time row stride of the diffusion's last layer)
    (40, 4, 16, 2, "tanh", None, "sigmoid", "general", 20),
    (36, 12, 16, 3, "softplus", None, "tanh", "general", 18),
    (48, 48, 32, 2, "softplus", None, None, "diagonal", 18),
]


@pytest.mark.parametrize("d,m,hidden,depth,act,final_f,final_g,noise,B", CASES)
def test_forward_and_backward_against_a_float64_statement_of_the_reference(d, m, hidden, depth, act, final_f, final_g, noise, B):
    import torchsde_amd
    from torchsde_amd import _native, neural_rheun
    steps, outputs = 12, [5, 12]
    f, g = _nets(d, m, hidden, depth, act, final_f, final_g, noise, seed=d + m)
    code = {"general": _native.NOISE_GENERAL, "diagonal": _native.NOISE_DIAGONAL, "scalar": _native.NOISE_SCALAR}[noise]
    assert neural_rheun.lds_bytes(d, m, f, g, code) > 0
    bm = torchsde_amd.BrownianInterval(0.0, steps * DT, size=(B, m), dtype=torch.float32, device=DEV, entropy=5, dt=DT)
    schedule, grid = _schedule(bm, steps, outputs, B)
    gen = torch.Generator(device=DEV).manual_seed(1)
    y0 = (0.3 * torch.randn(B, d, device=DEV, generator=gen)).requires_grad_(True)
    weights = torch.randn(3, B, d, device=DEV, generator=gen)
    ys = neural_rheun.solve(y0, f, g, code, m, schedule, grid.astype(np.float32), bm)
    want, _ = _reference(f, g, y0, bm, grid, outputs, noise, d, m)
    scale = want.detach().abs().max().item()
    assert (ys.detach().double() - want.detach()).abs().max().item() <= 2e-5 * scale + 1e-6
    params = [y0] + f.parameters() + g.parameters()
    got = torch.autograd.grad((ys * weights).sum(), params)
    ref = torch.autograd.grad((want * weights.double()).sum(), params)
    for i, (a, b) in enumerate(zip(got, ref)):
        s = b.abs().max().item() + 1e-12
        err = (a.double() - b).abs().max().item()
        assert err <= 2e-4 * s + 1e-7, (i, tuple(a.shape), err, s)


def test_chunked_backward_sweep_equals_the_single_launch(monkeypatch):
    """The stash budget cuts the sweep into launches of a few evaluations each: same gradients, bit for bit in the state."""
    import torchsde_amd
    from torchsde_amd import _native, neural_rheun
    d, m, B, steps = 8, 3, 40, 16
    f, g = _nets(d, m, 8, 3, "lipswish", "tanh", "tanh", "general", seed=3)
    bm = torchsde_amd.BrownianInterval(0.0, steps * DT, size=(B, m), dtype=torch.float32, device=DEV, entropy=9, dt=DT)
    schedule, grid = _schedule(bm, steps, [4, 9, 16], B)
    y0 = torch.full((B, d), 0.2, device=DEV, requires_grad=True)
    params = [y0] + f.parameters() + g.parameters()
    results = []
    for budget in (neural_rheun.STASH_BYTES, 3 * B * neural_rheun._Stash.floats_per_row(d, m, f, g, True) * 4):
        monkeypatch.setattr(neural_rheun, "STASH_BYTES", budget)
        ys = neural_rheun.solve(y0, f, g, _native.NOISE_GENERAL, m, schedule, grid.astype(np.float32), bm)
        results.append(torch.autograd.grad((ys ** 2).sum(), params))
    assert torch.equal(results[0][0], results[1][0])
    for a, b in zip(results[0][1:], results[1][1:]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("d,m,hidden,final,N", [(16, 3, 16, "tanh", 1000), (32, 16, 64, "sigmoid", 4099), (8, 5, 24, None, 77),
                                                 (3, 2, 8, "tanh", 16), (64, 16, 40, "sigmoid", 300)])
def test_last_layer_gradient_kernel_against_a_torch_statement(d, m, hidden, final, N):
    """``tsde_rheun_last_layer_grad``: dL/dW2, dL/db2 of a general diffusion net's last layer from stash rows (hid, p, q, wa, wb),
    against the (N, d m) cotangent formed in torch -- every tile path (outputs not a multiple of 16 or of the 256-output slice,
    rows not a multiple of 16, hidden between the tile widths, strides wider than the widths)."""
    from torchsde_amd import _native, neural_rheun
    f, g = _nets(d, m, hidden, 2, "tanh", None, final, "general", seed=N)
    gen = torch.Generator(device=DEV).manual_seed(N)
    pad = lambda n: (n + 3) // 4 * 4                                                      # noqa: E731
    mk = lambda w: torch.randn(N, pad(w), device=DEV, generator=gen)                      # noqa: E731
    hid, p, q, wa, wb = mk(hidden), mk(d), mk(d), mk(m), mk(m)
    acc_w = torch.zeros(d * m, hidden, device=DEV)
    acc_b = torch.zeros(d * m, device=DEV)
    neural_rheun._general_last_layer(g, acc_w, acc_b, hid, p, q, wa, wb, d, m)
    want_w, want_b = neural_rheun.general_last_layer_reference(
        g.rebuilt([t.double() for t in g.parameters()]), hid.double(), p.double(), q.double(), wa.double(), wb.double(), d, m)
    for got, want in ((acc_w, want_w), (acc_b, want_b)):
        scale = want.abs().max().item()
        assert (got.double() - want).abs().max().item() <= 2e-5 * scale + 1e-6, (got.shape, scale)

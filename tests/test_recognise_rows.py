"""The column interpreter of torchsde_amd/recognise_rows.py and the code it generates (CPU: no kernel is launched)."""
import os

import pytest
import torch
from torch import nn

from torchsde_amd import recognise_rows, specialise
from torchsde_amd.recognise import NotElementwise
from torchsde_amd.sde import ForwardSDE
from workloads.problems import StochasticLorenz as _Lorenz


def _evaluate(found, y, t):
    """The linearised statements evaluated in torch: what the generated model computes, column by column."""
    env = {"time": t}
    table = found.const_table()

    def value(name):
        if name.startswith("x.v["):
            return y[:, int(name[4:-1])]
        if name.startswith("c["):
            return table[int(name[2:-1])]
        if name.startswith("(T)"):
            return torch.tensor(float(name[3:]), dtype=y.dtype)
        return env[name]
    ops = {"add": torch.add, "sub": torch.sub, "mul": torch.mul, "div": torch.div, "neg": torch.neg, "exp": torch.exp,
           "log": torch.log, "sin": torch.sin, "cos": torch.cos, "tanh": torch.tanh, "sigmoid": torch.sigmoid,
           "softplus": nn.functional.softplus, "sqrt": torch.sqrt, "abs": torch.abs, "relu": torch.relu,
           "reciprocal": torch.reciprocal, "square": lambda v: v * v, "cube": lambda v: v * v * v}
    for name, op, operands in found.statements:
        env[name] = ops[op](*[value(o) for o in operands])
    cols = [value(o) * torch.ones(y.shape[0], dtype=y.dtype) for o in found.outputs]
    return torch.stack(cols[:found.d], dim=1), torch.stack(cols[found.d:], dim=1)


def test_the_reference_examples_lorenz_system_is_followed_column_by_column():
    sde = _Lorenz()
    y = torch.randn(7, 3)
    found = recognise_rows.recognise_rows(ForwardSDE(sde), torch.tensor(0.3), y)
    assert found.d == 3 and len(found.statements) == 12 and not found.consts
    f, g = _evaluate(found, y, torch.tensor(0.3))
    assert torch.equal(f, sde.f(None, y)) and torch.equal(g, sde.g(None, y))
    text = specialise.source_rows(found.structure(), 0, torch.float32, 0)
    assert "trajectory_prog_kernel<T, METHOD, 3, RowModel<T>>" in text and "x.v[0] * x.v[2]" in text


class _Indexed(nn.Module):
    noise_type, sde_type = "diagonal", "ito"

    def __init__(self):
        super().__init__()
        self.mu = nn.Parameter(torch.tensor(1.5))
        self.sigma = nn.Parameter(torch.tensor([0.2, 0.3]))

    def f(self, t, y):
        x, v = y[:, 0], y[:, 1]
        return torch.stack([v, self.mu * (1 - x ** 2) * v - x + 0.4 * torch.sin(2 * t)], dim=1)

    def g(self, t, y):
        x, v = y.unbind(dim=1)
        return torch.stack([torch.tanh(v), torch.sigmoid(x)], dim=1) * self.sigma


def test_indexing_unbind_stack_parameters_and_time():
    sde = _Indexed()
    y = torch.randn(9, 2)
    t = torch.tensor(0.7)
    found = recognise_rows.recognise_rows(ForwardSDE(sde), t, y)
    # the parameters are scalar constants of the generated model, passed at every launch (live values)
    assert len(found.consts) == 3 and torch.equal(found.const_table(), torch.tensor([1.5, 0.2, 0.3]))
    f, g = _evaluate(found, y, t)
    with torch.no_grad():
        torch.testing.assert_close(f, sde.f(t, y), rtol=1e-6, atol=1e-7)
        torch.testing.assert_close(g, sde.g(t, y), rtol=1e-6, atol=1e-7)
    again = recognise_rows.recognise_rows(ForwardSDE(sde), t, y, rows=5)
    assert again.structure() == found.structure()


@pytest.mark.parametrize("code,reason", [
    ("y - y.mean(dim=1, keepdim=True)", "mean"),
    ("torch.cat([y[:, 1:], y[:1, :1].expand(y.shape[0], 1)], dim=1)", "picked out|not made of columns"),
    ("y @ torch.eye(3)", "mm"),
    ("torch.cat([y[:, :1], y[:, :1]], dim=1)", "(rows, d)"),
    ("y.roll(1, dims=0)", "roll"),
])
def test_what_is_not_column_arithmetic_is_refused(code, reason):
    class M(nn.Module):
        noise_type, sde_type = "diagonal", "ito"

        def f(self, t, y):
            return eval(code)

        def g(self, t, y):
            return 0.1 * y
    with pytest.raises(NotElementwise, match=reason):
        recognise_rows.recognise_rows(ForwardSDE(M()), torch.tensor(0.0), torch.randn(6, 3))


def test_more_than_eight_channels_is_refused():
    class Wide(nn.Module):
        noise_type, sde_type = "diagonal", "ito"

        def f(self, t, y):
            return -y

        def g(self, t, y):
            return 0.1 * y
    with pytest.raises(NotElementwise, match="more than 8"):
        recognise_rows.recognise_rows(ForwardSDE(Wide()), torch.tensor(0.0), torch.randn(6, 9))


def test_side_effects_end_the_column_interpretation_too():
    class Counts(nn.Module):
        noise_type, sde_type = "diagonal", "ito"

        def __init__(self):
            super().__init__()
            self.register_buffer("nfe", torch.zeros(()))

        def f(self, t, y):
            self.nfe.add_(1)
            return torch.cat([y[:, 1:], y[:, :1]], dim=1)

        def g(self, t, y):
            return 0.3 * y

    class Noisy(Counts):
        def f(self, t, y):
            return torch.cat([y[:, 1:], y[:, :1]], dim=1) + 0.0 * torch.randn(y.shape[1])
    with pytest.raises(NotElementwise, match="existed before"):
        recognise_rows.recognise_rows(ForwardSDE(Counts()), torch.tensor(0.0), torch.randn(6, 3))
    with pytest.raises(NotElementwise, match="random"):
        recognise_rows.recognise_rows(ForwardSDE(Noisy()), torch.tensor(0.0), torch.randn(6, 3))


def test_compiled_programs_find_a_writable_directory(monkeypatch, tmp_path):
    """``~/.cache`` may not be writable (a container's read-only user): the units then go under the temporary directory, and a
    compilation that cannot happen at all is a recorded failure -- the interpreter / stepwise route stays -- not an exception out
    of a solve."""
    import tempfile
    monkeypatch.setenv("TSDE_SPECIALISE_CACHE", "/proc/no_such_place/specialised")
    monkeypatch.setattr(tempfile, "tempdir", str(tmp_path))
    where = specialise.cache_dir()
    assert where.startswith(str(tmp_path)) and os.access(where, os.W_OK)
    monkeypatch.setattr(specialise, "cache_dir", lambda: (_ for _ in ()).throw(OSError("nowhere to write")))
    specialise._compile("0" * 24, "// nothing", "gfx950")
    assert str(specialise._state["0" * 24]).startswith("failed: OSError")
    specialise._state.pop("0" * 24, None)

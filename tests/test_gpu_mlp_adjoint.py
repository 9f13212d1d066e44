"""``sdeint_adjoint(..., adjoint_method="euler" | "milstein")`` on the perceptron-drift module through the matrix-core kernels
(``-m gpu``; torchsde_amd/mlp_adjoint.py, csrc/mlp_adjoint.hip: tsde_adjoint_mlp_diag) against

* the REAL reference's `sdeint_adjoint` on the same Brownian path (tests/golden/closed_form_adjoint_*.npz, float64);
* the stepwise stochastic adjoint of this package on the same module and path
  (`adjoint_options={"trajectory_kernel": False}`: torch autograd VJPs + tsde_aug_update), which the golden adjoint
  fixtures of tests/test_gpu_adjoint.py pin to the reference for every noise type.

Both comparisons are up to the summation order of the matrix products (f32 MFMA accumulation vs library GEMMs)."""
import os

import pytest
import torch

from tests import helpers

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _cases():
    return sorted(f[len("closed_form_adjoint_"):-4] for f in os.listdir(helpers.GOLDEN)
                  if f.startswith("closed_form_adjoint_"))


def _close(got, want, what, tol=2e-3):
    err = (got.double().cpu() - want.double().cpu()).abs().max().item()
    scale = want.abs().max().item()
    assert err <= tol * scale + 1e-7, f"{what}: max error {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("name", _cases())
def test_matches_the_reference_adjoint(name):
    import torchsde_amd
    z = helpers.load(f"closed_form_adjoint_{name}.npz")
    B, d, hidden, steps = (int(v) for v in z["shape"])
    dt = float(z["dt"])
    sde = helpers.mlp_module_from(z, torch.float32, DEV)
    y0 = torch.tensor(z["y0"], dtype=torch.float32, device=DEV, requires_grad=True)
    ts = torch.tensor(z["ts"], dtype=torch.float32, device=DEV)
    bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(B, d), dtype=torch.float32, device=DEV,
                                       entropy=int(z["entropy"]), dt=dt)
    ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=bm, method=str(z["method"]),
                                     adjoint_method=str(z["adjoint_method"]), dt=dt)
    assert type(ys.grad_fn).__name__.startswith("_MlpAdjointFn"), "the matrix-core route was not taken"
    (ys * torch.tensor(z["weights"], dtype=torch.float32, device=DEV)).sum().backward()
    _close(ys.detach(), torch.tensor(z["ys"]), "ys", tol=1e-3)
    _close(y0.grad, torch.tensor(z["grad__y0"]), "dL/dy0")
    for pname, p in sde.named_parameters():
        _close(p.grad, torch.tensor(z["grad__" + pname]), f"dL/d{pname}")


SHAPES = [  # B, d, hidden, activation, diffusion, forward method, steps, output marks, sde type, adjoint method
    (64, 32, 32, "tanh", "affine", "euler", 24, (0, 24), "ito", "euler"),
    (100, 64, 64, "softplus", "sigmoid", "euler", 24, (0, 7, 24), "ito", "euler"),
    (48, 128, 128, "softplus", "sigmoid", "euler", 16, (0, 16), "ito", "euler"),
    (37, 20, 36, "tanh", "sigmoid", "milstein", 16, (0, 4, 9, 16), "ito", "euler"),   # padded channels, ragged batch
    (48, 64, 256, "softplus", "affine", "euler", 12, (0, 12), "ito", "euler"),
    (33, 128, 100, "tanh", "affine", "euler", 12, (0, 6, 12), "ito", "euler"),
    # Milstein backward steps (the default adjoint method of a diagonal Ito SDE), Ito and Stratonovich
    (100, 64, 64, "softplus", "sigmoid", "milstein", 24, (0, 7, 24), "ito", "milstein"),
    (48, 128, 128, "softplus", "sigmoid", "euler", 16, (0, 16), "ito", "milstein"),
    (37, 20, 36, "tanh", "sigmoid", "milstein", 16, (0, 4, 9, 16), "ito", None),
    (64, 32, 32, "tanh", "affine", "midpoint", 24, (0, 24), "stratonovich", "milstein"),
    (33, 128, 100, "softplus", "sigmoid", "milstein", 12, (0, 6, 12), "stratonovich", "milstein"),
    # every default: forward SRK (needs the space-time Levy area), backward Milstein
    (64, 64, 64, "softplus", "sigmoid", None, 24, (0, 7, 24), "ito", None),
    (48, 128, 128, "tanh", "affine", "srk", 16, (0, 16), "ito", "euler"),
]


@pytest.mark.parametrize("B,d,hidden,activation,diffusion,method,steps,marks,sde_type,adjoint_method", SHAPES)
def test_matches_the_stepwise_stochastic_adjoint(B, d, hidden, activation, diffusion, method, steps, marks, sde_type,
                                                 adjoint_method):
    import torchsde_amd
    dt = 2.0 ** -6
    gen = torch.Generator().manual_seed(B * 1000 + d)
    sigmoid = diffusion == "sigmoid"
    sde = torchsde_amd.MLPDriftDiagonalSDE(
        d, hidden, activation=activation, diffusion=diffusion, diff_scale=0.4 if sigmoid else 1.0, sde_type=sde_type,
        diff_rate=(2.0 if sigmoid else 0.2) * torch.rand(d, generator=gen) - 0.1,
        diff_shift=0.1 + 0.2 * torch.rand(d, generator=gen)).to(DEV)
    ts = torch.tensor([k * dt for k in marks], device=DEV)
    weights = torch.randn(len(marks), B, d, generator=gen).to(DEV)
    y0_init = (0.5 * torch.randn(B, d, generator=gen)).to(DEV)

    def run(fast, chunk_bytes=None):
        from torchsde_amd import mlp_adjoint
        y0 = y0_init.clone().requires_grad_(True)
        sde.zero_grad()
        bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(B, d), dtype=torch.float32, device=DEV, entropy=808,
                                           levy_area_approximation="space-time" if method in (None, "srk") else "none")
        keep = mlp_adjoint._MlpAdjointFn.STASH_BYTES
        if chunk_bytes is not None:
            mlp_adjoint._MlpAdjointFn.STASH_BYTES = chunk_bytes
        try:
            ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=bm, method=method, adjoint_method=adjoint_method, dt=dt,
                                             adjoint_options={"trajectory_kernel": fast})
            assert type(ys.grad_fn).__name__.startswith("_MlpAdjointFn") == fast
            (ys * weights).sum().backward()
        finally:
            mlp_adjoint._MlpAdjointFn.STASH_BYTES = keep
        grads = {name: p.grad.clone() for name, p in sde.named_parameters()}
        grads["y0"] = y0.grad.clone()
        return ys.detach(), grads

    ys_fast, g_fast = run(True)
    ys_ref, g_ref = run(False)
    _close(ys_fast, ys_ref, "ys", tol=5e-4)
    for key, want in g_ref.items():
        _close(g_fast[key], want, key)
    # chunking is invisible: a stash budget of a few steps gives the same y0 gradient bit for bit (weight gradients up to
    # the order in which the chunks' products are added)
    ys_small, g_small = run(True, chunk_bytes=5 * B * (2 * d + 2 * hidden) * 4)
    assert torch.equal(ys_small, ys_fast) and torch.equal(g_small["y0"], g_fast["y0"])
    for key, want in g_fast.items():
        _close(g_small[key], want, key, tol=1e-4)


def test_other_calls_keep_the_stepwise_adjoint():
    """Another adjoint method than Euler / Milstein, a subset of the parameters, or a grid the backward solver does not
    walk cell by cell: the stepwise stochastic adjoint runs, as before."""
    import torchsde_amd
    B, d, dt = 32, 32, 2.0 ** -5
    sde = torchsde_amd.MLPDriftDiagonalSDE(d, 32, activation="tanh").to(DEV)
    y0 = torch.full((B, d), 0.1, device=DEV, requires_grad=True)

    def grad_fn_of(**kw):
        ts = kw.pop("ts", torch.tensor([0.0, 8 * dt], device=DEV))
        bm = torchsde_amd.BrownianInterval(0.0, 1.0, size=(B, d), dtype=torch.float32, device=DEV, entropy=1)
        ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=bm, method="euler", dt=dt, **kw)
        ys.sum().backward()
        return type(ys.grad_fn).__name__

    assert grad_fn_of(adjoint_method="euler").startswith("_MlpAdjointFn")
    assert grad_fn_of().startswith("_MlpAdjointFn")                                        # default: milstein
    assert not grad_fn_of(adjoint_options={"grad_free": False, "trajectory_kernel": False}).startswith("_MlpAdjointFn")
    assert not grad_fn_of(adjoint_method="euler", adjoint_params=[sde.lin1.weight]).startswith("_MlpAdjointFn")
    assert not grad_fn_of(adjoint_method="euler", ts=torch.tensor([0.0, 0.1], device=DEV)).startswith("_MlpAdjointFn")

"""HIP-graph replay of a whole solve (options={'hip_graph': True}) equals the eager path bit for bit, also
after changing the Brownian seed, the initial state and the SDE parameters between replays."""
import pytest
import torch

from tests import problems

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.mark.parametrize("prob,method,levy,shape", [
    ("gbm_ito", "euler", "none", (256, 16, 16)),
    ("gbm_strat", "midpoint", "none", (256, 16, 16)),
    ("gbm_ito", "srk", "space-time", (128, 8, 8)),
    ("general_ito", "euler", "none", (128, 4, 4)),
    ("gbm_ito", "milstein", "none", (128, 8, 8)),
])
def test_graph_replay_equals_eager(prob, method, levy, shape):
    import torchsde_amd
    B, d, m = shape
    steps, dt = 24, 2.0 ** -6
    ts = torch.tensor([0.0, 10 * dt, steps * dt], device=DEV)
    sde = problems.make(prob, d=d, m=m).to(DEV)
    grad_free = {"grad_free": True} if method == "milstein" else {}   # derivative form needs autograd in-loop

    def solve(entropy, y0, graph):
        bm = torchsde_amd.BrownianInterval(0.0, steps * dt, size=(B, m), device=DEV, dtype=torch.float32,
                                           entropy=entropy, dt=dt, levy_area_approximation=levy)
        opts = dict(grad_free)
        if graph:
            opts["hip_graph"] = True
        with torch.no_grad():
            return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=method, dt=dt, options=opts)

    y_a = torch.full((B, d), 0.1, device=DEV)
    y_b = torch.rand(B, d, device=DEV) * 0.2
    for entropy, y0 in [(1, y_a), (2, y_a), (3, y_b)]:        # first call captures, later calls replay
        assert torch.equal(solve(entropy, y0, True), solve(entropy, y0, False)), (entropy,)
    with torch.no_grad():                                      # parameters are read through pointers
        for p in sde.parameters():
            p.mul_(0.9)
    assert torch.equal(solve(4, y_b, True), solve(4, y_b, False))

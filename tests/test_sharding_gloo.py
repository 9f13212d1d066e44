"""world_size-2 test of the batch-sharding path on CPU (gloo): row partition, global-row RNG offsets and the
gather reproduce the unsharded result. The per-rank compute is the oracle (tests may use it; the product
never does) injected through `solve_fn`, because the HIP kernels need a GPU."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import counter, solvers_ref
from workloads import problems

B, D, STEPS, DT, ENTROPY = 10, 4, 8, 2.0 ** -4, 424242


def _oracle_solve(sde, y0, ts, bm, method, dt, **kw):
    """Stands in for torchsde_amd.sdeint on CPU: reference arithmetic + C twin of the generator, using the
    row offset the product code put on the BrownianInterval."""
    n_rows, m = bm.shape
    edges = np.arange(STEPS + 1) * DT

    def bm_cpu(ta, tb, return_U=False):
        W, _, _ = counter.query(n_rows * m, bm.entropy, edges, float(ta), float(tb), dtype=np.float32,
                                elem0=bm.row_offset * m)
        return torch.from_numpy(W).reshape(n_rows, m)

    with torch.no_grad():
        return solvers_ref.integrate(sde, bm_cpu, y0, ts, dt, method)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from torchsde_amd import sharding
        sde = problems.make("gbm_ito", d=D)
        y0 = torch.linspace(0.05, 0.2, B * D).reshape(B, D)
        ts = torch.tensor([0.0, STEPS * DT])
        final = sharding.sdeint_sharded(sde, y0, ts, entropy=ENTROPY, method="euler", dt=DT, solve_fn=_oracle_solve)
        every = sharding.sdeint_sharded(sde, y0, ts, entropy=ENTROPY, method="euler", dt=DT, gather="all",
                                        solve_fn=_oracle_solve)
        p = torch.nn.Parameter(torch.ones(3))
        p.grad = torch.full((3,), float(rank + 1))
        # a parameter only rank 0 has a gradient for, and a float64 one next to the float32 ones: every rank must
        # reduce over the same flat layout (zeros for the missing gradient), per dtype
        lonely = torch.nn.Parameter(torch.ones(2))
        if rank == 0:
            lonely.grad = torch.full((2,), 5.0)
        wide = torch.nn.Parameter(torch.ones(2, dtype=torch.float64))
        wide.grad = torch.full((2,), 1.0 + 2.0 ** -40, dtype=torch.float64)
        frozen = torch.nn.Parameter(torch.ones(1), requires_grad=False)
        sharding.all_reduce_gradients([p, lonely, wide, frozen])
        assert frozen.grad is None and wide.grad.dtype == torch.float64
        q.put((rank, final.numpy(), every.numpy(), p.grad.numpy(), lonely.grad.numpy(), wide.grad.numpy()))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_unsharded(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    class _BM:   # the unsharded Brownian motion as the product would describe it
        shape, entropy, row_offset = (B, D), ENTROPY, 0
    sde = problems.make("gbm_ito", d=D)
    y0 = torch.linspace(0.05, 0.2, B * D).reshape(B, D)
    ts = torch.tensor([0.0, STEPS * DT])
    full = _oracle_solve(sde, y0, ts, _BM, "euler", DT).numpy()
    for rank, final, every, grad, lonely, wide in results:
        assert np.array_equal(final, full[-1]), rank
        assert np.array_equal(every, full), rank
        assert np.array_equal(grad, np.full(3, sum(range(1, world + 1)), dtype=np.float32))
        assert np.array_equal(lonely, np.full(2, 5.0, dtype=np.float32))
        assert np.array_equal(wide, np.full(2, world * (1.0 + 2.0 ** -40)))        # not rounded through float32


def test_shard_rows_partition():
    from torchsde_amd.sharding import shard_rows
    for B_ in (1, 7, 8, 65536, 262144):
        for world in (1, 2, 3, 8):
            spans = [shard_rows(B_, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == B_
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1

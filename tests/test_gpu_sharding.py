"""The batch-sharding path with the REAL kernels under several processes (run with ``-m gpu``): two and three ranks
share the one GPU of the test box (gloo carries the collectives -- RCCL refuses two ranks on one device -- staging
through the host; the data path is what is under test: row partition, global-row RNG offsets, the gather, the gradient
all-reduce of a sharded adjoint). The gathered result must equal the unsharded solve bit for bit."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
B, D, STEPS, DT, ENTROPY = 1000, 64, 32, 2.0 ** -6, 990077


def _problem():
    from workloads import problems
    return problems.make("gbm_ito", d=D).to("cuda")


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import torchsde_amd
        from torchsde_amd import sharding
        sde = _problem()
        y0 = torch.linspace(0.05, 0.2, B * D, device="cuda").reshape(B, D)
        ts = torch.tensor([0.0, 8 * DT, STEPS * DT], device="cuda")
        with torch.no_grad():
            final = sharding.sdeint_sharded(sde, y0, ts, entropy=ENTROPY, method="euler", dt=DT)
            every = sharding.sdeint_sharded(sde, y0, ts, entropy=ENTROPY, method="srk", dt=DT, gather="all")
        # sharded adjoint: local rows, then one all-reduce of the parameter gradients
        r0, r1 = sharding.shard_rows(B, world, rank)
        y_local = y0[r0:r1].clone().requires_grad_(True)
        bm = torchsde_amd.BrownianInterval(0.0, STEPS * DT, size=(r1 - r0, D), dtype=torch.float32, device="cuda",
                                           entropy=ENTROPY, row_offset=r0)
        ys = torchsde_amd.sdeint_adjoint(sde, y_local, ts, bm=bm, method="euler", adjoint_method="euler", dt=DT)
        ys[-1].sum().backward()
        params = list(sde.parameters())
        sharding.all_reduce_gradients(params)
        # (numpy: pickled by value -- tensors would travel as shared-memory handles that die with this process)
        q.put((rank, final.cpu().numpy(), every.cpu().numpy(), [p.grad.cpu().numpy() for p in params]))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_processes_reproduce_the_unsharded_solve(world):
    import torchsde_amd
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    sde = _problem()
    y0 = torch.linspace(0.05, 0.2, B * D, device="cuda").reshape(B, D)
    ts = torch.tensor([0.0, 8 * DT, STEPS * DT], device="cuda")

    def bm(levy="none"):
        return torchsde_amd.BrownianInterval(0.0, STEPS * DT, size=(B, D), dtype=torch.float32, device="cuda",
                                             entropy=ENTROPY, levy_area_approximation=levy)
    with torch.no_grad():
        full_euler = torchsde_amd.sdeint(sde, y0, ts, bm=bm(), method="euler", dt=DT)
        full_srk = torchsde_amd.sdeint(sde, y0, ts, bm=bm("space-time"), method="srk", dt=DT)
    y = y0.clone().requires_grad_(True)
    ys = torchsde_amd.sdeint_adjoint(sde, y, ts, bm=bm(), method="euler", adjoint_method="euler", dt=DT)
    ys[-1].sum().backward()
    want = [p.grad.cpu() for p in sde.parameters()]
    for rank, final, every, grads in results:
        assert torch.equal(torch.from_numpy(final), full_euler[-1].cpu()), rank
        assert torch.equal(torch.from_numpy(every), full_srk.cpu()), rank
        for got, ref in zip(grads, want):          # a sum over rows regrouped by rank: equal up to summation order
            torch.testing.assert_close(torch.from_numpy(got), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("launcher", ["torch.distributed.run", "self"])
@pytest.mark.parametrize("workload", [None, "c4_midpoint_diag_b32768_d64"])
def test_bench_multi_rank_logic_on_one_gpu(workload, launcher):
    """bench.py as the driver launches it for N > 1 (torch.distributed.run, one rank per GPU) and as the PLAIN command
    `python bench.py --gpus 2` (bench.py starts its own ranks), here with two ranks sharing the one device and gloo for
    the collectives (TSDE_BENCH_SHARE_GPU=1): one JSON line from rank 0, the whole job's trajectory-steps counted,
    weak scaling, and the line says how many ranks the process group really had. Run for the default workload and
    for THE configs[3] run (`--workload c4_midpoint_diag_b32768_d64`: Stratonovich midpoint, 32768 rows per rank)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, TSDE_BENCH_SHARE_GPU="1")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(key, None)
    head = [sys.executable] if launcher == "self" else [
        sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
        "127.0.0.1", "--master-port", str(_free_port())]
    cmd = head + [os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2",
                  "--warmup", "1"] + ([] if workload is None else ["--workload", workload])
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [line for line in proc.stdout.splitlines() if line.startswith("{")]
    assert len(lines) == 1, proc.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 2 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["ranks_seen"] == 2 and len(rec["rank_devices"]) == 2 and rec["collective_backend"].startswith("gloo")
    assert rec["all_gather_ms_per_solve"] > 0
    assert rec["config"]["global_batch"] == 2 * rec["config"]["batch_per_gpu"]
    if workload is not None:
        assert rec["config"]["workload"] == workload and rec["config"]["batch_per_gpu"] == 32768
        assert rec["config"]["method"] == "midpoint"
        assert "configs3" not in rec
    else:
        # the default multi-GPU line (what the driver's scale run launches) carries BASELINE configs[3] beside the headline:
        # sharded Stratonovich midpoint, 32768 rows per rank, gathered -- and rank 0's copy of a foreign shard is bit-identical
        # to a single-rank solve of those global rows
        c3 = rec["configs3"]
        assert "error" not in c3, c3
        assert c3["workload"] == "c4_midpoint_diag_default_route_b32768_d64" and c3["n_gpus"] == 2
        assert c3["global_batch"] == 2 * 32768 and c3["solver_steps"] == 1000 and c3["ranks_seen"] == 2
        assert c3["gathered_shard_equals_single_rank_solve"] is True
        assert c3["all_gather_ms_per_solve"] > 0
        want = c3["global_batch"] * c3["solver_steps"] / (c3["ms_per_step"] * 1e-3)
        assert abs(c3["value"] - want) <= 1e-6 * want
    expected = rec["config"]["global_batch"] * rec["config"]["solver_steps"] / (rec["ms_per_step"] * 1e-3)
    assert abs(rec["value"] - expected) <= 1e-6 * expected
    roof = rec["roofline"]
    assert roof["frac"] > 0 and rec["cpu_baseline"] is None and "also" not in rec
    # the solve-level fraction is the SURVEY 8d definition: bytes per trajectory-step x value / (n_gpus x peak)
    if roof["bound"] == "valu":
        # the headline's trajectory kernel: a fraction of the chip's vector-issue capacity (at most 1); the byte-priced
        # figure of SURVEY 8d lives on under `hbm_equivalent`, per GPU
        assert 0 < roof["frac"] <= 1.0 and roof["unit"].endswith("issue-cycles/s")
        hbm = roof["hbm_equivalent"]
        want = hbm["bytes_per_traj_step"] * rec["value"] / 2 / 8e12
        assert abs(hbm["solve_over_hbm_peak"] - want) <= 1e-9 + 1e-6 * want
    else:
        want = roof["bytes_per_traj_step"] * rec["value"] / 2 / (roof["peak"] * 1e9)
        assert abs(roof["solve_frac"] - want) <= 1e-9 + 1e-6 * want


def _run_bench(cmd, env, root):
    import json
    import subprocess
    proc = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert proc.returncode == 0, proc.stderr[-3000:]
    lines = [line for line in proc.stdout.splitlines() if line.startswith("{")]
    assert lines and len(lines[-1]) < 4096, proc.stdout[-2000:]
    return json.loads(lines[-1])


def test_bench_one_rank_over_rccl_matches_the_plain_run():
    """RCCL on the driver's box, every round: bench.py launched by torch.distributed.run with ONE rank opens the `nccl`
    backend (init_process_group(device_id=...)), runs the all_gather_into_tensor of final states after every solve and
    times it -- the same code path as N = 2, 4, 8 -- and its throughput is the plain single-process run's within 3 %
    (so the N = 1 point of a scaling curve agrees with the headline). SURVEY 8(e)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for key in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "TSDE_BENCH_SHARE_GPU"):
        env.pop(key, None)
    tail = [os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "2", "--no-also", "--no-cpu-baseline"]
    dist = _run_bench([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                       "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + tail, env, root)
    plain = _run_bench([sys.executable] + tail, env, root)
    assert dist["collective_backend"].startswith("nccl") and dist["ranks_seen"] == 1 and dist["n_gpus"] == 1
    assert dist["all_gather_ms_per_solve"] > 0 and dist["all_gather_bytes_per_rank"] == 65536 * 64 * 4
    assert plain["collective_backend"] is None and plain["ranks_seen"] == 1
    assert dist["roofline"]["frac"] > 0 and plain["roofline"]["frac"] > 0
    # (20 solves of ~3 ms: the closing barrier over RCCL is the only extra in the timed region)
    assert abs(dist["value"] - plain["value"]) <= 0.03 * plain["value"], (dist["value"], plain["value"])

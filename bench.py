"""Headline benchmark: SDE steps/sec (batch x timesteps / sec) of the fixed-step hot path.

    python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run)

One bench "step" = one full solve of the workload: BASELINE.json configs[1], diagonal-noise Ito
Euler-Maruyama, batch 65536 x state 64, 1000 fixed solver steps (dyadic dt = 2^-10 so the count is exact
in float32), geometric Brownian motion f = mu*y, g = sigma*y as user torch code, Brownian increments
generated in registers by the fused step kernel. Inputs are resident in HBM before the timed region.
N > 1: every rank solves its own 65536 rows (weak scaling; RNG rows are global, so results are the rows an
unsharded run would produce) and the final states are all-gathered once per solve over RCCL.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel: tsde_step_diag; algorithmic bytes per launch
= 16*d bytes per trajectory-step x batch = 4 streams x B*d*4 B, over the kernel's average duration measured
with HIP events inside the library) and `cpu_baseline` (the oracle's port of the reference's CPU algorithm,
timed on this host's cores on a bounded sample). The default single-GPU run also carries `also`: two short side
measurements taken after the timed region (the same job with the SDE in closed form; perceptron-drift sampling and
a perceptron-drift training step on the matrix cores); they are not part of `value`, and `--no-also` skips them.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3   # dense f32-in/f32-accumulate matrix rate (same guide)

from workloads.configs import WORKLOADS, make_problem as _make_problem  # noqa: E402


def _cpu_baseline(cfg, budget_s=20.0):
    """The oracle's restatement of the reference CPU path (tree-based BrownianInterval + Euler loop), timed on
    this host with all cores on a bounded number of solver steps of the same workload."""
    try:
        from oracle import brownian_ref, solvers_ref
    except Exception as e:  # oracle piece missing: report, don't fake
        return {"value": None, "unit": "trajectory-steps/s", "cores": os.cpu_count(), "kind": "port",
                "sample": f"unavailable: {e}"}
    if cfg.get("adjoint") or cfg.get("train"):
        return {"value": None, "unit": "trajectory-steps/s", "cores": os.cpu_count(), "kind": "port",
                "sample": "not timed for the forward + backward workloads"}
    B, d, dt = cfg["B"], cfg["d"], cfg["dt"]
    sde = _make_problem(cfg["problem"], d, cfg["m"], "cpu")
    y0 = torch.full((B, d), 0.1)
    t1 = cfg["nsteps"] * dt
    step = solvers_ref.STEPS[cfg["method"]]

    def run(threads, budget, min_steps):
        torch.set_num_threads(threads)
        bm = brownian_ref.BrownianIntervalRef(t0=0.0, t1=t1, size=(B, cfg["m"]), dtype=torch.float32, entropy=20240601,
                                              dt=dt, levy_area_approximation=cfg["levy"])
        n, y, t = 0, y0, torch.tensor(0.0)
        start = time.perf_counter()
        with torch.no_grad():
            while n < cfg["nsteps"]:
                t_next = t + dt
                y = step(sde, bm, t, t_next, y)
                t = t_next
                n += 1
                if n >= min_steps and time.perf_counter() - start > budget:
                    break
        return n, time.perf_counter() - start

    # torch CPU elementwise ops on 4M-element tensors do not scale to hundreds of threads; give the
    # baseline its best thread count (probed on a few steps each) and report the count used.
    ncpu = os.cpu_count() or 1
    candidates = sorted({min(ncpu, c) for c in (8, 16, 32, 64, ncpu)})
    best, best_rate = candidates[0], 0.0
    for c in candidates:
        n, el = run(c, 1.5, 2)
        if n / el > best_rate:
            best, best_rate = c, n / el
    n, el = run(best, budget_s, 8)
    return {"value": B * n / el, "unit": "trajectory-steps/s", "cores": best, "kind": "port",
            "host_cpus": ncpu,
            "sample": f"{n} of {cfg['nsteps']} solver steps of the same workload (B={B}, d={d}) in {el:.1f} s; "
                      f"oracle port of the reference CPU algorithm (tree BrownianInterval + Euler loop, torch CPU "
                      f"ops), best of thread counts {candidates} -> {best} threads"}



def _side_measurements(dev):
    """Short measurements reported under `also`: outside the headline's timed region and NOT part of `value`.

    The headline job when the SDE is handed over in closed form (whole solve in one launch), and neural-SDE sampling
    on the matrix cores. A failure here is reported in place and never takes the headline down with it.
    """
    also = {}
    for name in ("c2_euler_closed_form_b65536_d64_s1000", "c5_sampling_mlp_b32768_d128_s500",
                 "c5_training_mlp_b32768_d128_s500"):
        try:
            also[name] = _side_measurement(dev, WORKLOADS[name])
        except Exception as e:
            also[name] = {"error": f"{type(e).__name__}: {e}"}
    return also


def _side_measurement(dev, c):
    import torchsde_amd
    sde = _make_problem(c["problem"], c["d"], c["m"], dev)
    train = c.get("train", False)
    y0 = torch.full((c["B"], c["d"]), 0.1, device=dev, requires_grad=train)
    t1 = c["nsteps"] * c["dt"]
    ts = torch.tensor([0.0, t1], device=dev)

    def solve(i):
        bm = torchsde_amd.BrownianInterval(t0=0.0, t1=t1, size=(c["B"], c["m"]), dtype=torch.float32, device=dev,
                                           entropy=777 + i, dt=c["dt"], levy_area_approximation=c["levy"])
        if train:               # forward + loss.backward() through the solver
            ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=c["method"], dt=c["dt"])
            y0.grad = None
            sde.zero_grad()
            ys[-1].sum().backward()
            return y0.grad
        with torch.no_grad():
            return torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=c["method"], dt=c["dt"])

    for i in range(2):
        solve(i)
    torch.cuda.synchronize()
    start = time.perf_counter()
    for i in range(5):
        out = solve(10 + i)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - start) / 5 * 1e3
    assert torch.isfinite(out).all()
    rec = {"ms_per_solve": ms, "trajectory_steps_per_s": c["B"] * c["nsteps"] / ms * 1e3, "kernel": c["kernel"]}
    if train:
        rec["what"] = "forward + backward per training step"
    elif c.get("mfma_flops_per_traj_step"):
        rec["tflops_f32"] = c["mfma_flops_per_traj_step"] * c["B"] * c["nsteps"] / ms / 1e9
    return rec

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2_euler_diag_b65536_d64_s1000")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the short side measurements reported under `also`")
    ap.add_argument("--eager", action="store_true", help="issue every solve eagerly instead of replaying a HIP graph")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    # TSDE_BENCH_SHARE_GPU=1 (tests only): all ranks use device 0 and gloo carries the collectives, so that the
    # multi-rank logic of this file can be exercised on a one-GPU box (RCCL refuses two ranks on one device)
    share_gpu = os.environ.get("TSDE_BENCH_SHARE_GPU") == "1"
    device_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    # launched by torch.distributed.run (also with a single rank): use the collective path, so that the very same
    # code runs at N = 1, 2, 4, 8
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    import torchsde_amd
    from torchsde_amd import kernels as K

    cfg = WORKLOADS[args.workload]
    B, d, m, nsteps, dt = cfg["B"], cfg["d"], cfg["m"], cfg["nsteps"], cfg["dt"]
    adjoint = cfg.get("adjoint", False)
    # Forward solves are captured once into a HIP graph and replayed (the warm-up solves pay for the capture);
    # the derivative form of Milstein runs its diffusion VJP through autograd INSIDE the captured region (fine: same
    # kernels every step), and so does the Levy-area JVP of the general-noise Milstein extension; the adjoint replays
    # one graph for the forward solve and one for the backward sweep.
    trajectory = cfg.get("trajectory", False)
    use_graph = (not args.eager) and not trajectory
    extra_options = dict(cfg.get("options") or {})
    sde = _make_problem(cfg["problem"], d, m, dev)
    train = cfg.get("train", False)
    y0 = torch.full((B, d), 0.1, device=dev, requires_grad=adjoint or train)
    ts = torch.tensor([0.0, nsteps * dt], device=dev)
    gathered = torch.empty((world * B, d), device=dev) if use_dist else None

    def one_solve(i, graph=None):
        graph = use_graph if graph is None else graph
        bm = torchsde_amd.BrownianInterval(t0=0.0, t1=nsteps * dt, size=(B, m), dtype=torch.float32, device=dev,
                                           entropy=20240601 + i, dt=dt, levy_area_approximation=cfg["levy"],
                                           row_offset=rank * B)
        if adjoint:
            with torch.enable_grad():
                gopt = {"hip_graph": True} if graph else {}
                ys = torchsde_amd.sdeint_adjoint(sde, y0, ts, bm=bm, method=cfg["method"],
                                                 adjoint_method=cfg["adjoint_method"], dt=dt, options=dict(gopt),
                                                 adjoint_options=dict(gopt))
                y0.grad = None
                ys[-1].sum().backward()
            if use_dist:
                from torchsde_amd import sharding
                sharding.all_reduce_gradients(list(sde.parameters()))
            return y0.grad
        if train:
            with torch.enable_grad():
                ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=cfg["method"], dt=dt)
                y0.grad = None
                sde.zero_grad()
                ys[-1].sum().backward()
            if use_dist:
                from torchsde_amd import sharding
                sharding.all_reduce_gradients(list(sde.parameters()))
            return y0.grad
        ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method=cfg["method"], dt=dt,
                                 options=dict(extra_options, hip_graph=True) if graph else (extra_options or None))
        if use_dist:
            dist.all_gather_into_tensor(gathered, ys[-1])
            return gathered
        return ys[-1]

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.no_grad():
        for i in range(args.warmup):
            one_solve(i)
        barrier()
        t_start = time.perf_counter()
        for i in range(args.steps):
            out = one_solve(1000 + i)
        barrier()
        elapsed = time.perf_counter() - t_start
        if use_dist:
            t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = t.item()

        # roofline of the dominant kernel: every launch of the step kernel in one more solve is bracketed by
        # HIP events on the launch stream inside the library (tsde_prof_begin / tsde_prof_end).
        # (issued eagerly: event records are host-side calls and are not part of a replayed graph)
        if trajectory:
            K.prof_begin(cfg["kid"], 16 if not train else 8 * (nsteps + 1))
            for i in range(8):
                one_solve(5000 + i, graph=False)
        else:
            K.prof_begin(cfg["kid"], nsteps * cfg["launches_per_step"] + 8)
            # park the stream while the host enqueues the whole solve, so that no bracket contains queue-empty time
            K.gpu_delay(min(2.0e6, 40.0 * nsteps * (2 + cfg["launches_per_step"])), dev)
            one_solve(5000, graph=False)
        torch.cuda.synchronize()
        k_ms, k_launches = K.prof_end()

        # Second HIP-event measurement of the same kernel without per-launch markers: ONE event pair around a run
        # of back-to-back launches on live data (the last state and its f, g), so marker latency is amortised away.
        b2b_us = None
        if cfg["kid"] == 1:
            from torchsde_amd.kernels import NoiseSpec, _raw_step_diag
            yy = [out[:B].clone().contiguous(), torch.empty(B, d, device=dev)]
            ff, gg = sde.f(ts[0], yy[0]).contiguous(), sde.g(ts[0], yy[0]).contiguous()
            cf = float(dt) if cfg["method"] != "midpoint" else float(dt)
            specs = [NoiseSpec((B, m), torch.float32, dev, entropy=7, elem0=0, cell=i, h=dt) for i in range(220)]
            for i in range(20):
                _raw_step_diag(yy[i & 1], ff, gg, cf, 1.0, specs[i], yy[(i + 1) & 1])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            K.gpu_delay(3000.0, dev)          # let the host enqueue all launches first
            e0.record()
            for i in range(20, 220):
                _raw_step_diag(yy[i & 1], ff, gg, cf, 1.0, specs[i], yy[(i + 1) & 1])
            e1.record()
            torch.cuda.synchronize()
            b2b_us = e0.elapsed_time(e1) * 1e3 / 200
    assert torch.isfinite(out).all()

    value = world * B * nsteps * args.steps / elapsed
    roofline = None
    if k_launches > 0 and trajectory:
        # One launch per solve: it reads y0 and writes the requested outputs, nothing else touches HBM. The kernel is
        # bound by the VALU work of the counter RNG (Philox-4x32-10 + Box-Muller per element-step), so its HBM
        # roofline fraction is ~0 by design; `valu_*` restate the same launch against the vector-ALU issue peak.
        avg_s = k_ms * 1e-3 / k_launches
        if cfg.get("mfma_flops_per_traj_step"):
            # 8 solves were timed; a solve is one launch, or (reverse sweep) one launch per chunk of steps
            flops = cfg["mfma_flops_per_traj_step"] * B * nsteps * 8 / k_launches
            achieved = flops / avg_s / 1e12
            roofline = {"bound": "mfma", "kernel": cfg["kernel"], "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": achieved / MFMA_F32_PEAK_TFLOPS, "traffic": None,
                        "flops_per_launch": flops, "avg_launch_us": avg_s * 1e6, "launches_timed": k_launches,
                        "note": "f32-in / f32-accumulate MFMA (exact f32); peak = dense f32 matrix rate of "
                                "guides/MI355X_MICROARCH.md",
                        "timing": "HIP events bracketing every launch of this kernel in 8 eagerly issued solves"}
        bytes_per_launch = 2 * B * d * 4
        achieved = bytes_per_launch / avg_s / 1e9
        roofline = roofline or {"bound": "hbm", "kernel": cfg["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "bytes_per_launch": bytes_per_launch,
                    "avg_launch_us": avg_s * 1e6, "launches_timed": k_launches,
                    "element_steps_per_s": B * d * nsteps / avg_s,
                    "note": "VALU-bound by design (state in registers, increments from the counter RNG); see DESIGN.md "
                            "for the VALU utilisation measured with rocprofv3 PMC counters",
                    "timing": "HIP events bracketing the single launch of each of 8 eagerly issued solves"}
    elif k_launches > 0:
        # Each bracket is (event record, kernel, event record) on the launch stream of an eagerly issued solve that
        # was fully enqueued behind a delay kernel (the queue never runs dry). A bracket = kernel + the marker
        # packets' latency, i.e. an UPPER bound on the kernel time, and `achieved` is therefore a LOWER bound.
        # The marker latency cannot be calibrated away reliably on this stack (a bracket around a self-timed
        # single-thread spin kernel costs `event_bracket_overhead_us`, more than around a streaming kernel), so
        # nothing is subtracted; the pure kernel duration is in the rocprofv3 kernel trace of this same command
        # (profiles/, `kernel_us_rocprofv3` below when the summary is present).
        raw_s = k_ms * 1e-3 / k_launches
        # `achieved` uses the back-to-back HIP-event figure when it exists: it is the one that agrees with the
        # rocprofv3 kernel trace of this command (profiles/); the per-launch bracket is kept as an upper bound.
        avg_s = raw_s if b2b_us is None else b2b_us * 1e-6
        bytes_per_launch = cfg["bytes_per_traj_step"] * B / cfg["launches_per_step"]
        achieved = bytes_per_launch / avg_s / 1e9
        traffic = rocprof_us = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath) and args.workload == "c2_euler_diag_b65536_d64_s1000":
            # HBM bytes per launch from rocprofv3 PMC passes of this same command (tools/profile.sh), corrected as
            # guides/MI355X_MICROARCH.md prescribes; collected offline because counters need their own passes.
            try:
                with open(tpath) as fh:
                    for kname, rec in json.load(fh).items():
                        if "StepDiagOp<float>" in kname:
                            traffic = rec["traffic_bytes_per_launch"]
                            rocprof_us = rec.get("kernel_avg_us")
            except Exception:
                traffic = None
        roofline = {"bound": "hbm", "kernel": cfg["kernel"],
                    "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                    "traffic": traffic, "bytes_per_launch": bytes_per_launch, "avg_launch_us": avg_s * 1e6, "bracket_us_in_situ": raw_s * 1e6,
                    "kernel_us_rocprofv3": rocprof_us,
                    "timing": ("HIP events around 200 back-to-back launches on live data (avg_launch_us); bracket_us_in_situ = "
                               "HIP events bracketing each of the launches of one eagerly issued solve, an upper bound "
                               "that includes marker-packet latency") if b2b_us is not None else
                              "HIP events bracketing every launch of one eagerly issued solve (upper bound)",
                    "launches_timed": k_launches}
    also = None
    if world == 1 and not args.no_also and args.workload == "c2_euler_diag_b65536_d64_s1000":
        also = _side_measurements(dev)
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = _cpu_baseline(cfg)
        line = {
            "metric": "SDE steps/sec (batch x timesteps / sec)", "value": value, "unit": "trajectory-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "sde": cfg["problem"] + (" (closed-form coefficients; f, g evaluated in the kernel)" if trajectory
                                                          else " (f, g are user torch ops)"),
                       "method": cfg["method"] + ("+adjoint:" + cfg["adjoint_method"] if adjoint else "") +
                                 (" + loss.backward() through the solver" if train else ""),
                       "batch_per_gpu": B, "global_batch": world * B, "state": d, "brownian_channels": m,
                       "solver_steps": nsteps, "dt": dt, "brownian": "counter-RNG, generated in the step kernel",
                       "launch": ("trajectory kernels: one forward launch, the reverse sweep in chunks of steps, two weight-gradient "
                                  "products per chunk" if train else
                                  "one trajectory-kernel launch per solve" if trajectory else
                                  "HIP graph replay of the whole solve" + (" and of the backward sweep" if adjoint else "")
                                  if use_graph else "eager launches"),
                       "parallelism": f"batch-sharded x{world}, one all_gather of final states per solve"},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if also is not None:
            line["also"] = also
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

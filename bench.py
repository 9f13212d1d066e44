"""Headline benchmark: SDE steps/sec (batch x timesteps / sec) of the fixed-step hot path.

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL, either way it is started: under the driver's `python -m torch.distributed.run
--nproc-per-node N ... bench.py --gpus N`, or as the plain command above, in which case bench.py starts the N ranks
itself (and refuses, exit code 3, when the node has fewer than N GPUs). The line reports what the process group
really was (`ranks_seen`, `rank_devices`, `collective_backend`) and the all-gather's own time per solve.

One bench "step" = one full solve of the workload: BASELINE.json configs[1], diagonal-noise Ito Euler-Maruyama, batch
65536 x state 64, 1000 fixed solver steps (dyadic dt = 2^-10 so the count is exact in float32), geometric Brownian
motion f = mu*y, g = sigma*y as an UNCHANGED user module of plain torch code, handed to `sdeint` with no options -- the
drop-in call. torchsde_amd/recognise.py interprets f and g at every solve, finds them per-channel affine, and the solve
is one launch of the trajectory kernel (state in registers, increments from the counter RNG). The same job on the
STEPWISE route (user f, g as torch kernels between the per-step kernels; the route of every SDE that is not a
per-channel expression) is measured right after and reported inside the line as `stepwise`. Inputs are resident in HBM
before the timed region. N > 1: every rank solves its own rows (weak scaling; RNG rows are global, so results are the
rows an unsharded run would produce) and the final states are all-gathered once per solve over RCCL.
BASELINE configs[3] (Stratonovich midpoint, 262144 x 64 sharded over 8 GPUs) is
`--gpus 8 --workload c4_midpoint_diag_default_route_b32768_d64` (stepwise: `c4_midpoint_diag_b32768_d64`): 32768 rows per GPU.

The LAST stdout line (rank 0) is the headline JSON, under 4 KB; the side measurements are printed before it, one short
line per workload, and written whole to bench_also.json:

* `value` = trajectory-steps/s over the K timed solves (barrier + synchronize on both sides, max over ranks);
  `median_ms_per_step` / `value_median` restate it from the median of the per-solve times (HIP events recorded
  between the solves of the same timed region);
* `roofline` (trajectory kernel): `achieved` = SURVEY 8d's bytes per trajectory-step x the trajectory-steps one launch
  processes / the launch's duration (HIP events inside the library, on the launch stream); `frac` = that / 8 TB/s and is
  ABOVE 1, because a one-launch solve never moves those bytes -- `traffic` is what the launch really moves (y0 in,
  final state out). The kernel is VALU-bound (Philox + Box-Muller per element-step);
* `stepwise.roofline`: the per-step kernel's two fractions of the HBM peak, named for what they are --
    `frac`        KERNEL level: the dominant kernel's SURVEY-8d algorithmic bytes per solver step / the duration of its
                  launches, measured live with HIP events around the replay of a HIP graph of 200 launches on live,
                  rotating operands (agrees with the rocprofv3 kernel trace of the solve under profiles/);
    `solve_frac`  SOLVE level: algorithmic bytes per trajectory-step x `value` / (N x 8e12), i.e.
                  with the user's f and g torch kernels and every launch gap inside;
  `traffic` (HBM bytes per launch from rocprofv3 PMC passes, tools/profile_traffic.sh) and `kernel_us_rocprofv3` are
  taken from profiles/traffic_latest.json ONLY when that file was collected on the kernel sources this run uses;
* `cpu_baseline`: the oracle's port of the reference's CPU algorithm, timed on this host's cores on a bounded sample;
* side measurements (single-GPU default run; `also` lines + bench_also.json): every other BASELINE configuration at its
  single-GPU size on the stepwise path (configs[2] Euler-general and the Milstein-general extension, the configs[3]
  shard, configs[4] sdeint_adjoint), each with ms per solve (median of 5) and the same kernel-level measurement, bytes,
  fractions and counter traffic; then the default-route and closed-form variants. Not part of `value`; `--no-also` skips them, `--also-budget S` (default 100 s; 0 = no limit) bounds their total time.
"""
import argparse
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

# (program kernels compiled at run time, torchsde_amd/specialise.py: compile in the calling thread, so that the timed solves of
#  a program workload are the compiled kernel's -- a training loop reaches that state after its first few iterations)
os.environ.setdefault("TSDE_SPECIALISE", "sync")
from workloads.configs import WORKLOADS, make_problem as _make_problem  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3   # dense f32-in/f32-accumulate matrix rate (same guide)
# vector-issue capacity: 256 CUs x 4 SIMD-32 units, one wave64 VALU instruction occupying its SIMD for >= 2 cycles
# (same guide); in units of 1e12 SIMD cycles per second at the nominal 2.4 GHz
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 1e12
HEADLINE = "c2_euler_diag_default_route_b65536_d64_s1000"
# the stepwise BASELINE configurations, then what the closed-form route makes of the same jobs
ALSO = ("c2_milstein_diag", "c2_srk_diag",
        "c3_euler_general_b16384_d32_m16", "c3_milstein_general_b16384_d32_m16",
        "c3_milstein_general_gradfree_b16384_d32_m16", "c3_euler_additive_shared_b16384_d32_m16",
        "c3_euler_additive_shared_b262144_d64_m32", "c4_midpoint_diag_b32768_d64", "c5_adjoint_latent_b32768_d128_s500",
        "c2_euler_expdiff_b65536_d64_s1000",
        "c3_euler_general_default_route_b16384_d32_m16", "c3_midpoint_general_default_route_b16384_d32_m16",
        "c3_euler_general_bf16x3_default_route_b16384_d32_m16",
        "c2_srk_netdiag_b65536_d64_s1000", "c2_srk_netdiag_default_route_b65536_d64_s1000",
        "c2_milstein_diag_default_route", "c2_srk_diag_default_route", "c4_midpoint_diag_default_route_b32768_d64",
        "c2_euler_expdiff_default_route_b65536_d64_s1000", "c2_euler_training_default_route_b65536_d64_s1000",
        "c2_euler_scheduled_b65536_d64_s1000", "c2_euler_scheduled_default_route_b65536_d64_s1000",
        "c2_srk_scheduled_default_route_b65536_d64_s1000",
        "c2_euler_doublewell_b65536_d64_s1000", "c2_euler_doublewell_default_route_b65536_d64_s1000",
        "c2_heun_diag_default_route_b65536_d64_s1000", "c2_heun_diag_b65536_d64_s1000",
        "exadditive_srk_default_route_b65536_d64_m8", "exadditive_euler_default_route_b65536_d64_m8",
        "exadditive_srk_b65536_d64_m8", "exadditive_euler_b65536_d64_m8",
        "neuraladditive_srk_default_route_b65536_d64_m8", "neuraladditive_srk_b65536_d64_m8",
        "c2_euler_exscalar_b65536_d64_s1000", "c2_srk_exscalar_b65536_d64_s1000",
        "c2_srk_exscalar_default_route_b65536_d64_s1000",
        "c2_euler_exscalar_default_route_b65536_d64_s1000", "c2_euler_exscalar_training_default_route_b65536_d64_s1000",
        "c5_sampling_mlp_b32768_d128_s500", "c5_sampling_mlp_srk_b32768_d128_s500", "c5_training_mlp_b32768_d128_s500", "c5_adjoint_mlp_b32768_d128_s500",
        "c5_adjoint_mlp_milstein_b32768_d128_s500", "c5_adjoint_mlp_defaults_b32768_d128_s500",
        "c5_adjoint_latent_default_route_b32768_d128_s500", "c5_adjoint_latent_defaults_default_route_b32768_d128_s500",
        # SURVEY 8(f) rows, measured on the routes they have
        "c5_rheun_adjoint_latent_b32768_d128_s500", "c5_logqp_adjoint_latent_b32768_d128_s500",
        "c3_log_ode_general_b16384_d32_m16",
        # a row-coupled system (the reference's StochasticLorenz): generated model, one lane per row
        "lorenz_euler_default_route_b262144_d3_s1000", "lorenz_euler_b262144_d3_s1000",
        "lorenz_srk_default_route_b1024_d3_s1000", "lorenz_srk_b1024_d3_s1000",
        # the reversible pair on the matrix cores, beside its stepwise twins
        "sdegan_rheun_adjoint_default_route_b1024_d16_m3_s63", "sdegan_rheun_adjoint_b1024_d16_m3_s63",
        "sdegan_midpoint_default_route_b16384_d16_m3_s1000", "sdegan_midpoint_b16384_d16_m3_s1000",
        "c3_rheun_general_default_route_b16384_d32_m16", "c3_rheun_general_b16384_d32_m16",
        "c3_rheun_adjoint_general_default_route_b16384_d32_m16", "c3_rheun_adjoint_general_b16384_d32_m16")


# The default run gives the side measurements a time budget (`--also-budget`, seconds; 0 = no limit): these go first, what
# the budget does not reach is recorded as skipped (`python bench.py --workload W` measures any of them on its own;
# profiles/r6_bench_also.json holds a run without the limit).
ALSO_FIRST = ("c3_euler_general_default_route_b16384_d32_m16", "c3_midpoint_general_default_route_b16384_d32_m16",
              "c4_midpoint_diag_default_route_b32768_d64", "c2_milstein_diag_default_route", "c2_srk_diag_default_route",
              "c5_adjoint_latent_default_route_b32768_d128_s500",
              "c3_rheun_general_default_route_b16384_d32_m16", "c3_rheun_adjoint_general_default_route_b16384_d32_m16",
              "sdegan_rheun_adjoint_default_route_b1024_d16_m3_s63", "lorenz_euler_default_route_b262144_d3_s1000",
              "c2_euler_exscalar_default_route_b65536_d64_s1000", "c2_srk_exscalar_default_route_b65536_d64_s1000",
              "c2_euler_exscalar_training_default_route_b65536_d64_s1000", "exadditive_srk_default_route_b65536_d64_m8",
              "neuraladditive_srk_default_route_b65536_d64_m8", "c2_srk_netdiag_default_route_b65536_d64_s1000",
              "c2_heun_diag_default_route_b65536_d64_s1000", "c5_rheun_adjoint_latent_b32768_d128_s500",
              "c5_logqp_adjoint_latent_b32768_d128_s500", "c2_milstein_diag", "c2_srk_diag", "c3_euler_general_b16384_d32_m16",
              "c4_midpoint_diag_b32768_d64", "c5_adjoint_latent_b32768_d128_s500")
ALSO = ALSO_FIRST + tuple(w for w in ALSO if w not in ALSO_FIRST)


def csrc_digest():
    """sha256 over the kernel sources and the C header (what a PMC measurement is a measurement OF)."""
    h = hashlib.sha256()
    csrc = os.path.join(ROOT, "torchsde_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    files.append(os.path.join(ROOT, "include", "torchsde_amd.h"))
    for path in files:
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def csrc_file_digests():
    """sha256 of every kernel source on its own: a counter file stays a measurement of a kernel as long as the files that kernel
    is made of have not changed, whatever happened to the others."""
    csrc = os.path.join(ROOT, "torchsde_amd", "csrc")
    files = sorted(os.path.join(csrc, f) for f in os.listdir(csrc) if f.endswith((".hip", ".h")))
    files.append(os.path.join(ROOT, "include", "torchsde_amd.h"))
    out = {}
    for path in files:
        with open(path, "rb") as fh:
            out[os.path.basename(path)] = hashlib.sha256(fh.read()).hexdigest()[:16]
    return out


# the files the stepwise kernels (and the affine / expression / program trajectory kernels) are made of: what
# tools/profile_traffic.sh measures
_COMMON_SOURCES = ("tsde_common.h", "tsde_rng.h", "tsde_bridge.h", "tsde_launch.h", "tsde_schemes.h", "torchsde_amd.h")
_STEPWISE_SOURCES = _COMMON_SOURCES + ("steps.hip", "rheun.hip", "brownian.hip", "milstein_general.hip", "capi.hip")
_TRAJECTORY_SOURCES = _COMMON_SOURCES + ("trajectory.hip", "capi.hip")


def _reference_package():
    """The REAL reference (google-research/torchsde) if this host has it: /root/reference (or $TORCHSDE_REFERENCE) with
    the `trampoline` stand-in of tests/golden/_ref_shim ahead of it on the path. None elsewhere -- the
    GPU box has no /root/reference -- and then the baseline is the oracle's port (`kind: "port"`)."""
    root = os.environ.get("TORCHSDE_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(root, "torchsde")):
        return None
    sys.dont_write_bytecode = True       # the reference tree is read-only
    for path in (root, os.path.join(ROOT, "tests", "golden", "_ref_shim")):
        if path not in sys.path:
            sys.path.insert(0, path)
    try:
        import torchsde
        return torchsde
    except Exception:
        return None


def _cpu_baseline(cfg, budget_s=20.0):
    """The reference's CPU path on this host's cores, on a bounded number of solver steps of the same workload: the real
    package when the host has it (`kind: "reference"`: torchsde.sdeint under no_grad, BrownianInterval built for the full
    horizon with the dt hint, ts shortened to the sample), else the oracle's restatement of it (`kind:
    "port"`: tree-based BrownianInterval + solver loop, same torch CPU ops; same bits as the reference at equal thread
    counts and 1.2-1.4x slower than it at 4-8 threads, where both could be run: profiles/r6_cpu_port_vs_reference.txt)."""
    ncpu = os.cpu_count() or 1
    base = {"value": None, "unit": "trajectory-steps/s", "cores": None, "host_cpus": ncpu, "kind": "port"}
    if cfg.get("adjoint") or cfg.get("train"):
        return dict(base, sample="not timed for the forward + backward workloads")
    B, d, dt = cfg["B"], cfg["d"], cfg["dt"]
    sde = _make_problem(cfg["problem"], d, cfg["m"], "cpu")
    y0 = torch.full((B, d), 0.1)
    t1 = cfg["nsteps"] * dt
    reference = _reference_package()
    if reference is not None:
        kind, what = "reference", "the reference package itself (torchsde.sdeint, CPU)"

        def run(threads, n):
            torch.set_num_threads(threads)
            bm = reference.BrownianInterval(t0=0.0, t1=t1, size=(B, cfg["m"]), dtype=torch.float32, entropy=20240601,
                                            dt=dt, levy_area_approximation=cfg["levy"])
            start = time.perf_counter()
            with torch.no_grad():
                reference.sdeint(sde, y0, torch.tensor([0.0, n * dt]), bm=bm, method=cfg["method"], dt=dt)
            return time.perf_counter() - start
    else:
        try:
            from oracle import brownian_ref, solvers_ref
        except Exception as e:  # oracle piece missing: report, don't fake
            return dict(base, sample=f"unavailable: {e}")
        kind, what = "port", ("oracle port of the reference CPU algorithm (A/B vs the reference where it exists: bit-identical, 1.2-1.4x "
                      "slower, profiles/r6_cpu_port_vs_reference.txt)")
        step = solvers_ref.STEPS[cfg["method"]]

        def run(threads, n):
            torch.set_num_threads(threads)
            bm = brownian_ref.BrownianIntervalRef(t0=0.0, t1=t1, size=(B, cfg["m"]), dtype=torch.float32,
                                                  entropy=20240601, dt=dt, levy_area_approximation=cfg["levy"])
            y, t = y0, torch.tensor(0.0)
            start = time.perf_counter()
            with torch.no_grad():
                for _ in range(n):
                    y = step(sde, bm, t, t + dt, y)
                    t = t + dt
            return time.perf_counter() - start

    # torch CPU elementwise ops on 4M-element tensors do not scale to hundreds of threads; give the baseline its
    # best thread count (probed on a few steps each) and report the count used
    candidates = sorted({min(ncpu, c) for c in (8, 16, 32, 64, ncpu)})
    run(candidates[0], 1)                      # lazy initialisation out of the way
    probe = {c: run(c, 3) / 3 for c in candidates}
    best = min(probe, key=probe.get)
    n = int(max(8, min(cfg["nsteps"], budget_s / probe[best])))
    elapsed = run(best, n)
    return {"value": B * n / elapsed, "unit": "trajectory-steps/s", "cores": best, "host_cpus": ncpu, "kind": kind,
            "sample": f"{n} of {cfg['nsteps']} solver steps of the same workload (B={B}, d={d}, {cfg['method']}) in "
                      f"{elapsed:.1f} s; {what}; best of thread counts {candidates}"}


class Job:
    """One workload set up on one device: `solve(i)` runs one full solve (forward, or forward + backward) with the
    Brownian seed of solve i; inputs live in HBM before any timing starts."""

    def __init__(self, name, dev, rank=0, world=1, dist=None, graph=True):
        import torchsde_amd
        self.name, self.cfg, self.dev, self.rank, self.world, self.dist = name, WORKLOADS[name], dev, rank, world, dist
        c = self.cfg
        self.trajectory = c.get("trajectory", False)
        self.adjoint, self.train = c.get("adjoint", False), c.get("train", False)
        # Forward solves are captured once into a HIP graph and replayed (the warm-up solves pay for the capture); the
        # adjoint replays one graph for the forward solve and one for the backward sweep; trajectory kernels are one
        # launch per solve anyway.
        self.use_graph = graph and not self.trajectory and not c.get("eager", False)
        self.sde = _make_problem(c["problem"], c["d"], c["m"], dev)
        self.y0 = torch.full((c["B"], c["d"]), 0.1, device=dev, requires_grad=self.adjoint or self.train)
        self.ts = torch.tensor([0.0, c["nsteps"] * c["dt"]], device=dev)
        if c.get("output_every_step"):      # (examples/sde_gan.py: an output at every step of the grid)
            self.ts = torch.arange(c["nsteps"] + 1, device=dev, dtype=torch.float32) * c["dt"]
        # two gather buffers: the all-gather of solve i runs (on RCCL's stream) while solve i + 1 computes
        self.gathered = [torch.empty((world * c["B"], c["d"]), device=dev) for _ in range(2)] if dist is not None else None
        self._gathers, self._pending = 0, []
        self._prepared = not c.get("method") == "reversible_heun"       # (the pair's first solve is the verifying one: stepwise)
        self._sdeint, self._sdeint_adjoint = torchsde_amd.sdeint, torchsde_amd.sdeint_adjoint
        self._BM = torchsde_amd.BrownianInterval

    def solve(self, i, graph=None):
        c = self.cfg
        graph = self.use_graph if graph is None else graph
        bm = self._BM(t0=0.0, t1=c["nsteps"] * c["dt"], size=(c["B"], c["m"]), dtype=torch.float32, device=self.dev,
                      entropy=20240601 + i, dt=c["dt"], levy_area_approximation=c["levy"], row_offset=self.rank * c["B"])
        extra_options = dict(c.get("options") or {})
        if not self.trajectory:
            # a stepwise measurement: the GBM / exp-diffusion user modules would otherwise be recognised as per-channel
            # expressions and run as one trajectory launch (torchsde_amd/recognise.py) -- that is the headline's route
            extra_options.setdefault("trajectory_kernel", False)
        if self.adjoint or self.train:
            with torch.enable_grad():
                if self.adjoint:
                    gopt = dict(extra_options, hip_graph=bool(graph))      # (False, not absent: the default is "auto")
                    plain = bool(c.get("recognised"))                      # the drop-in call: no options at all
                    ys = self._sdeint_adjoint(self.sde, self.y0, self.ts, bm=bm, method=c["method"],
                                              adjoint_method=c["adjoint_method"], dt=c["dt"],
                                              options=None if plain else dict(gopt),
                                              adjoint_options=None if plain else dict(gopt), logqp=bool(c.get("logqp")))
                    if c.get("logqp"):
                        # (ys, log-ratio (T - 1, B)): a latent-SDE user's loss has a term of the path and the KL term
                        states, log_ratio = ys
                        ys = torch.cat([states[-1], log_ratio.sum(0).unsqueeze(-1)], dim=1).unsqueeze(0)
                    if plain and not type(ys.grad_fn).__name__.startswith(("_MlpAdjointFn", "ReversibleHeunFn")) \
                            and self._prepared:
                        raise RuntimeError(f"{self.name}: sdeint_adjoint did not take the matrix-core route")
                else:
                    ys = self._sdeint(self.sde, self.y0, self.ts, bm=bm, method=c["method"], dt=c["dt"],
                                      options=dict(extra_options) or None)
                self.y0.grad = None
                self.sde.zero_grad()
                (ys.sum() if c.get("output_every_step") else ys[-1].sum()).backward()
            if self.dist is not None:
                from torchsde_amd import sharding
                sharding.all_reduce_gradients(list(self.sde.parameters()))
            return self.y0.grad
        with torch.no_grad():
            # (a recognised workload is the drop-in call: no options at all)
            ys = self._sdeint(self.sde, self.y0, self.ts, bm=bm, method=c["method"], dt=c["dt"],
                              options=(dict(extra_options) or None) if c.get("recognised")
                              else dict(extra_options, hip_graph=bool(graph)))
            if self.dist is not None:
                # The one collective of a solve, issued asynchronously: it waits for this solve's kernels on RCCL's own
                # stream and overlaps the NEXT solve's compute; at most two are in flight (double buffer), the one before
                # the previous is waited for here, and `finish()` -- called before the closing barrier of every timed
                # region -- waits for the rest. (With a 3 ms solve a blocking 8-rank gather of 128 MiB would be a
                # visible part of every step; at 26 ms it was not.)
                out = self.gathered[self._gathers & 1]
                self._gathers += 1
                work = self.dist.all_gather_into_tensor(out, ys[-1], async_op=True)
                self._pending.append((work, ys))
                while len(self._pending) > 1:
                    self._pending.pop(0)[0].wait()
                return out
            return ys[-1]

    def finish(self):
        """Wait (stream-wise) for every all-gather still in flight; the caller synchronises afterwards."""
        while self._pending:
            self._pending.pop(0)[0].wait()

    def prepare(self):
        """Everything a first solve pays once, before any warm-up or timing: recording the HIP graph(s) of this workload
        (with the checks that come with it) and their probation -- the first two replays of a recorded graph run next
        to the eager path and are compared with it (torchsde_amd/graph.py)."""
        if self.use_graph or self.cfg.get("recognised"):
            for i in range(4):
                self.solve(9000 + i)
                self._prepared = True
            torch.cuda.synchronize()
        if self.cfg.get("recognised"):
            # the first solve ran both ways and compared (solvers._integrate_recognised); the timed ones must be launches
            from torchsde_amd import solvers
            book = getattr(self.sde, solvers.BaseSDESolver._RECOGNISED_ATTR, None)
            if not book or list(book["trusted"].values()) != [True]:
                why = list((book or {}).get("refused", {}).values()) + [v for v in (book or {}).get("trusted", {}).values()
                                                                         if v is not True]
                raise RuntimeError(f"{self.name}: the user module was not routed to the trajectory kernel: {why}")

    def live_state(self, out):
        """A state tensor of this workload's shape with live values: the solve's final state where `out` is one."""
        B, d = self.cfg["B"], self.cfg["d"]
        if not (self.adjoint or self.train) and out.dim() == 2 and out.shape[1] == d and out.shape[0] >= B:
            return out[:B].detach()
        return self.y0.detach()

    # ---- dominant kernel -----------------------------------------------------------------------------------------
    def bracket_dominant_kernel(self):
        """Trajectory kernels only (one launch of milliseconds per solve, so the ~2.5 us of marker latency a bracket adds
        is noise): (total ms, launches) of the dominant kernel in 8 eagerly issued solves, bracketed by HIP events on
        the launch stream inside the library (tsde_prof_begin / tsde_prof_end)."""
        from torchsde_amd import kernels as K
        c = self.cfg
        assert self.trajectory
        K.prof_begin(c["kid"], 16 if not (self.train or self.adjoint) else 8 * (c["nsteps"] + 1))
        for i in range(8):
            self.solve(5000 + i, graph=False)
        torch.cuda.synchronize()
        return K.prof_end()

    def back_to_back_us(self, live_state):
        """THE kernel-level timing of every stepwise workload: 200 launches of the dominant kernel recorded into ONE HIP
        graph (the launch regime of the timed region) and replayed between one pair of HIP events on the replay's stream
        -- no marker packets between launches, no host launch rate in the figure. The operands are live (a solve's
        last state, the SDE's f and g there) and ROTATE over enough distinct copies that a launch never finds its inputs in
        the per-XCD L2 from the launch before (a loop over constant f, g shows 4.1 us = "103 % of the HBM peak" at the
        8 MiB-per-stream shard size where the solve's kernel takes 6.7 us; with the rotation the figure agrees with the
        rocprofv3 kernel trace of the solve, profiles/). Returns {label: us per launch} (one entry, or the four SRK
        stages), each the best of three replays."""
        from torchsde_amd import kernels as K
        from torchsde_amd.kernels import NoiseSpec
        c, dev, sde = self.cfg, self.dev, self.sde
        B, d, m, dt, kid = c["B"], c["d"], c["m"], float(c["dt"]), c["kid"]
        if kid not in (1, 2, 3, 4, 5, 7, 11, 12):
            return None              # (a workload without a dominant kernel of its own to time back to back)
        n = 200
        y = live_state[:B].detach().clone().contiguous()
        t0 = self.ts[0]
        spec = [NoiseSpec((B, m), torch.float32, dev, entropy=7, elem0=0, cell=i, h=dt) for i in range(n)]
        with torch.no_grad():
            f, g = (None, None) if kid == 5 else (sde.f(t0, y).contiguous(), sde.g(t0, y))
            g = g if kid in (5, 12) else g.contiguous()
            if kid in (1, 3, 4) and g.dim() == 3 and g.shape[-1] == 1:      # scalar noise: g (B, d, 1), one increment per row
                g = g.squeeze(-1).contiguous()
                spec = [NoiseSpec((B, d), torch.float32, dev, entropy=7, elem0=0, cell=i, h=dt) for i in range(n)]

        def copies(*tensors):
            """Enough distinct copies of an operand set that the rotation's working set is >= 128 MiB (4x the L2s)."""
            nbytes = sum(t.numel() * t.element_size() for t in tensors)
            k = max(2, min(32, -(-(128 << 20) // max(nbytes, 1))))
            return [tuple(t.clone() for t in tensors) for _ in range(k)]

        if kid == 1:
            sets = copies(y, f, g)
            coefs = [(dt, 1.0)] if c["launches_per_step"] == 1 else [(0.5 * dt, 0.5), (dt, 1.0)]   # midpoint: two stages

            def launch(i):
                (ya, fa, ga), yb = sets[i % len(sets)], sets[(i + 1) % len(sets)][0]
                cf, cg = coefs[i % len(coefs)]
                K._raw_step_diag(ya, fa, ga, cf, cg, spec[i], yb)
            out = {"tsde_step_diag": _graph_replay_us([(lambda i=i: launch(i)) for i in range(n)], dev)}
            if "tsde_heun_final" in (c.get("step_kernels") or {}):
                hsets = copies(y, f, f, g, g)

                def final(i):
                    (ya, fa, fb, ga, gb), yb = hsets[i % len(hsets)], hsets[(i + 1) % len(hsets)][0]
                    K.heun_final(ya, fa, fb, ga, gb, dt, 0, spec[i], out=yb)
                out["tsde_heun_final"] = _graph_replay_us([(lambda i=i: final(i)) for i in range(n)], dev)
            return out
        if kid == 7:
            # reversible Heun, diagonal noise: the four elementwise kernels of a forward + backward step
            sets = copies(y, y.clone(), f, g, f.clone(), g.clone())

            def z_launch(i):
                (ya, za, fa, ga, _, _), zb = sets[i % len(sets)], sets[(i + 1) % len(sets)][1]
                K.rheun_z(ya, za, fa, ga, dt, 1.0, spec[i], out=zb)

            def y_launch(i):
                (ya, _, fa, ga, fb, gb), yb = sets[i % len(sets)], sets[(i + 1) % len(sets)][0]
                K.rheun_y(ya, fa, fb, ga, gb, 0.5 * dt, 1.0, spec[i], out=yb)

            def a_launch(i):
                ya, _, fa, ga, _, _ = sets[i % len(sets)]
                K.rheun_adj_a(ya, fa, ga, 0.5 * dt, spec[i])

            def b_launch(i):
                ya, za, fa, _, _, _ = sets[i % len(sets)]
                K.rheun_adj_b(ya, za, fa, dt, 0.5 * dt, spec[i])
            return {label: _graph_replay_us([(lambda i=i, fn=fn: fn(i)) for i in range(n)], dev)
                    for label, fn in (("tsde_rheun_z", z_launch), ("tsde_rheun_y", y_launch), ("tsde_rheun_adj_a", a_launch),
                                      ("tsde_rheun_adj_b", b_launch))}
        if kid == 2:
            sets = copies(y, f, g)

            def launch(i):
                (ya, fa, ga), yb = sets[i % len(sets)], sets[(i + 1) % len(sets)][0]
                K._raw_step_general(ya, fa, ga, dt, 1.0, spec[i], yb)
            return {"tsde_step_general": _graph_replay_us([(lambda i=i: launch(i)) for i in range(n)], dev)}
        if kid == 12:
            S = g[0].contiguous()           # the one (d, m) matrix behind sigma.expand(B, d, m)
            sets = copies(y, f)

            def launch(i):
                (ya, fa), yb = sets[i % len(sets)], sets[(i + 1) % len(sets)][0]
                K._raw_step_shared(ya, fa, S, 1.0, dt, 1.0, 0, 0.0, 0.0, 0.0, spec[i], yb)
            return {"tsde_step_shared": _graph_replay_us([(lambda i=i: launch(i)) for i in range(n)], dev)}
        if kid == 3:
            sets = copies(y, f, g, (g * f * dt).contiguous())

            def launch(i):
                (ya, fa, ga, da), yb = sets[i % len(sets)], sets[(i + 1) % len(sets)][0]
                K._raw_milstein_diag(ya, fa, ga, da, dt, spec[i], yb)
            return {"tsde_milstein_diag": _graph_replay_us([(lambda i=i: launch(i)) for i in range(n)], dev)}
        if kid == 4:
            rdt, sqrt_dt = 1.0 / dt, dt ** 0.5
            with torch.no_grad():      # one real step's intermediates as the later stages' operands
                _, acc, p13 = K.srk_diag_stage(2, (y, f, g, f.clone(), g.clone()), dt, rdt, sqrt_dt, spec[0])
            # y0, f0, g0, f1, g1, f2, g2, g3, acc, P: ten operands per step
            sets = copies(y, f, g, f, g, f, g, g, acc, p13)

            def stage(k, i):
                y0, f0, g0, f1, g1, f2, g2, g3, ac, p = sets[i % len(sets)]
                if k == 1:
                    return lambda: K.srk_diag_stage(1, (y0, f0, g0), dt, rdt, sqrt_dt, spec[i])
                if k == 2:
                    return lambda: K.srk_diag_stage(2, (y0, f0, g0, f1, g1), dt, rdt, sqrt_dt, spec[i])
                if k == 3:
                    return lambda: K.srk_diag_stage(3, (p, ac, f2, g2), dt, rdt, sqrt_dt, spec[i])
                return lambda: K.srk_diag_stage(4, (ac, g3), dt, rdt, sqrt_dt, spec[i], out_last=y0)
            return {f"tsde_srk_diag_stage {k}": _graph_replay_us([stage(k, i) for i in range(n)], dev) for k in (1, 2, 3, 4)}
        if kid == 11:
            integrals = torch.randn(B, m, m, device=dev) * dt
            with torch.no_grad():
                support = K.milstein_gf_general_support(y, f, g, dt, dt ** 0.5, True)
                gk = sde.g(t0, support.reshape(m * B, d)).reshape(m, B, d, m).contiguous()
            sets = copies(g, gk, integrals)
            return {"tsde_milstein_gf_general_correction": _graph_replay_us(
                [(lambda i=i: K.milstein_gf_general_correction(*sets[i % len(sets)], dt ** 0.5)) for i in range(50)], dev)}
        if kid == 5:
            params = [p for p in sde.parameters() if p.requires_grad]
            wide = d + 1 if c.get("logqp") else d          # (logqp: the state carries one more column, base_sde.py:240-306)
            base = [torch.rand(B, wide, device=dev) for _ in range(2)] + [torch.randn(B, wide, device=dev) for _ in range(4)]
            sets = copies(*base)
            pst = [[torch.zeros_like(p), torch.zeros_like(p), torch.randn_like(p), torch.randn_like(p)] for p in params]

            def launch(i):
                ya, aa, t0_, t1_, t2_, t3_ = sets[i % len(sets)]
                yb, ab = sets[(i + 1) % len(sets)][:2]
                segs = [dict(out=yb, s=ya, F=t0_, G=t1_, sF=-1.0, sG=-1.0), dict(out=ab, s=aa, F=t2_, G=t3_)]
                segs += [dict(out=q[(i + 1) & 1], s=q[i & 1], F=q[2], G=q[3]) for q in pst]
                K.aug_update(segs, dt, 1.0, torch.float32, dev)
            return {"tsde_aug_update": _graph_replay_us([(lambda i=i: launch(i)) for i in range(n)], dev)}
        return None

    def roofline(self, value, k_ms, k_launches):
        """The `roofline` object of a trajectory-kernel workload given its trajectory-steps/s and kernel brackets."""
        c = self.cfg
        B, d, nsteps = c["B"], c["d"], c["nsteps"]
        if k_launches <= 0:
            return None
        raw_s = k_ms * 1e-3 / k_launches
        if self.trajectory and c.get("mfma_flops_per_traj_step"):
            # 8 solves were timed; a solve is one launch, or (reverse sweep) one launch per chunk of steps
            flops = c["mfma_flops_per_traj_step"] * B * nsteps * 8 / k_launches
            achieved = flops / raw_s / 1e12
            split = (c.get("options") or {}).get("matrix_precision") == "bf16x3"
            # (the opt-in split-bf16 mode runs most of its products on bf16 instructions: its exact-f32-equivalent rate is
            #  reported, but not as a fraction of the f32 peak -- it can exceed it)
            return {"bound": "mfma", "kernel": c["kernel"], "achieved": achieved, "peak": MFMA_F32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": None if split else achieved / MFMA_F32_PEAK_TFLOPS, "traffic": None,
                    "flops_per_launch": flops, "avg_launch_us": raw_s * 1e6, "launches_timed": k_launches,
                    "note": "f32-in / f32-accumulate MFMA (exact f32); peak = dense f32 matrix rate of "
                            "guides/MI355X_MICROARCH.md",
                    "timing": "HIP events bracketing every launch of this kernel in 8 eagerly issued solves"}
        if self.trajectory and c.get("bytes_per_traj_step"):
            # One launch is the whole solve: B x nsteps trajectory-steps with the state in registers, the increments from
            # the counter RNG and f, g evaluated in the kernel. It moves y0 in and the outputs out -- SURVEY 8d's bytes per
            # trajectory-step never reach HBM -- so its roof is the rate at which a SIMD ISSUES vector instructions:
            #   achieved = (issue cycles of the step loop's VALU instructions per wave-step: the loop's instruction histogram
            #               from the compiled kernel x per-instruction issue cycles measured on the device,
            #               tools/valu_model.py + tools/microbench_valu.hip) x wave-steps per launch / launch duration
            #   peak     = 1024 SIMDs x 2.4e9 cycles/s (every SIMD issuing VALU work every cycle at the nominal clock)
            # `hbm_equivalent` keeps the contract's byte-priced number (above the HBM peak by construction).
            per_launch = c["bytes_per_traj_step"] * B * nsteps
            moved = 2 * B * d * 4
            hbm = {"is": "SURVEY 8d bytes per trajectory-step x trajectory-steps / time: what a one-kernel-per-step design "
                         "would stream; not a bound of this kernel",
                   "achieved_GBps": per_launch / raw_s / 1e9, "over_hbm_peak": per_launch / raw_s / 1e9 / HBM_PEAK_GBPS,
                   "bytes_per_traj_step": c["bytes_per_traj_step"], "bytes_per_launch": per_launch,
                   "solve_achieved_GBps": c["bytes_per_traj_step"] * value / self.world / 1e9,
                   "solve_over_hbm_peak": c["bytes_per_traj_step"] * value / self.world / 1e9 / HBM_PEAK_GBPS}
            roof = {"bound": "valu", "kernel": c["kernel"], "traffic": moved,
                    "traffic_is": "HBM bytes one launch really moves: y0 in, final state out",
                    "traffic_over_hbm_equivalent": moved / per_launch, "hbm_equivalent": hbm,
                    "avg_launch_us": raw_s * 1e6, "launches_timed": k_launches, "launches_per_solve": 1,
                    "element_steps_per_s": B * d * nsteps / raw_s,
                    "timing": "HIP events bracketing the single launch of each of 8 eagerly issued solves"}
            model = _valu_model(self.name)
            if model is None:
                roof.update(achieved=None, peak=VALU_ISSUE_PEAK, unit="T SIMD issue-cycles/s", frac=None,
                            frac_is="no VALU model for this kernel instantiation (tools/valu_model.py)")
                return roof
            lanes = B * d // model["elements_per_lane_step"]
            wave_steps = (lanes + 63) // 64 * nsteps
            issue_cycles = model["issue_cycles_per_wave_step"] * wave_steps
            achieved = issue_cycles / raw_s / 1e12
            roof.update(achieved=achieved, peak=VALU_ISSUE_PEAK, unit="T SIMD issue-cycles/s", frac=achieved / VALU_ISSUE_PEAK,
                        frac_is="measured issue cycles of the step loop's VALU instructions x wave-steps per launch / "
                                "(1024 SIMDs x 2.4 GHz x launch time)",
                        wave_steps_per_launch=wave_steps,
                        issue_cycles_per_wave_step=model["issue_cycles_per_wave_step"],
                        valu_instructions_per_wave_step=model["valu_instructions_per_wave_step"],
                        frac_if_every_instruction_took_2_cycles=model["issue_cycles_per_wave_step_all_plain"] * wave_steps
                        / raw_s / 1e12 / VALU_ISSUE_PEAK,
                        model_source=model.get("source"), rates=model.get("rates"))
            _attach_headline_pmc(roof, self.name)
            return roof
        if self.trajectory:
            # One launch per solve: it reads y0 and writes the requested outputs, nothing else touches HBM. The kernel is
            # bound by the VALU work of the counter RNG (Philox-4x32-10 + Box-Muller per element-step) and of f, g.
            bytes_per_launch = 2 * B * d * 4
            achieved = bytes_per_launch / raw_s / 1e9
            return {"bound": "hbm", "kernel": c["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": achieved / HBM_PEAK_GBPS, "traffic": None, "bytes_per_launch": bytes_per_launch,
                    "avg_launch_us": raw_s * 1e6, "launches_timed": k_launches,
                    "element_steps_per_s": B * d * nsteps / raw_s,
                    "note": "VALU-bound by design (state in registers, increments from the counter RNG, f and g evaluated "
                            "in the kernel): the step's 16*d bytes per trajectory-step never reach HBM, so neither "
                            "fraction of the HBM peak describes this kernel; see DESIGN.md for its VALU utilisation",
                    "timing": "HIP events bracketing the single launch of each of 8 eagerly issued solves"}
        raise AssertionError("stepwise workloads use roofline_stepwise")

    def roofline_stepwise(self, value, b2b):
        """The `roofline` object of a stepwise workload from its trajectory-steps/s and the back-to-back kernel times.
        `frac` prices the dominant kernel(s) with SURVEY 8d's ALGORITHMIC bytes per trajectory-step; where the
        implementation moves more than that (SRK: partial sums between the four stage kernels) `moved_frac` says what
        the memory system actually carried."""
        c = self.cfg
        B, per_step = c["B"], c["launches_per_step"]
        contract = c["bytes_per_traj_step"]
        moved = c.get("bytes_moved_per_traj_step", contract)
        if b2b is None:
            return None
        # a step made of several different kernels prices each at its own launch time (`step_kernels`: launches per step)
        counts = c.get("step_kernels")
        if counts is not None:
            if set(counts) - set(b2b):
                return None
            step_us = sum(b2b[k] * n for k, n in counts.items())
            per_step = sum(counts.values())
        else:
            step_us = sum(b2b.values()) if c["kid"] == 4 else next(iter(b2b.values())) * per_step
        achieved = contract * B / (step_us * 1e-6) / 1e9
        solve_achieved = contract * value / self.world / 1e9
        roof = {"bound": "hbm", "kernel": c["kernel"],
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "frac_is": "kernel level: SURVEY 8d bytes of one solver step / duration of the step's dominant-kernel launches",
                "solve_achieved": solve_achieved, "solve_frac": solve_achieved / HBM_PEAK_GBPS,
                "solve_frac_is": "solve level: bytes_per_traj_step x value / (n_gpus x peak), user f, g kernels and gaps inside",
                "bytes_per_traj_step": contract, "bytes_per_launch": contract * B / per_step,
                "launches_per_step": per_step, "avg_launch_us": step_us / per_step,
                "launch_us": {k: round(v, 3) for k, v in b2b.items()},
                "traffic": None,
                "timing": "HIP events around a replayed HIP graph of 200 launches on rotating live operands (>= 128 MiB), "
                          "best of 3"}
        if moved != contract:
            d = c["d"]
            roof["streams"] = {"survey_8d": contract // (4 * d), "implementation": moved // (4 * d),
                               "why": "four stage kernels with user f, g between them: partial sums cross HBM (DESIGN.md)"}
            roof["bytes_moved_per_traj_step"] = moved
            roof["moved_achieved"] = moved * B / (step_us * 1e-6) / 1e9
            roof["moved_frac"] = roof["moved_achieved"] / HBM_PEAK_GBPS
            roof["moved_is"] = ("bytes the implementation's kernels stream per trajectory-step (DESIGN.md: 23 streams for "
                                "SRID2 with user code between the stages vs the 16 of SURVEY 8d)")
        return roof


_VALU_MODELS = {}
_TRAJ_METHOD_CODES = {("euler", "ito"): 0, ("milstein", "ito"): 1, ("milstein", "stratonovich"): 2,
                      ("midpoint", "stratonovich"): 3, ("srk", "ito"): 4, ("heun", "stratonovich"): 5,
                      ("euler_heun", "stratonovich"): 6}                         # include/torchsde_amd.h TSDE_TRAJ_*


def _valu_model(name):
    """The VALU-issue model of a workload's trajectory kernel (tools/valu_model.py) -- the constant-coefficient affine
    kernels with one 16-byte group per lane, i.e. the GBM workloads at their benchmark sizes -- from
    profiles/valu_model_latest.json when that file was made from the kernel sources this run uses, else computed now
    (hipcc -S of trajectory.hip, ~25 s, no GPU involved); None for other kernels or when neither is possible."""
    cfg = WORKLOADS[name]
    if not cfg["problem"].startswith("gbm_") or cfg.get("train") or cfg["B"] * cfg["d"] // 4 < 256 * 8 * 64:
        return None
    sde_type = "stratonovich" if "strat" in cfg["problem"] else "ito"
    code = _TRAJ_METHOD_CODES.get((cfg["method"], sde_type))
    if code is None:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import valu_model
    symbol = valu_model.affine_symbol(code)
    if symbol in _VALU_MODELS:
        return _VALU_MODELS[symbol]
    digest = csrc_digest()
    rec = None
    try:
        with open(os.path.join(ROOT, "profiles", "valu_model_latest.json")) as fh:
            on_file = json.load(fh)
        if on_file.get("csrc_sha") == digest and symbol in on_file.get("kernels", {}):
            rec = dict(on_file["kernels"][symbol], source=f"profiles/valu_model_latest.json @ csrc {digest}")
    except (OSError, ValueError):
        pass
    if rec is None:
        try:
            rec = dict(valu_model.model(symbol), source=f"computed by this run @ csrc {digest}")
        except Exception as e:        # no hipcc on this host, or the loop could not be cut out
            print(f"[bench] no VALU model: {type(e).__name__}: {e}", file=sys.stderr)
    _VALU_MODELS[symbol] = rec
    return rec


def _attach_headline_pmc(roofline, workload):
    """SQ counters of the same kernel from a rocprofv3 --pmc run of this workload (tools/profile_trajectory.sh ->
    profiles/headline_pmc_latest.json), attached only if they were collected on the kernel sources this run uses:
    instructions per wave-step as the hardware counted them, and the share of the kernel's cycles in which the VALU was
    busy -- the counter-side cross-check of the model's fraction."""
    path = os.path.join(ROOT, "profiles", "headline_pmc_latest.json")
    try:
        with open(path) as fh:
            rec = json.load(fh)
    except (OSError, ValueError):
        return
    digest = csrc_digest()
    if rec.get("csrc_sha") != digest and rec.get("workload") == workload and rec.get("files"):
        # other kernels changed since: the counters stand if the files the trajectory kernel is made of did not
        now = csrc_file_digests()
        if all(now.get(f) == rec["files"].get(f) for f in _TRAJECTORY_SOURCES):
            digest = rec["csrc_sha"]
    if rec.get("csrc_sha") != digest or rec.get("workload") != workload:
        roofline["pmc"] = (f"profiles/headline_pmc_latest.json is for {rec.get('workload')} @ csrc {rec.get('csrc_sha')}; this "
                           f"run is {workload} @ {digest}: stale, not reported (re-run tools/profile_trajectory.sh)")
        return
    # (the counters themselves and the formulae are in the file; the line carries what follows from them)
    roofline["pmc"] = {k: (round(rec[k], 4) if isinstance(rec[k], float) else rec[k])
                       for k in ("valu_instructions_per_wave_step", "valu_busy", "effective_clock_ghz", "kernel_avg_us",
                                 "cycles_per_wave_step_per_simd") if k in rec}
    roofline["pmc"]["source"] = f"profiles/headline_pmc_latest.json @ csrc {digest} (rocprofv3 --pmc, tools/profile_trajectory.sh)"


def _graph_replay_us(launches, dev, replays=3):
    """Average microseconds per launch of `launches` (zero-argument callables, one kernel launch each) recorded into
    one HIP graph and replayed: best of `replays`, HIP events around the replay on the replay's stream."""
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side), torch.no_grad():
        for fn in launches[:8]:
            fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    # thread_local: API calls from other threads (the RCCL watchdog of a multi-GPU run) must not invalidate the capture
    with torch.no_grad(), torch.cuda.graph(graph, capture_error_mode="thread_local"):
        for fn in launches:
            fn()
    graph.replay()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(replays):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        graph.replay()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    del graph
    return best * 1e3 / len(launches)


def _attach_offline_traffic(roofline, workload):
    """HBM bytes per launch from rocprofv3 PMC passes of `bench.py --workload <this one>` (tools/profile_traffic.sh),
    corrected as guides/MI355X_MICROARCH.md prescribes -- collected offline because counters need their own passes,
    and attached only if the file says it was collected on the kernel sources this run uses."""
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    cfg = WORKLOADS[workload]
    if roofline is None or not cfg.get("kernel_match") or not os.path.exists(tpath):
        return
    try:
        with open(tpath) as fh:
            rec = json.load(fh)
        digest = csrc_digest()
        if rec.get("csrc_sha") != digest:
            # other kernels changed since: the counters still stand if the files THESE kernels are made of did not
            now, then = csrc_file_digests(), rec.get("files") or {}
            needed = _TRAJECTORY_SOURCES if cfg.get("trajectory") else _STEPWISE_SOURCES
            changed = [f for f in needed if now.get(f) != then.get(f)]
            if changed:
                roofline["traffic_source"] = (f"profiles/traffic_latest.json is for kernel sources {rec.get('csrc_sha')}, this "
                                              f"run uses {digest} and {', '.join(changed)} changed since: stale, not reported "
                                              "(re-run tools/profile_traffic.sh)")
                return
            digest = f"{rec.get('csrc_sha')} (this run: {digest}; the files of the measured kernels are unchanged)"
        kernels = rec.get("workloads", {}).get(workload, {}).get("kernels")
        if kernels is None and workload == HEADLINE:
            kernels = rec.get("kernels")          # (the one-workload layout of rounds 1-2)
        if not kernels:
            roofline["traffic_source"] = f"profiles/traffic_latest.json has no counters for {workload}"
            return
        # one row per dominant kernel of a step (one, or the four SRK stages): every pattern must find its kernel
        rows = []
        for pattern in cfg["kernel_match"]:
            hit = [k for name, k in kernels.items() if pattern in name]
            if not hit:
                roofline["traffic_source"] = f"profiles/traffic_latest.json: no kernel matching {pattern!r} for {workload}"
                return
            rows.append(max(hit, key=lambda k: k.get("launches", 0)))
        if cfg.get("trajectory"):
            # one launch is the whole solve: the counters give what it really moves (y0 in, outputs out)
            roofline["traffic"] = rows[0]["traffic_bytes_per_launch"]
            roofline["traffic_is"] = "HBM bytes per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)"
            roofline["traffic_over_algorithmic"] = roofline["traffic"] / roofline["bytes_per_launch"]
            roofline["traffic_source"] = f"profiles/traffic_latest.json @ csrc {digest}"
            return
        per_step = cfg["launches_per_step"]
        launches_each = per_step // len(rows)
        step_bytes = sum(k["traffic_bytes_per_launch"] for k in rows) * launches_each
        roofline["traffic"] = step_bytes / per_step
        roofline["traffic_per_traj_step"] = step_bytes / cfg["B"]
        roofline["traffic_over_algorithmic"] = roofline["traffic_per_traj_step"] / cfg["bytes_per_traj_step"]
        us = [k.get("kernel_avg_us") for k in rows]
        if all(u is not None for u in us):
            roofline["kernel_us_rocprofv3"] = sum(us) / len(us)
            roofline["frac_rocprofv3"] = (cfg["bytes_per_traj_step"] * cfg["B"] / (sum(us) * launches_each * 1e-6) / 1e9
                                          / HBM_PEAK_GBPS)
        roofline["traffic_source"] = (f"profiles/traffic_latest.json @ csrc {digest}: separate rocprofv3 --pmc FETCH_SIZE / "
                                      f"WRITE_SIZE passes, bytes = (2 x FETCH + WRITE) x 1024")
    except Exception as e:
        roofline["traffic_source"] = f"profiles/traffic_latest.json unreadable: {e}"


def _side_measurement(dev, name):
    """One `also` entry: 2 warm-up solves, 5 timed solves (median), then the dominant kernel timed back to back (the
    headline's method) with the counter traffic of profiles/traffic_latest.json attached."""
    job = Job(name, dev)
    full_steps = job.cfg["nsteps"]
    budget = job.cfg.get("bench_steps")
    if budget:          # a workload too slow to time whole: a fixed number of its (homogeneous) solver steps, extrapolated
        job.cfg = dict(job.cfg, nsteps=budget)
        job.ts = torch.tensor([0.0, budget * job.cfg["dt"]], device=dev)
    c = job.cfg
    job.prepare()
    for i in range(2):
        job.solve(i)
    torch.cuda.synchronize()
    times = []
    for i in range(5):
        start = time.perf_counter()
        out = job.solve(10 + i)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - start) * 1e3)
    assert torch.isfinite(out).all()
    ms = statistics.median(times)
    value = c["B"] * c["nsteps"] / ms * 1e3
    rec = {"ms_per_solve": ms, "ms_per_solve_all": times, "trajectory_steps_per_s": value, "kernel": c["kernel"],
           "launch": "one trajectory-kernel launch per solve" if job.trajectory else
                     "HIP graph replay" if job.use_graph else "eager launches"}
    if budget:
        scale = full_steps / budget
        rec.update(ms_per_solve=ms * scale, ms_per_solve_all=[t * scale for t in times],
                   extrapolated=f"timed {budget} of the {full_steps} solver steps per solve (ms as measured: {ms:.2f}), "
                                f"x{scale:g}: the stepping loop is homogeneous")
    if job.train or job.adjoint:
        rec["what"] = "forward + backward per solve"
    if job.use_graph:
        from torchsde_amd import graph as graph_module
        rec["graphs"] = [line[:240] for line in graph_module.describe_cache(job.sde)]
    if job.trajectory:
        k_ms, k_launches = job.bracket_dominant_kernel()
        roof = job.roofline(value, k_ms, k_launches)
    else:
        roof = job.roofline_stepwise(value, job.back_to_back_us(job.live_state(out)))
        _attach_offline_traffic(roof, name)
    if roof is not None:
        for key in ("bound", "achieved", "unit", "frac", "solve_achieved", "solve_frac", "bytes_per_traj_step",
                    "bytes_per_launch", "flops_per_launch", "avg_launch_us", "launches_timed", "launch_us",
                    "launches_per_step", "bytes_moved_per_traj_step", "moved_frac", "streams", "traffic", "traffic_per_traj_step",
                    "traffic_over_algorithmic", "kernel_us_rocprofv3", "frac_rocprofv3", "traffic_source", "timing"):
            if key in roof:
                rec[("kernel_" + key) if key in ("achieved", "frac") else key] = roof[key]
        if roof["bound"] == "mfma":
            rec["tflops_f32"] = roof["achieved"]
    return rec


def _stepwise_beside(dev, name, args):
    """The same job on the STEPWISE route (options={"trajectory_kernel": False}: the user's f, g as torch kernels between
    the per-step kernels, HIP-graph replay), reported inside a recognised headline: its trajectory-steps/s over the same
    number of solves, the per-step kernel's HBM fraction (kernel level) and the solve-level fraction of SURVEY 8d."""
    job = Job(name, dev, graph=not args.eager)
    c = job.cfg
    job.prepare()
    for i in range(args.warmup):
        job.solve(i)
    torch.cuda.synchronize()
    start = time.perf_counter()
    for i in range(args.steps):
        out = job.solve(1000 + i)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - start
    value = c["B"] * c["nsteps"] * args.steps / elapsed
    roof = job.roofline_stepwise(value, job.back_to_back_us(job.live_state(out)))
    _attach_offline_traffic(roof, name)
    keep = ("kernel", "achieved", "frac", "solve_achieved", "solve_frac", "bytes_per_traj_step", "bytes_per_launch",
            "avg_launch_us", "launches_per_step", "traffic", "traffic_over_algorithmic", "kernel_us_rocprofv3",
            "frac_rocprofv3")
    return {"workload": name, "value": value, "ms_per_step": elapsed / args.steps * 1e3,
            "route": "options={'trajectory_kernel': False}: user f, g torch kernels + one fused step kernel per step, "
                     "HIP graph replay",
            "roofline": {k: roof[k] for k in keep if k in roof}}


CONFIG3 = "c4_midpoint_diag_default_route_b32768_d64"


def _config3_beside(dev, rank, world, dist, args):
    """BASELINE configs[3] -- Stratonovich midpoint, 262144 x 64 rows sharded 32768 per GPU over 8 GPUs, one RCCL gather of
    final states per solve -- measured by EVERY multi-GPU run of the default bench line, beside the headline (VERDICT r5 next 8:
    the driver's scale run passes no --workload). Same timing contract as the headline (barrier + synchronize on both sides,
    max over ranks). Also checks, once, that the rows rank 0 GATHERED from another rank's shard are bit-identical to a
    single-rank solve of those global rows on rank 0 (the increments are addressed by global row: sharding.py)."""
    job = Job(CONFIG3, dev, rank=rank, world=world, dist=dist, graph=not args.eager)
    c = job.cfg
    job.prepare()

    def barrier():
        job.finish()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
    for i in range(args.warmup):
        job.solve(i)
    barrier()
    start = time.perf_counter()
    for i in range(args.steps):
        out = job.solve(1000 + i)
    barrier()
    elapsed = torch.tensor([time.perf_counter() - start], device=dev, dtype=torch.float64)
    dist.all_reduce(elapsed, op=dist.ReduceOp.MAX)
    elapsed = elapsed.item()
    # bit-equality of one foreign shard: the last rank's rows, solved again here as a single-rank job with its row offset
    other = world - 1
    gathered = job.solve(77)
    barrier()
    shard = gathered[other * c["B"]:(other + 1) * c["B"]].clone()
    alone = Job(CONFIG3, dev, rank=other, world=world, dist=None, graph=not args.eager)
    alone.sde = job.sde
    for i in range(2):
        alone.solve(9000 + i)                   # (the module's route is already trusted; warm the plan)
    same = bool(torch.equal(alone.solve(77), shard))
    flag = torch.tensor([1 if same else 0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    ranks = _what_the_ranks_saw(job, dev, dist, os.environ.get("TSDE_BENCH_SHARE_GPU") == "1")
    value = world * c["B"] * c["nsteps"] * args.steps / elapsed
    return {"workload": CONFIG3, "is": "BASELINE configs[3]: Stratonovich midpoint, diagonal noise, batch sharded "
            f"{c['B']} rows per GPU", "value": value, "unit": "trajectory-steps/s", "n_gpus": world,
            "global_batch": world * c["B"], "state": c["d"], "solver_steps": c["nsteps"],
            "ms_per_step": elapsed / args.steps * 1e3, "scaling": "weak",
            "gathered_shard_equals_single_rank_solve": bool(flag.item()), "shard_checked": f"rank {other}'s rows on every rank",
            "ranks_seen": ranks["ranks_seen"], "all_gather_ms_per_solve": ranks["all_gather_ms_per_solve"],
            "all_gather_bytes_per_rank": ranks.get("all_gather_bytes_per_rank"),
            "collective_backend": ranks["collective_backend"]}


def _side_measurements(dev, budget=0.0):
    """Short measurements reported under `also`: outside the headline's timed region and NOT part of `value`. A
    failure here is reported in place and never takes the headline down with it."""
    also = {}
    began = time.time()
    for name in ALSO:
        if name not in WORKLOADS:
            continue
        if budget > 0 and time.time() - began > budget:
            also[name] = {"skipped": f"the side measurements' time budget ({budget:.0f} s, --also-budget) was used up; "
                                     f"python bench.py --workload {name}"}
            continue
        try:
            also[name] = _side_measurement(dev, name)
        except Exception as e:
            also[name] = {"error": f"{type(e).__name__}: {e}"}
        torch.cuda.empty_cache()
    return also


HEADLINE_LIMIT = 4096      # the driver keeps only the tail of stdout: the headline is the LAST line and stays under this
ALSO_LINE_LIMIT = 2048
_ALSO_PROSE = ("timing", "traffic_source", "ms_per_solve_all", "graphs", "extrapolated", "frac_is", "solve_frac_is")


def _emit_also(also):
    """The side measurements go to `bench_also.json` (whole records; next to this file, and under gpurun_out/ when that
    exists) and to stdout as one short line per workload BEFORE the headline -- never inside it."""
    paths = [os.path.join(ROOT, "bench_also.json")]
    if os.path.isdir(os.path.join(ROOT, "gpurun_out")):
        paths.append(os.path.join(ROOT, "gpurun_out", "bench_also.json"))
    written = None
    for path in paths:
        try:
            with open(path, "w") as fh:
                json.dump({"csrc_sha": csrc_digest(), "also": also}, fh, indent=1)
            written = written or os.path.relpath(path, ROOT)
        except OSError:
            pass
    for name, rec in also.items():
        short = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in rec.items() if k not in _ALSO_PROSE}
        text = json.dumps({"also": name, **short})
        if len(text) > ALSO_LINE_LIMIT:
            text = json.dumps({"also": name, **{k: short[k] for k in ("ms_per_solve", "trajectory_steps_per_s", "kernel",
                                                                      "kernel_frac", "solve_frac", "error") if k in short}})
        print(text)
    return written


def _print_headline(line):
    """The LAST stdout line: the headline with `roofline` and `cpu_baseline`, under HEADLINE_LIMIT bytes. Prose is
    dropped before numbers if a future field ever pushes it over."""
    text = json.dumps(line)
    for key in ("timing", "traffic_source", "solve_frac_is", "frac_is", "moved_is", "note"):
        if len(text) <= HEADLINE_LIMIT:
            break
        if isinstance(line.get("roofline"), dict):
            line["roofline"].pop(key, None)
        text = json.dumps(line)
    if len(text) > HEADLINE_LIMIT and isinstance(line.get("cpu_baseline"), dict):
        line["cpu_baseline"]["sample"] = line["cpu_baseline"].get("sample", "")[:160]
        text = json.dumps(line)
    sys.stdout.flush()
    print(text, flush=True)


def _what_the_ranks_saw(job, dev, dist, share_gpu):
    """What a reader needs to believe an N-rank line: how many ranks the process group really had, which device each
    one ran on, the backend, and what the one collective of a solve (the all-gather of final states) costs by itself."""
    name = torch.cuda.get_device_name(dev)
    if dist is None:
        return {"ranks_seen": 1, "rank_devices": [f"rank 0: cuda:{dev.index} {name}"], "collective_backend": None,
                "all_gather_ms_per_solve": None}
    seen = [None] * dist.get_world_size()
    dist.all_gather_object(seen, f"rank {dist.get_rank()}: cuda:{dev.index} {name} (pid {os.getpid()})")
    gather_ms = None
    if job.gathered is not None and not (job.adjoint or job.train):
        local = torch.zeros((job.cfg["B"], job.cfg["d"]), device=dev)
        for _ in range(3):
            dist.all_gather_into_tensor(job.gathered[0], local)
        dist.barrier()
        torch.cuda.synchronize()
        start = time.perf_counter()
        for _ in range(20):
            dist.all_gather_into_tensor(job.gathered[0], local)
        torch.cuda.synchronize()
        t = torch.tensor([(time.perf_counter() - start) / 20 * 1e3], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        gather_ms = t.item()
    return {"ranks_seen": dist.get_world_size(), "rank_devices": seen,
            "collective_backend": dist.get_backend() + (" (TSDE_BENCH_SHARE_GPU: every rank on device 0)" if share_gpu
                                                        else " (RCCL)" if dist.get_backend() == "nccl" else ""),
            "all_gather_ms_per_solve": gather_ms,
            "all_gather_bytes_per_rank": job.cfg["B"] * job.cfg["d"] * 4 if gather_ms is not None else None}


def _self_launch(n_gpus, share_gpu):
    """`python bench.py --gpus N` with no launcher around it: start the N ranks here, exactly as the driver's
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py ...`
    would, and hand back its exit code. Refuses to start when the node has fewer than N GPUs."""
    import socket
    import subprocess
    have = torch.cuda.device_count()
    if not share_gpu and have < n_gpus:
        print(f"bench.py: --gpus {n_gpus} asked for, this node shows {have} GPU(s): not starting "
              f"(one rank per GPU; RCCL refuses two ranks on one device)", file=sys.stderr)
        return 3
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)      # (a solve is ~3 ms on the default route: 40 give a stable figure)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--workload", default=HEADLINE, choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-also", action="store_true", help="skip the short side measurements reported under `also`")
    ap.add_argument("--also-budget", type=float, default=100.0,
                    help="seconds the side measurements may take in all (0 = no limit); the rest is recorded as skipped")
    ap.add_argument("--no-stepwise", action="store_true",
                    help="skip the stepwise solve of the same workload reported beside a recognised headline")
    ap.add_argument("--eager", action="store_true", help="issue every solve eagerly instead of replaying a HIP graph")
    ap.add_argument("--profile-steps", type=int, default=0,
                    help="tools/profile_traffic.sh only: solve this many solver steps of the workload and print a line "
                         "marked profile_run (per-launch counters do not need the full 1000 steps); NOT a measurement")
    args = ap.parse_args()

    # TSDE_BENCH_SHARE_GPU=1 (tests only): all ranks use device 0 and gloo carries the collectives, so that the
    # multi-rank logic of this file can be exercised on a one-GPU box (RCCL refuses two ranks on one device)
    share_gpu = os.environ.get("TSDE_BENCH_SHARE_GPU") == "1"
    # dmabuf IPC is what RCCL needs on this host driver; read by the HSA runtime when the first rank touches its GPU
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: this process becomes the launcher of N ranks (one per GPU, RCCL)
        raise SystemExit(_self_launch(args.gpus, share_gpu))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if not share_gpu and torch.cuda.device_count() < local_rank + 1:
        raise SystemExit(f"bench.py: rank {rank} needs GPU {local_rank}, this node shows {torch.cuda.device_count()}")
    device_index = 0 if share_gpu else local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    # launched by torch.distributed.run (also with a single rank): use the collective path, so that the very same
    # code runs at N = 1, 2, 4, 8
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    dist = None
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)

    job = Job(args.workload, dev, rank=rank, world=world, dist=dist, graph=not args.eager)
    if args.profile_steps > 0:
        # (counters serialise every kernel: skip the capture-time tuning that replays whole solves several times)
        job.cfg = dict(job.cfg, nsteps=args.profile_steps, options=dict(job.cfg.get("options") or {}, overlap_f_g=False))
        job.ts = torch.tensor([0.0, args.profile_steps * job.cfg["dt"]], device=dev)
        for i in range(args.warmup + args.steps):
            job.solve(i)
        torch.cuda.synchronize()
        print(json.dumps({"profile_run": True, "workload": args.workload, "solver_steps": args.profile_steps,
                          "solves": args.warmup + args.steps, "csrc_sha": csrc_digest()}))
        return
    cfg = job.cfg
    B, d, m, nsteps, dt = cfg["B"], cfg["d"], cfg["m"], cfg["nsteps"], cfg["dt"]
    job.prepare()

    def barrier():
        if use_dist:
            job.finish()
            torch.cuda.synchronize()
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        job.solve(i)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    barrier()
    t_start = time.perf_counter()
    for i in range(args.steps):
        marks[i].record()
        out = job.solve(1000 + i)
    marks[args.steps].record()
    barrier()
    elapsed = time.perf_counter() - t_start
    per_solve_ms = [marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps)]
    median_ms = statistics.median(per_solve_ms)
    if use_dist:
        t = torch.tensor([elapsed, median_ms], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, median_ms = t[0].item(), t[1].item()
    assert torch.isfinite(out).all()
    value = world * B * nsteps * args.steps / elapsed
    ranks = _what_the_ranks_saw(job, dev, dist if use_dist else None, share_gpu)

    if job.trajectory:
        k_ms, k_launches = job.bracket_dominant_kernel()
        roofline = job.roofline(value, k_ms, k_launches)
        if roofline is not None and roofline.get("bound") == "hbm" and cfg.get("kernel_match"):
            _attach_offline_traffic(roofline, args.workload)
    else:
        roofline = job.roofline_stepwise(value, job.back_to_back_us(job.live_state(out)))
        _attach_offline_traffic(roofline, args.workload)

    config3 = None
    if world > 1 and args.workload == HEADLINE and use_dist:
        try:
            config3 = _config3_beside(dev, rank, world, dist, args)
        except Exception as e:           # (never takes the headline down with it -- but every rank must fail alike)
            config3 = {"workload": CONFIG3, "error": f"{type(e).__name__}: {e}"}
    stepwise = None
    if world == 1 and cfg.get("stepwise") and not args.no_stepwise:
        stepwise = _stepwise_beside(dev, cfg["stepwise"], args)
    also = None
    if world == 1 and not args.no_also and args.workload == HEADLINE:
        del job
        torch.cuda.empty_cache()
        also = _side_measurements(dev, args.also_budget)
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = _cpu_baseline(cfg)
        trajectory, adjoint, train = cfg.get("trajectory", False), cfg.get("adjoint", False), cfg.get("train", False)
        line = {
            "metric": "SDE steps/sec (batch x timesteps / sec)", "value": value, "unit": "trajectory-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "median_ms_per_step": median_ms,
            "value_median": world * B * nsteps / median_ms * 1e3,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload,
                       "sde": cfg["problem"] + (" (UNCHANGED user module, f, g plain torch code; recognised as per-channel "
                                                "expressions at every solve and evaluated in the kernel)"
                                                if cfg.get("recognised") else
                                                " (closed-form coefficients; f, g evaluated in the kernel)" if trajectory
                                                else " (f, g are user torch ops)"),
                       "method": cfg["method"] + ("+adjoint:" + cfg["adjoint_method"] if adjoint else "") +
                                 (" + loss.backward() through the solver" if train else ""),
                       "batch_per_gpu": B, "global_batch": world * B, "state": d, "brownian_channels": m,
                       "solver_steps": nsteps, "dt": dt, "brownian": "counter-RNG, generated in the step kernel",
                       "launch": ("trajectory kernels: one forward launch, the reverse sweep in chunks of steps, two "
                                  "weight-gradient products per chunk" if train else
                                  "one trajectory-kernel launch per solve" if trajectory else
                                  "HIP graph replay of the whole solve" + (" and of the backward sweep" if adjoint else "")
                                  if not args.eager else "eager launches"),
                       "parallelism": f"batch-sharded x{world}, one all_gather of final states per solve",
                       "csrc_sha": csrc_digest()},
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if stepwise is not None:
            line["stepwise"] = stepwise
        if config3 is not None:
            line["configs3"] = config3
        line.update(ranks)
        if also is not None:
            line["also_file"] = _emit_also(also)
        _print_headline(line)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Same-thread-count A/B of the CPU baseline bench.py times (`cpu_baseline.kind = "port"`: oracle/brownian_ref.py +
oracle/solvers_ref.py) against the REAL reference package (/root/reference, dev container only), on the headline
workload (B 65536 x d 64, Euler, dt = 2^-10 with the dt hint): per-step time of each over the same number of steps,
and that both produce the same numbers (the port reproduces the reference's Brownian sequences bit for bit).

    python tools/ab_port_vs_reference.py [threads ...]  > profiles/r2_cpu_port_vs_reference.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden", "_ref_shim"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

import torch  # noqa: E402
import torchsde  # noqa: E402  (the reference)

from oracle import brownian_ref, solvers_ref  # noqa: E402
from workloads import configs  # noqa: E402

cfg = configs.WORKLOADS["c2_euler_diag_b65536_d64_s1000"]
B, d, dt, nsteps = cfg["B"], cfg["d"], cfg["dt"], cfg["nsteps"]
STEPS = 24
sde = configs.make_problem(cfg["problem"], d, d, "cpu")
y0 = torch.full((B, d), 0.1)
t1 = nsteps * dt


def run_port():
    bm = brownian_ref.BrownianIntervalRef(t0=0.0, t1=t1, size=(B, d), dtype=torch.float32, entropy=20240601, dt=dt,
                                          levy_area_approximation="none")
    y, t = y0, torch.tensor(0.0)
    start = time.perf_counter()
    with torch.no_grad():
        for _ in range(STEPS):
            y = solvers_ref.euler_step(sde, bm, t, t + dt, y)
            t = t + dt
    return time.perf_counter() - start, y


def run_reference():
    bm = torchsde.BrownianInterval(t0=0.0, t1=t1, size=(B, d), dtype=torch.float32, entropy=20240601, dt=dt,
                                   levy_area_approximation="none")
    ts = torch.tensor([0.0, STEPS * dt])
    start = time.perf_counter()
    with torch.no_grad():
        ys = torchsde.sdeint(sde, y0, ts, bm=bm, method="euler", dt=dt)
    return time.perf_counter() - start, ys[-1]


print(f"command: python tools/ab_port_vs_reference.py {' '.join(sys.argv[1:])}")
print(f"host: {os.cpu_count()} logical CPUs; torch {torch.__version__}; workload B={B} d={d} Euler dt=2^-10, {STEPS} steps, "
      f"BrownianInterval(t1={t1}, dt hint) as bench.py's cpu_baseline builds it")
threads = [int(a) for a in sys.argv[1:]] or [os.cpu_count()]
for n in threads:
    torch.set_num_threads(n)
    run_port(), run_reference()                      # warm-up
    best_p, best_r = float("inf"), float("inf")
    for _ in range(3):
        tp, yp = run_port()
        tr, yr = run_reference()
        best_p, best_r = min(best_p, tp), min(best_r, tr)
    same = torch.equal(yp, yr)
    print(f"threads={n:3d}  port {best_p / STEPS * 1e3:7.2f} ms/step = {B * STEPS / best_p:.3e} traj-steps/s   "
          f"reference {best_r / STEPS * 1e3:7.2f} ms/step = {B * STEPS / best_r:.3e} traj-steps/s   "
          f"port/reference time = {best_p / best_r:.3f}   final states bit-identical: {same}")

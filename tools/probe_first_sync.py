"""Which call synchronises with the host during the FIRST screened solve of a process? Prints every sync-debug warning of
three consecutive eager solves with the Python stack that raised it."""
import os
import sys
import traceback
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import torchsde_amd  # noqa: E402
from torchsde_amd import solvers  # noqa: E402
from workloads import problems  # noqa: E402

dev = "cuda"
sde = problems.make("gbm_ito", d=8).to(dev)
y0 = torch.full((128, 8), 0.1, device=dev)
ts = torch.tensor([0.0, 5 / 64, 16 / 64], device=dev)
real_run = solvers.BaseSDESolver._run


def watched_run(self, plan, y):
    def show(message, category, filename, lineno, file=None, line=None):
        if "synchronizing" in str(message):
            print("SYNC inside _run:", str(message)[:100])
            print("".join(traceback.format_stack(limit=14)[:-1]))
    old = warnings.showwarning
    warnings.showwarning = show
    torch.cuda.set_sync_debug_mode("warn")
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("always")
            return real_run(self, plan, y)
    finally:
        torch.cuda.set_sync_debug_mode(0)
        warnings.showwarning = old


solvers.BaseSDESolver._run = watched_run
for k in range(3):
    print("== solve", k)
    bm = torchsde_amd.BrownianInterval(0.0, 16 / 64, size=(128, 8), device=dev, dtype=torch.float32, entropy=k)
    with torch.no_grad():
        torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="euler", dt=1 / 64, options={"hip_graph": False})
    torch.cuda.synchronize()
print("done")

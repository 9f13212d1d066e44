"""One adaptive solve at the headline shape (65536 x 64 GBM, Milstein, 4 output times) for an API / kernel trace:

    rocprofv3 --hip-trace --kernel-trace --stats -d <dir> -- python tools/prof_adaptive.py device|host

`device`: accept / reject decided by the controller kernel, the host reads the state back once per round of attempts;
`host`: the reference's structure, one read-back (`.item()`) per attempted step. The HIP API statistics of the two runs
differ in the number of device->host copies / stream synchronisations, the kernel statistics in the controller,
commit and merge kernels."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchsde_amd  # noqa: E402
from torchsde_amd import adaptive  # noqa: E402
from workloads import problems  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "device"
dev = "cuda"
B, d = 65536, 64
sde = problems.make("gbm_ito", d=d).to(dev)
y0 = torch.full((B, d), 0.1, device=dev)
ts = torch.tensor([0.0, 0.25, 0.5, 0.75, 1.0], device=dev)
for i in range(3):
    bm = torchsde_amd.BrownianInterval(0.0, 1.0, size=(B, d), device=dev, dtype=torch.float32, entropy=5 + i)
    with torch.no_grad():
        ys = torchsde_amd.sdeint(sde, y0, ts, bm=bm, method="milstein", dt=0.05, adaptive=True, rtol=1e-3, atol=1e-4,
                                 options={"device_adaptive": mode == "device"})
    torch.cuda.synchronize()
print(mode, "control; last solve:", adaptive.last_stats if mode == "device" else "one sync per attempt", float(ys[-1].mean()))

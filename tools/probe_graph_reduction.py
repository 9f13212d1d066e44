"""Is a torch reduction recorded in a HIP graph idempotent across replays on this stack? (Multi-block reductions keep
semaphores that are reset by a hipMemsetAsync per launch.) Prints max |graph - eager| per replay for a few shapes."""
import torch

dev = "cuda"
torch.manual_seed(0)
for shape, dim in (((4096, 128), 0), ((32768, 128), 0), ((32768, 128), 1), ((1 << 20,), 0), ((4096, 128, 16), 0)):
    x = torch.randn(*shape, device=dev)
    want = x.sum(dim)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        x.sum(dim)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = (x * 1.0).sum(dim)
    errs = []
    for k in range(4):
        g.replay()
        torch.cuda.synchronize()
        errs.append((y - want).abs().max().item())
    print(shape, "sum over dim", dim, "| max |graph - eager| per replay:", " ".join(f"{e:.3e}" for e in errs),
          "| scale", f"{want.abs().max().item():.2e}")

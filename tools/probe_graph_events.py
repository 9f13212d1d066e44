import torch
x = torch.rand(1 << 24, device="cuda")
y = torch.empty_like(x)
e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    y.copy_(x)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        e[0].record()
        y.copy_(x * 2)
        e[1].record()
        y.add_(1)
        e[2].record()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("captured events ok:", e[0].elapsed_time(e[1]) * 1e3, "us,", e[1].elapsed_time(e[2]) * 1e3, "us")
except Exception as ex:
    print("FAILED:", type(ex).__name__, ex)

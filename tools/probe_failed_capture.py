"""What a FAILED HIP-graph capture leaves behind on this stack, and whether restoring the caller's stream is enough to go
on (graph._capturing). Run on the GPU box; prints a verdict per step. Each scenario runs in its own process."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SCENARIOS = {
    "pinv_in_capture": "torch.linalg.pinv(torch.rand(8, 4, 4, device='cuda'))",
    "item_in_capture": "float(torch.rand(4, device='cuda').sum())",
    "python_exception_in_capture": "raise ValueError('user code failed')",
}


def child(name):
    import torch
    from torchsde_amd import graph
    dev = torch.device("cuda", 0)
    x = torch.rand(1024, device=dev)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    try:
        with graph._capturing(g, dev):
            y = x * 2
            exec(SCENARIOS[name])
        print(name, ": capture did NOT fail")
    except BaseException as e:
        print(name, ": capture failed with", type(e).__name__, str(e).splitlines()[0][:120])
    try:
        print("  capturing flag after failure:", torch.cuda.is_current_stream_capturing())
        z = (x + 1).sum().item()
        print("  eager work after the failure: ok", round(z, 2))
    except BaseException as e:
        print("  eager work after the failure: BROKEN --", type(e).__name__, str(e).splitlines()[0][:160])
        return
    try:
        g2 = torch.cuda.CUDAGraph()
        with graph._capturing(g2, dev):
            w = x * 3
        g2.replay()
        torch.cuda.synchronize()
        print("  a new capture + replay after the failure: ok", bool(torch.allclose(w, x * 3)))
    except BaseException as e:
        print("  a new capture after the failure: BROKEN --", type(e).__name__, str(e).splitlines()[0][:160])


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for name in SCENARIOS:
            out = subprocess.run([sys.executable, os.path.abspath(__file__), name], capture_output=True, text=True, timeout=300)
            print(out.stdout.strip() or "(no output)")
            if out.returncode != 0:
                print("  process exit code", out.returncode, "|", out.stderr.strip().splitlines()[-1][:200] if out.stderr.strip() else "")

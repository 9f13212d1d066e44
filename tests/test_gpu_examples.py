"""The scripts under examples/ run (run with ``-m gpu``)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("argv", [
    ["quickstart.py"],
    ["train_latent_sde.py", "--steps", "4"],
    ["train_latent_sde.py", "--steps", "4", "--adjoint"],
    ["monte_carlo_closed_form.py"],
    ["sample_neural_sde.py"],
    ["train_neural_sde.py", "--iters", "12"],
    ["neural_general_sde.py"],
    ["scalar_noise_training.py"],
    ["additive_noise_sde.py"],
    ["train_reversible_heun.py", "--iters", "6"],
])
def test_example_runs(argv):
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "examples", argv[0])] + argv[1:], capture_output=True,
                          text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert proc.stdout.strip()


def test_c_abi_demo_builds_and_runs(tmp_path):
    """examples/c_abi_gbm.cpp: the boundary driven from plain C++ (no Python, no torch) -- a GBM solve step by step and
    as one trajectory launch, bit-identical, sample mean against the closed form."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    lib_dir = os.path.join(ROOT, "torchsde_amd", "csrc")
    exe = str(tmp_path / "c_abi_gbm")
    build = subprocess.run([hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                            os.path.join(ROOT, "examples", "c_abi_gbm.cpp"), "-L" + lib_dir, "-ltorchsde_amd",
                            "-Wl,-rpath," + lib_dir, "-o", exe], capture_output=True, text=True, timeout=300)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr[-2000:]
    assert "bit-identical" in run.stdout

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
])
def test_example_runs(argv):
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "examples", argv[0])] + argv[1:], capture_output=True,
                          text=True, timeout=300)
    assert proc.returncode == 0, proc.stderr[-2000:]
    assert proc.stdout.strip()

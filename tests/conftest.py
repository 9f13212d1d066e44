import os
import sys

import pytest

# Program kernels compiled at run time (torchsde_amd/specialise.py) are exercised by tests/test_gpu_specialise.py, which
# switches them on; everywhere else the suite pins the interpreter (hundreds of different programs pass through it here, and
# each would start a 4 s compilation).
os.environ.setdefault("TSDE_SPECIALISE", "0")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a ROCm GPU (MI355X); run with `-m gpu` on the GPU box")


@pytest.fixture(scope="session", autouse=True)
def _built_libraries():
    """Make sure the HIP library (cross-compiled, no GPU needed) and the oracle's C twin exist."""
    from torchsde_amd import _native
    if not _native.is_built():
        import __graft_entry__
        __graft_entry__.build()
    from oracle import build as oracle_build
    oracle_build.build()
    yield


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests need a ROCm device: on a host without one they are skipped, not failed (a plain `pytest tests`
    on the dev container runs the CPU suite and reports the GPU suite as skipped)."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a ROCm GPU (run on the MI355X box with -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)

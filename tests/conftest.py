import os
import socket
import sys

import pytest
import torch  # noqa: F401  -- BEFORE libboundmpc_hip.so is loaded: torch brings its own HIP runtime, and the GPU tests
#                              that hand torch tensors to the C ABI need both to share the one that was loaded first

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_missing():
    """Reason why gpu-marked tests cannot run here, or None."""
    if not os.path.exists(os.path.join(ROOT, "boundplanner_amd", "csrc", "libboundmpc_hip.so")):
        return "libboundmpc_hip.so is not built (python -c 'import __graft_entry__ as g; g.build()')"
    import bench
    if bench.visible_gpus() < 1:            # KFD topology, not the HIP runtime: pytest itself never initialises the GPU here
        return "no HIP device visible"
    return None


def pytest_collection_modifyitems(config, items):
    """A plain `pytest` on a host without a GPU skips the gpu-marked tests instead of erroring in HipBoundMPC().
    With `-m gpu` (the GPU box) nothing is skipped: a missing library or device must fail loudly there."""
    if "gpu" in (config.getoption("-m") or ""):
        return
    why = _gpu_missing()
    if why is None:
        return
    skip = pytest.mark.skip(reason=why)
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]

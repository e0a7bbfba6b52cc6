import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: GPU tests that also run the CPU oracle's backward pass at full size (minutes)")


@pytest.fixture(scope="session")
def built_lib():
    """Make sure libhific_hip.so exists (cross-compiles on CPU-only hosts)."""
    import __graft_entry__ as g
    so = os.path.join(g.PKG, "libhific_hip.so")
    if not os.path.exists(so) or not os.path.exists(os.path.join(g.PKG, "libhific_host.so")):
        g.build()
    return so


@pytest.fixture(scope="session", autouse=True)
def _libs_built(built_lib):
    """Every test module may assume both shared libraries exist (fresh clones have neither: *.so is git-ignored)."""
    return built_lib


@pytest.fixture(scope="session")
def hific(built_lib):
    import hific_amd
    return hific_amd


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def pytest_collection_modifyitems(config, items):
    # the reference's own codec emits NumPy deprecation warnings by the hundred when it is run as the pin
    for item in items:
        if "test_host_rans" in item.nodeid:
            item.add_marker(pytest.mark.filterwarnings("ignore::DeprecationWarning"))

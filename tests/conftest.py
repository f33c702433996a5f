import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    """The product package (directory name is not a Python identifier -> loaded by path)."""
    import __graft_entry__
    return __graft_entry__.load_package()


@pytest.fixture(scope="session")
def hip(pkg):
    """The C-ABI HIP library binding; fails loudly when the .so is missing."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return pkg.hip.lib()

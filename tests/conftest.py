import importlib
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# as the package does on import — but some GPU tests touch torch.cuda before they import it
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pop():
    """The product package (directory name has a '-', hence importlib)."""
    return importlib.import_module("pred-occ-planner_amd")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle binding (test infrastructure)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    mod = importlib.import_module("oracle.binding")
    mod.lib()
    return mod

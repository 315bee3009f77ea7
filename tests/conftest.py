import os
import sys

import pytest

# every tensor handed to a kernel is checked for a dense layout in the test suite (ops._ptr); set
# before the package is imported, which reads the switch once
os.environ.setdefault("ADELL_CHECK_DENSE", "1")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cuda():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    return torch.device("cuda:0")

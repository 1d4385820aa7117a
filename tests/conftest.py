import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def pytest_collection_modifyitems(config, items):
    # A `-m gpu` run on a box without a GPU must fail loudly, not skip; but a plain `pytest tests/`
    # on a CPU-only container should not error on GPU tests.
    markexpr = config.getoption("-m") or ""
    if "gpu" in markexpr and "not gpu" not in markexpr:
        return
    if not _has_gpu():
        skip = pytest.mark.skip(reason="no GPU in this container (run with -m gpu on the GPU box)")
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")

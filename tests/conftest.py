import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A plain `pytest tests` on a machine without a HIP device skips the gpu-marked tests; an
    explicit `-m gpu` keeps them and they fail loudly (a GPU run that silently skipped everything
    would read as green)."""
    markexpr = (config.getoption("-m") or "").strip()
    asked_for_gpu = "gpu" in markexpr and "not gpu" not in markexpr
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    if asked_for_gpu:
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(pytest.mark.usefixtures("_no_gpu_fail"))
        return
    skip = pytest.mark.skip(reason="needs a HIP device (select with -m gpu on the GPU box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture
def _no_gpu_fail():
    pytest.fail("-m gpu was selected but no HIP device is visible (there is no CPU fallback)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")

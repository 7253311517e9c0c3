import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


PRECISIONS = ("fp32", "split_bf16")


def pytest_generate_tests(metafunc):
    """Every test that uses the `precision` fixture (all `-m gpu` parity modules do, through `pytestmark`) runs once per
    contraction precision of the library: exact fp32 MFMA and split-bf16 (the mode bench.py reports)."""
    if "precision" in metafunc.fixturenames:
        metafunc.parametrize("precision", PRECISIONS, indirect=True)


@pytest.fixture
def precision(request):
    from incremental_multimodal_medical_learning_ii_amd import _lib
    old = _lib.get_precision()
    _lib.set_precision(request.param)
    try:
        yield request.param
    finally:
        _lib.set_precision(old)


@pytest.fixture(params=["policy", "wide"])
def wide(request):
    """Planes GEMM / convolution tests run twice: with the library's own tile policy, and with the 256x256 LDS-DMA kernel forced
    on every planes launch (so small and ragged shapes exercise it too)."""
    from incremental_multimodal_medical_learning_ii_amd import _lib
    lib = _lib.load()
    old = lib.cxrk_set_wide_mode(2 if request.param == "wide" else 1)
    try:
        yield request.param
    finally:
        lib.cxrk_set_wide_mode(old)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")

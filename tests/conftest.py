import os
import sys

import pytest

# torch ships its own HIP runtime: when a test wants torch device buffers next to libmi355_lzss.so (tests/test_zz_no_alloc.py),
# torch must have loaded first or it finds "no HIP GPUs" in a process whose runtime the library already started - the
# same order as bench.py
try:
    import torch  # noqa: F401
except ImportError:  # pragma: no cover
    torch = None

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

FIX = os.path.join(HERE, "golden", "fixtures")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def fixture_bytes(name):
    with open(os.path.join(FIX, name), "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def dict_naive():
    return fixture_bytes("dict_naive")


@pytest.fixture(scope="session")
def average_block():
    return fixture_bytes("average_block.bin")

import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_sessionstart(session):
    # a GPU box shows 256 logical CPUs and grants 16: keep ATen's host thread team sane
    from diart_amd.hostinfo import limit_host_threads
    limit_host_threads()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The HIP library is built in-tree once (hipcc cross-compiles without a GPU)."""
    from diart_amd import _lib, build
    if not _lib.lib_path().exists():
        build.build()
    yield


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from diart_amd import _lib
    _lib.load()  # fail loudly if the extension is missing on a GPU box
    return torch.device("cuda", 0)

import os
import sys

import pytest

# torch first: it brings its own copy of RCCL, which libdaisyriot_hip.so then shares (dlopen RTLD_NOLOAD).  A test that made the
# library load /opt/rocm's copy before a later test imported torch would leave two RCCL runtimes in one process (they abort at exit).
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def uv50():
    from daisyriot_amd import scenes
    return scenes.visibility_samples(50)

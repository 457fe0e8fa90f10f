import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """A fresh checkout has no built libraries (they are git-ignored): build them once per session, as
    __graft_entry__.build() does.  hipcc cross-compiles gfx950 without a GPU."""
    import subprocess
    lib = os.path.join(ROOT, "volxel_amd", "libvolxel_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "volxel_amd", "csrc"), "-s"])
    yield


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def native_lib():
    """libvolxel_hip.so (host parts are usable without a GPU)."""
    from volxel_amd import _abi
    if not os.path.exists(_abi.LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "volxel_amd", "csrc"), "-s"])
    return _abi.load_library()

import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as g
    so = ROOT / "dril.jl_amd" / "csrc" / "libdril_hip.so"
    if not so.exists():            # fresh checkout (the .so is git-ignored): compile it — hipcc cross-compiles gfx950 without a GPU
        g.build()
    return g.load_package()


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle_lib
    oracle_lib.ensure_built()
    return oracle_lib


@pytest.fixture(scope="session")
def hip(pkg):
    """the product library; GPU tests fail loudly if it is missing (no fallback)."""
    return pkg._capi.load_library()

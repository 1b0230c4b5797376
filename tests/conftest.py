import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A checkout without the built library (the .so is git-ignored) builds it once before the tests: hipcc is part of the image on the
    build container and on the GPU box.  This is not a fallback -- the product still refuses to run without its HIP library."""
    lib = ROOT / "phoskintime_amd" / "libphoskin_hip.so"
    if not lib.exists():
        import shutil
        if shutil.which("hipcc") or Path("/opt/rocm/bin/hipcc").exists():
            import __graft_entry__ as g
            g.build()


def _is_wide(f) -> bool:
    """Systems beyond one wavefront's lane groups (csrc/pk_wide.hpp): distmod / succmod with more than 64 states, randmod n >= 7."""
    _, model, n, _ = f.stem.split("_", 3)
    n = int(n[1:])
    return n >= 7 if model == "randmod" else n + 2 > 64


@pytest.fixture(scope="session")
def golden_files():
    files = [f for f in sorted((ROOT / "tests" / "golden").glob("protein_*.npz")) if not _is_wide(f)]
    assert files, "tests/golden is empty: run tools/make_golden.py in the build container"
    return files


@pytest.fixture(scope="session")
def golden_wide_files():
    files = [f for f in sorted((ROOT / "tests" / "golden").glob("protein_*.npz")) if _is_wide(f)]
    assert files, "no wide-system fixtures: run tools/make_golden.py randmod 7 / distmod 100 / ... in the build container"
    return files


@pytest.fixture(scope="session")
def built_lib():
    """libphoskin_hip.so, built on demand (hipcc cross-compiles for gfx950 without a GPU)."""
    import __graft_entry__ as g
    g.build()
    from phoskintime_amd import _capi
    return _capi.load()

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_sessionstart(session):
    """Both native libraries are built HERE, before any test (and so before anything initialises the GPU): tests/helpers.oracle_lib()
    only loads the stamped library afterwards."""
    import __graft_entry__ as g

    for build, lib in ((g.build_hip, g.HIP_LIB), (g.build_oracle, g.ORACLE_LIB)):
        try:
            build()
        except Exception:       # no compiler here: an existing library is still what gets tested
            if not os.path.isfile(lib):
                raise


@pytest.fixture(scope="session")
def built():
    """Make sure both native libraries exist (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g

    for build, lib in ((g.build_hip, g.HIP_LIB), (g.build_oracle, g.ORACLE_LIB)):
        try:
            build()
        except Exception:       # no compiler here: an existing library is still what gets tested
            if not os.path.isfile(lib):
                raise
    return g


@pytest.fixture(autouse=True)
def _sgw_options_back_to_defaults():
    """Dispatcher options a test sets (``N.set_option`` -- sgw_set_option with a NULL engine) are process-wide: every test
    starts and ends on the defaults."""
    yield
    from sorrel_amd import _native as N

    if N._lib is not None:
        N.reset_options()          # (the defaults + what SGW_OPTIONS asked for at load)

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)



def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def built():
    """Native artefacts must exist; build them once if a fresh checkout."""
    import __graft_entry__ as g
    lib = os.path.join(ROOT, "hwang_amd", "libhwang_amd.so")
    ora = os.path.join(ROOT, "oracle", "_build", "liboracle_h264.so")
    import glob
    pymod = glob.glob(os.path.join(ROOT, "hwang", "_python*.so"))
    if not (os.path.exists(lib) and os.path.exists(ora) and pymod):
        g.build()
    return True


@pytest.fixture(scope="session")
def hw(built):
    import hwang_amd
    return hwang_amd


@pytest.fixture(scope="session")
def oracle(built):
    from oracle import oracle_py
    return oracle_py

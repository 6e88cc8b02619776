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


STAMP = 100  # decoder type the protocol fixture is registered as (>= HWANG_DECODER_FIRST_REGISTERED)


def load_stamp_backend():
    """Builds tests/fixtures/stamp_backend.c (a backend in plain C behind the library's plug-in boundary)
    and registers it with libhwang_amd.so as decoder type STAMP.  Returns the fixture library (its call
    counters: hwang_stamp_trace / hwang_stamp_reset)."""
    import ctypes
    import subprocess
    from hwang_amd._capi import lib
    out = os.path.join(ROOT, "tests", "_build", "libstamp_backend.so")
    src = os.path.join(ROOT, "tests", "fixtures", "stamp_backend.c")
    if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(out), exist_ok=True)
        tmp = "%s.%d.tmp" % (out, os.getpid())   # (pytest-xdist: several workers may get here at once)
        subprocess.check_call(["gcc", "-O1", "-fPIC", "-shared", "-pthread", src, "-o", tmp])
        os.replace(tmp, out)
    fixture = ctypes.CDLL(out)
    vtable = ctypes.c_void_p.in_dll(fixture, "hwang_stamp_vtable")
    assert lib.hwang_register_decoder_backend(STAMP, ctypes.addressof(vtable)) == 0
    fixture.hwang_stamp_trace.argtypes = [ctypes.POINTER(ctypes.c_int32)]
    return fixture


@pytest.fixture(scope="session")
def hw(built):
    import hwang_amd
    hwang_amd.stamp_fixture = load_stamp_backend()
    hwang_amd.STAMP = STAMP
    return hwang_amd


@pytest.fixture(scope="session")
def oracle(built):
    from oracle import oracle_py
    return oracle_py

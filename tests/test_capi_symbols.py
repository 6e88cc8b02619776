"""The C-ABI library loads on a CPU-only box and exports every symbol
include/hwang_amd.h declares; the ctypes table mirrors the header 1:1."""
import ctypes
import os
import re

from tests.util import ROOT


def header_functions():
    text = open(os.path.join(ROOT, "include", "hwang_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hwang_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported(hw):
    from hwang_amd import _capi
    names = header_functions()
    assert len(names) > 50
    lib = ctypes.CDLL(_capi.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_capi.PROTOTYPES) == names


def test_no_torch_types_in_header():
    text = open(os.path.join(ROOT, "include", "hwang_amd.h")).read()
    assert "torch" not in text and "at::" not in text and "std::" not in text


def test_product_does_not_reference_oracle():
    """The product path must never import, link or call the oracle."""
    bad = []
    for base in ("hwang_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".cpp", ".hip", ".c")):
                    t = open(os.path.join(dirpath, f), errors="ignore").read()
                    if f.endswith(".py"):
                        hit = re.search(r'^\s*(from|import)\s+oracle\b|liboracle|oracle_py', t, flags=re.M)
                    else:
                        hit = re.search(r'#include\s+"[^"]*oracle|liboracle|h264o_|h264i_|hevco_', t)
                    if hit:
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
    mk = open(os.path.join(ROOT, "Makefile")).read()
    lib_rule = mk[mk.index("$(LIB): $(OBJS)"):].split("\n\n")[0]
    assert "oracle" not in lib_rule

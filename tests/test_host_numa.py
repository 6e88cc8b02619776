"""util/numa.h: the sysfs cpulist parser is called by every lane thread of a multi-GPU retriever at the same moment
(it was written with strtok, whose cursor is process-wide)."""
import os
import subprocess

from tests.util import ROOT


def test_cpulist_parser_is_reentrant(tmp_path):
    exe = str(tmp_path / "numa_parse")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-Wall",
                           os.path.join(ROOT, "tests", "fixtures", "numa_parse.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr


def test_no_strtok_in_the_product():
    for base, _dirs, files in os.walk(os.path.join(ROOT, "hwang_amd", "csrc")):
        for f in files:
            text = open(os.path.join(base, f), errors="replace").read()
            assert "strtok(" not in text, f

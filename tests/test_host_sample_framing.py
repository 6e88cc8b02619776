"""Which sync samples start a closed GOP (hwang_amd/csrc/util/avc_sample.h): the retriever makes a work item of every
closed GOP that holds a wanted row, the automata tells the backend where it may stop decoding.  Until round 6's last
pass an HEVC clip's hvcC box was read as an avcC box there ("1-byte NAL lengths", then "not an IDR picture" for every
sample): correct frames, but byte-adjacent GOPs stayed one work item and were decoded to their end."""
import os
import subprocess

from tests.util import ROOT


def test_closed_gop_detection_for_both_codecs(tmp_path):
    exe = str(tmp_path / "sample_framing_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "hwang_amd", "csrc"),
                           os.path.join(ROOT, "tests", "fixtures", "sample_framing_test.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr

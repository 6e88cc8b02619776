"""k_inter's window path works on packed data (hwang_amd/csrc/amd/inter_filters.h): horizontal six-tap sums by the 8-bit
dot product, vertical ones on 16-bit pairs, averages on four bytes at once.

CPU: the header bound to plain C restatements of the four gfx950 instructions it uses, against a scalar restatement of
8.4.2.2.1 / 8.4.2.2.2 (tests/fixtures/inter_filters_host.cpp) - every fractional position, random / saturated /
flat windows.  And a toolchain guard: ROCm 7.2 selects v_ashr_pk_u8_i32 for `clip8(x >> n) | clip8(y >> n) << 8`
and assumes the upper half of its result is zero, which the hardware does not honour for negative inputs
(found by the GPU test below) - no kernel of this repo may contain it.

GPU: the same header compiled for the device and for the host (restatements) agrees bit for bit, which pins the
restatements to the hardware."""
import glob
import os
import subprocess

import pytest

from tests.util import ROOT

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
HIPCC = "/opt/rocm/bin/hipcc"
BUILD = os.path.join(ROOT, "tests", "_build")


def test_packed_interpolation_matches_the_standard(tmp_path):
    if not os.path.exists(CLANG):
        pytest.skip("no clang++ (the header uses clang vector extensions)")
    exe = str(tmp_path / "inter_filters_host")
    subprocess.check_call([CLANG, "-std=c++17", "-O1", "-Wall", "-Wno-unused-function",
                           os.path.join(ROOT, "tests", "fixtures", "inter_filters_host.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr
    assert int(out.stdout.split()[1]) > 2_000_000


def test_no_kernel_contains_the_miscompiled_pack(tmp_path):
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    for src in sorted(glob.glob(os.path.join(ROOT, "hwang_amd", "csrc", "amd", "*.hip"))):
        asm = str(tmp_path / (os.path.basename(src) + ".s"))
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "--offload-device-only", "-S", src,
                               "-o", asm], stderr=subprocess.DEVNULL)
        text = open(asm).read()
        for bad in ("v_ashr_pk_u8_i32", "v_ashr_pk_i8_i32"):
            assert bad not in text, "%s in %s: see inter_filters.h clip_pack4" % (bad, os.path.basename(src))


def gpu_binary():
    exe = os.path.join(BUILD, "inter_filters_gpu")
    src = os.path.join(ROOT, "tests", "fixtures", "inter_filters_gpu.hip")
    hdr = os.path.join(ROOT, "hwang_amd", "csrc", "amd", "inter_filters.h")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        os.makedirs(BUILD, exist_ok=True)
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-std=c++17", "-O2", "-Wno-unused-value", src, "-o", exe])
    return exe


def test_gpu_harness_builds():
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc")
    assert os.path.exists(gpu_binary())


@pytest.mark.gpu
def test_device_and_host_bindings_agree():
    out = subprocess.run([gpu_binary()], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr

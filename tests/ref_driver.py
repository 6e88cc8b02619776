"""Builds and runs tests/fixtures/ref_automata_driver.cpp: the REFERENCE's own DecoderAutomata
(/root/reference/hwang/decoder_automata.cpp, compiled where it lies) over INTEGRATION.md's adapter (extracted verbatim
from that file) and libhwang_amd.so's C ABI.  The binary lands in tests/_build/ (git-ignored, travels to the GPU box
with the snapshot); it can only be BUILT where /root/reference exists."""
import os
import re
import subprocess

from tests.util import ROOT

REF = "/root/reference"
DRIVER = os.path.join(ROOT, "tests", "_build", "ref_automata_driver")
STAMP_LIB = os.path.join(ROOT, "tests", "_build", "libstamp_backend.so")


def shim_source():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = [b for b in re.findall(r"```cpp\n(.*?)```", doc, flags=re.S) if "class AMDVideoDecoderShim" in b]
    assert len(blocks) == 1
    return blocks[0]


def build_driver(force=False):
    """Returns the driver's path, building it when the reference tree is here; None when it is neither built nor buildable."""
    src = os.path.join(ROOT, "tests", "fixtures", "ref_automata_driver.cpp")
    have_ref = os.path.exists(os.path.join(REF, "hwang", "decoder_automata.cpp"))
    deps = [src, os.path.join(ROOT, "INTEGRATION.md"), os.path.join(ROOT, "include", "hwang_amd.h"),
            os.path.join(ROOT, "hwang_amd", "libhwang_amd.so")]
    fresh = os.path.exists(DRIVER) and all(os.path.getmtime(DRIVER) >= os.path.getmtime(d) for d in deps if os.path.exists(d))
    if fresh and not force:
        return DRIVER
    if not have_ref:
        return DRIVER if os.path.exists(DRIVER) else None
    inc = os.path.join(ROOT, "tests", "_build", "shim_inc")
    os.makedirs(os.path.join(inc, "hwang", "impls", "amd"), exist_ok=True)
    with open(os.path.join(inc, "hwang", "impls", "amd", "amd_video_decoder_shim.h"), "w") as f:
        f.write(shim_source())
    cmd = ["g++", "-std=c++14", "-O1", "-pthread", "-Wall", "-Wno-sign-compare", "-Wno-unused-variable",
           "-include", "cstdint", "-include", "string", "-include", "functional",
           "-I", os.path.join(ROOT, "tests", "stubs"), "-I", REF, "-I", os.path.join(ROOT, "include"), "-I", inc,
           src, os.path.join(REF, "hwang", "decoder_automata.cpp"), "-o", DRIVER,
           "-L", os.path.join(ROOT, "hwang_amd"), "-lhwang_amd", "-ldl", "-Wl,-rpath,$ORIGIN/../../hwang_amd"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        raise RuntimeError("ref_automata_driver does not build:\n" + r.stderr[-3000:])
    return DRIVER


def write_plan(path, idx, plan, chunks, extradata=b""):
    """EncodedData per interval as python/hwang/decoder.py:36-63 fills it (bytes of the interval, offsets rebased)."""
    offs, sizes = list(idx.sample_offsets()), list(idx.sample_sizes())
    offs.append(offs[-1] + sizes[-1])
    sizes.append(0)
    kfs = list(idx.keyframe_indices())
    lines = ["%d %d %s" % (idx.frame_width(), idx.frame_height(), idx.format()), extradata.hex() or "-", str(len(plan))]
    for (s, e), valid in plan:
        start = offs[s]
        lines.append("%d %d %d %d" % (s, e, start, offs[e] + sizes[e] - start))
        lines.append(" ".join(str(o - start) for o in offs[s:e]))
        lines.append(" ".join(str(x) for x in sizes[s:e]))
        lines.append(" ".join(str(k) for k in kfs if s <= k <= e))
        lines.append(" ".join(str(v) for v in valid))
    lines.append(" ".join(str(c) for c in chunks))
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")


def run_driver(plan_path, blob_path, out_path, decoder_type, device_type, stamp=False, timeout=300):
    env = dict(os.environ)
    if stamp:
        env["HWANG_TEST_STAMP_LIB"] = STAMP_LIB
    r = subprocess.run([DRIVER, plan_path, blob_path, out_path, str(decoder_type), str(device_type)],
                       capture_output=True, text=True, env=env, timeout=timeout)
    return r

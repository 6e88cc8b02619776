"""Sanitizer runs of the host half (GPU sanitizers are not available on this
pool; the kernels are covered by the parity tests).

* ThreadSanitizer: ParallelRetriever over three STAMP devices and 12 lanes, a
  two-request batch, DecoderAutomata's feeder / caller threads through
  Queue<T> (ADVICE r1 found a real race in this area), and the H.264 parser with
  1, 2 and 5 entropy workers (num_devices > 1) over clips with temporal direct
  and reference B pictures, records equal to the serial decode;
* AddressSanitizer + UBSan: MP4 indexer -> proto codec -> planner -> H.264
  parser (both hand-over forms: full and compact motion) over corpus clips
  and over 24 bit-flipped copies of each."""
import os
import subprocess

import pytest

from tests.corpus_util import make_clip
from tests.util import ROOT

CSRC = os.path.join(ROOT, "hwang_amd", "csrc")
SOURCES = [os.path.join(ROOT, "tests", "sanitize", "san_main.cpp")] + [os.path.join(CSRC, f) for f in (
    "retriever.cpp", "decoder_automata.cpp", "registered_backend.cpp", "video_decoder_factory.cpp", "video_index.cpp",
    "mp4_index_creator.cpp", "h264/h264_parser.cpp", "h264/h264_cabac_init.cpp", "hevc/hvcc.cpp", "capi_probe.cpp")]


def build(tmp_path, name, flags):
    exe = str(tmp_path / name)
    fixture_obj = str(tmp_path / (name + "_stamp.o"))   # the STAMP fixture is C: compile it as such, sanitized too
    r = subprocess.run(["gcc", "-O1", "-g", "-pthread", "-fno-omit-frame-pointer"] + [f for f in flags if f.startswith("-fsan")] +
                       ["-c", os.path.join(ROOT, "tests", "fixtures", "stamp_backend.c"), "-o", fixture_obj],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-march=x86-64-v3", "-pthread", "-fno-omit-frame-pointer", "-I", CSRC,
           "-I", os.path.join(ROOT, "include")] + flags + SOURCES + [fixture_obj, "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_thread_sanitizer(built, tmp_path):
    clips = []
    for i, kw in enumerate([dict(frames=26, gop=13, bframes=2, refs=2, temporal_direct=1, implicit_wp=1, slices=2, width=96, height=64),
                            dict(frames=34, gop=17, bframes=3, refs=4, bpyramid=1, mixed_slices=1, slice_mbs=7, multi_pps=2, cabac=0,
                                 width=80, height=64)]):
        p = str(tmp_path / ("t%d.mp4" % i))
        make_clip(p, yuv=False, **kw)
        clips.append(p)
    exe = build(tmp_path, "san_tsan", ["-fsanitize=thread"])
    r = subprocess.run([exe, "threads"] + clips, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=0 exitcode=66"))
    assert r.returncode == 0 and "threads: ok" in r.stdout and "WARNING: ThreadSanitizer" not in r.stderr, \
        (r.stdout + r.stderr)[-4000:]


def test_address_and_ub_sanitizer(built, tmp_path):
    clips = []
    for i, kw in enumerate([dict(frames=7, gop=7, bframes=2, cabac=1, t8x8=1, refs=2, width=96, height=64),
                            dict(frames=6, gop=3, bframes=1, cabac=0, t8x8=0, refs=2, width=64, height=48, slices=2),
                            dict(frames=8, gop=8, bframes=0, refs=3, mmco=1, rplm=1, explicit_wp=1, scaling=3, width=80,
                                 height=64, fragmented=1),
                            dict(frames=13, gop=13, bframes=3, refs=4, bpyramid=1, mixed_slices=1, slice_mbs=5, multi_pps=2,
                                 temporal_direct=1, implicit_wp=1, fn_bits=4, poc_bits=5, extra_nals=1, nal_length_size=2,
                                 width=96, height=80),
                            dict(frames=12, gop=6, bframes=2, refs=3, poc_type=1, constrained_intra=1, direct8x8=0,
                                 rate_kbit=3, crop_left=3, crop_top=2, width=90, height=70, cabac=0)]):
        p = str(tmp_path / ("c%d.mp4" % i))
        make_clip(p, yuv=False, **kw)
        clips.append(p)
    exe = build(tmp_path, "san_asan", ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"])
    r = subprocess.run([exe, "memory"] + clips, capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "memory: ok" in r.stdout, (r.stdout + r.stderr)[-4000:]

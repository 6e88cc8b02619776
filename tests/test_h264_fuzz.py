"""Damaged H.264 streams through the product's host parser (hwang_amd/csrc/h264/h264_parser.cpp), built with
AddressSanitizer and UBSan on the CPU (tests/fixtures/h264_parser_fuzz.cpp), in both hand-over forms (full motion
records; compact records whose vectors k_mv derives).  Beyond memory safety of the parser itself
(tests/test_sanitizers.py) the fixture's sink checks every record of every picture the kernels of k_recon.hip would be
given: level runs inside the picture's level stream, level groups the mask has, prediction modes the tables have,
reference surfaces the picture holds, list indices inside the weight tables, the compact stream's row table and length.
On the GPU, tests/test_gpu_decode.py::test_corrupt_streams_never_hang_the_gpu runs damaged streams end to end; it cannot
see a read outside a buffer - this can."""
import os
import subprocess

import pytest

from tests.corpus_util import make_clip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "_build", "h264_parser_fuzz")
CLIPS = [
    ("plain_slices", dict(frames=12, gop=6, bframes=2, refs=3, slices=2, seed=5)),
    ("pyramid_mixed_slices_direct_wp", dict(frames=13, gop=13, bframes=3, refs=4, bpyramid=1, mixed_slices=1, slice_mbs=23, multi_pps=2,
                                            temporal_direct=1, implicit_wp=1, seed=6)),
    ("cavlc_wp_mmco_rplm_scaling", dict(frames=14, gop=7, bframes=1, refs=4, cabac=0, explicit_wp=1, wp_denom=5, mmco=1, rplm=1, scaling=3,
                                        extra_nals=1, seed=7)),
    ("t8x8_cover_wild", dict(frames=10, gop=10, bframes=2, refs=2, t8x8=1, cover=1, wild_mv=1, qp_swing=1, seed=8)),
    ("cavlc_nostss_poc2_cip", dict(frames=9, gop=9, bframes=0, refs=3, cabac=0, poc_type=2, constrained_intra=1, pcm_period=7, seed=9)),
]


@pytest.fixture(scope="module")
def fuzzer(built):
    csrc = os.path.join(ROOT, "hwang_amd", "csrc", "h264")
    src = [os.path.join(ROOT, "tests", "fixtures", "h264_parser_fuzz.cpp"), os.path.join(csrc, "h264_parser.cpp"), os.path.join(csrc, "h264_cabac_init.cpp")]
    deps = src + [os.path.join(csrc, "h264_parser.h"), os.path.join(ROOT, "hwang_amd", "csrc", "amd", "frame_job.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(EXE) < os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(EXE), exist_ok=True)
        tmp = "%s.%d" % (EXE, os.getpid())   # (pytest-xdist: each worker builds its own and renames it into place)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-march=x86-64-v3", "-pthread", "-fsanitize=address,undefined",
                               "-fno-sanitize-recover=undefined", "-Wall", "-Wno-sign-compare", "-o", tmp] + src)
        os.replace(tmp, EXE)
    return EXE


@pytest.mark.parametrize("name,kw", CLIPS, ids=[c[0] for c in CLIPS])
def test_damaged_streams_fail_cleanly_and_never_emit_a_record_the_kernels_cannot_follow(fuzzer, hw, tmp_path, name, kw):
    path = str(tmp_path / (name + ".mp4"))
    make_clip(path, width=176, height=144, yuv=False, **kw)
    idx = hw.index_video(path)
    data = open(path, "rb").read()
    (tmp_path / "avcc").write_bytes(bytes(idx.metadata_bytes()))
    (tmp_path / "blob").write_bytes(b"".join(data[o:o + s] for o, s in zip(idx.sample_offsets(), idx.sample_sizes())))
    (tmp_path / "sizes").write_text("\n".join(str(s) for s in idx.sample_sizes()))
    r = subprocess.run([fuzzer, str(tmp_path / "avcc"), str(tmp_path / "blob"), str(tmp_path / "sizes"), str(sum(map(ord, name))), "120"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), (r.stdout[-500:], r.stderr[-3000:])
    assert " 0 pictures checked" not in r.stdout

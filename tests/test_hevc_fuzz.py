"""Damaged HEVC streams through the product's host parser (hwang_amd/csrc/hevc/hevc_parser.cpp), built with
AddressSanitizer and UBSan on the CPU (tests/fixtures/hevc_parser_fuzz.cpp): flipped bits, overwritten bytes, bursts,
truncated / swapped samples, damaged parameter sets.  The parser must stay inside its memory, end every sample with
success or an error TEXT, and - what the GPU depends on - never hand the kernels of k_hevc.hip a record that points
outside the picture, its lists or its surfaces (the fixture's sink checks every record of every picture it is given).
The reference leaves damaged streams to libavcodec (software_video_decoder.cpp:349-457: a failed avcodec_send_packet is
logged and the frame is dropped); this backend fails the interval with a Result instead."""
import os
import subprocess

import pytest

from tests.corpus_util import HEVC_MATRIX, make_hevc_clip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "_build", "hevc_parser_fuzz")
CASES = ["i_ctb32_sao_pcm", "i_pcm2_tskip_bypass", "i_scaling_random_modes", "i_slices_cip_nolf", "i_wpp", "ibp_amp_wp_ctb32",
         "pyramid_sao_sdh_cover", "rps_sps_pocwrap_ctb64", "everything", "lt_sps_b_tmvp", "lt_pyramid_pocwrap", "inter_rps_slices_pyramid", "rplm_lt_inter_rps_tmvp", "keep_unused_b_rplm", "sub_layers_b", "sh_extras_slices_wpp", "multi_pps", "open_gop_bb", "open_gop_tools", "radl_bb_tmvp"]


@pytest.fixture(scope="module")
def fuzzer(built):
    src = [os.path.join(ROOT, "tests", "fixtures", "hevc_parser_fuzz.cpp"), os.path.join(ROOT, "hwang_amd", "csrc", "hevc", "hevc_parser.cpp")]
    deps = src + [os.path.join(ROOT, "hwang_amd", "csrc", "hevc", "hevc_parser.h"), os.path.join(ROOT, "hwang_amd", "csrc", "amd", "hevc_frame_job.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(EXE) < os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(EXE), exist_ok=True)
        tmp = "%s.%d" % (EXE, os.getpid())   # (pytest-xdist: each worker builds its own and renames it into place)
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                               "-Wall", "-Wno-sign-compare", "-o", tmp] + src)
        os.replace(tmp, EXE)
    return EXE


@pytest.mark.parametrize("name", CASES)
def test_damaged_streams_fail_cleanly_and_never_emit_a_record_outside_the_picture(fuzzer, hw, tmp_path, name):
    kw = dict(dict(HEVC_MATRIX)[name])
    w, h = kw.pop("width", 128), kw.pop("height", 96)
    path = str(tmp_path / (name + ".mp4"))
    make_hevc_clip(path, width=w, height=h, yuv=False, **kw)
    idx = hw.index_video(path)
    data = open(path, "rb").read()
    (tmp_path / "hvcc").write_bytes(bytes(idx.metadata_bytes()))
    (tmp_path / "blob").write_bytes(b"".join(data[o:o + s] for o, s in zip(idx.sample_offsets(), idx.sample_sizes())))
    (tmp_path / "sizes").write_text("\n".join(str(s) for s in idx.sample_sizes()))
    r = subprocess.run([fuzzer, str(tmp_path / "hvcc"), str(tmp_path / "blob"), str(tmp_path / "sizes"), str(sum(map(ord, name))), "150"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), (r.stdout[-500:], r.stderr[-3000:])
    # (damage that the syntax absorbs - a flipped level, a different mode - must still decode: not every stream fails)
    assert " 0 pictures checked" not in r.stdout

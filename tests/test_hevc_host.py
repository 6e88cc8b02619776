"""The product's HOST half of the HEVC path (hwang_amd/csrc/hevc/hevc_parser.cpp: CABAC, candidate derivation, DPB,
output order, the records of amd/hevc_frame_job.h) without a GPU: tests/fixtures/hevc_records_cpu.cpp links the parser
with a sink that interprets its records with scalar loops (a test fixture; nothing of it is in libhwang_amd.so) and
the result must be the corpus encoder's reconstruction, frame for frame.  What k_hevc.hip makes of the same records
is tests/test_gpu_hevc.py's business."""
import os
import subprocess

import numpy as np
import pytest

from tests.corpus_util import HEVC_MATRIX, make_hevc_clip
from tests.util import ROOT

EXE = os.path.join(ROOT, "tests", "_build", "hevc_records_cpu")


@pytest.fixture(scope="module")
def records_cpu(built):
    src = [os.path.join(ROOT, "tests", "fixtures", "hevc_records_cpu.cpp"), os.path.join(ROOT, "hwang_amd", "csrc", "hevc", "hevc_parser.cpp")]
    deps = src + [os.path.join(ROOT, "hwang_amd", "csrc", "hevc", "hevc_parser.h"), os.path.join(ROOT, "hwang_amd", "csrc", "amd", "hevc_frame_job.h"),
                  os.path.join(ROOT, "hwang_amd", "csrc", "hevc", "hevc_tables.h")]
    if not os.path.exists(EXE) or any(os.path.getmtime(EXE) < os.path.getmtime(d) for d in deps):
        os.makedirs(os.path.dirname(EXE), exist_ok=True)
        # (pytest-xdist: several workers may find the binary stale at once - each builds its own and renames it into place)
        tmp = "%s.%d" % (EXE, os.getpid())
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wno-sign-compare", "-o", tmp] + src)
        os.replace(tmp, EXE)
    return EXE


def run_host(exe, hw, path, tmp_path, wanted=(), keyframes=(), first=0):
    idx = hw.index_video(path)
    offs, sizes = idx.sample_offsets(), idx.sample_sizes()
    data = open(path, "rb").read()
    (tmp_path / "hvcc").write_bytes(bytes(idx.metadata_bytes()))
    offs, sizes = offs[first:], sizes[first:]
    (tmp_path / "blob").write_bytes(b"".join(data[o:o + s] for o, s in zip(offs, sizes)))
    (tmp_path / "sizes").write_text("\n".join(str(s) for s in sizes))
    r = subprocess.run([exe, str(tmp_path / "hvcc"), str(tmp_path / "blob"), str(tmp_path / "sizes"), str(tmp_path / "out.yuv")] +
                       [str(w) for w in wanted] + (["kf"] + [str(k) for k in keyframes] if keyframes else []), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr[-1000:]
    info = dict(kv.split("=") for kv in r.stdout.split())
    return np.fromfile(str(tmp_path / "out.yuv"), dtype=np.uint8), {k: int(v) for k, v in info.items()}


@pytest.mark.parametrize("name,kw", HEVC_MATRIX, ids=[m[0] for m in HEVC_MATRIX])
def test_host_parser_records_reproduce_the_clip(records_cpu, hw, tmp_path, name, kw):
    kw = dict(kw)
    w, h = kw.pop("width", 128), kw.pop("height", 96)
    path = str(tmp_path / (name + ".mp4"))
    gold = make_hevc_clip(path, width=w, height=h, **kw)
    got, info = run_host(records_cpu, hw, path, tmp_path)
    got = got.reshape(-1, gold.shape[1])
    assert len(got) == len(gold) == info["frames"] and info["skipped_positions"] == 0
    for i in range(len(gold)):
        assert np.array_equal(got[i], gold[i]), "%s: frame %d" % (name, i)


@pytest.mark.parametrize("sub_layers", [1, 2])
def test_sparse_hint_skips_only_pictures_nothing_refers_to(records_cpu, hw, tmp_path, sub_layers):
    """(sub_layers = 2: the pictures nothing refers to are TSA_N pictures of the upper temporal sub-layer - still the
    highest layer's sub-layer non-reference pictures, still the only ones a sparse request may leave out.)"""
    path = str(tmp_path / "c.mp4")
    gold = make_hevc_clip(path, frames=13, gop=13, bframes=3, bpyramid=1, refs=4, sub_layers=sub_layers)
    wanted = [0, 2, 6, 9, 12]      # 2 is a reference B picture (the middle of 1..3), 6 and 9 are not
    got, info = run_host(records_cpu, hw, path, tmp_path, wanted=wanted)
    got = got.reshape(-1, gold.shape[1])
    assert len(got) == 13 and info["skipped"] > 0 and info["skipped"] == info["skipped_positions"]
    assert info["decoded"] + info["skipped"] == 13
    for r in wanted:
        assert np.array_equal(got[r], gold[r]), "wanted row %d" % r
    # whatever was decoded although unwanted is right too; the skipped positions are the zero frames
    for r in range(13):
        assert np.array_equal(got[r], gold[r]) or not got[r].any()


@pytest.mark.parametrize("bframes", [0, 2])
def test_gop_tails_and_the_interval_tail_are_not_decoded(records_cpu, hw, tmp_path, bframes):
    """An interval of four closed GOPs with wanted rows in the first, second and fourth: told where the GOPs begin
    (Parser::set_keyframes, what DecoderAutomata passes through SparseDecodeHints::set_interval_keyframes), the parser
    decodes each GOP up to its last wanted position plus the reorder depth and nothing of the third; every sample is
    still one output position, and the wanted ones hold the right frames."""
    path = str(tmp_path / "k.mp4")
    gold = make_hevc_clip(path, width=64, height=64, frames=48, gop=12, bframes=bframes, refs=2)
    wanted = [1, 13, 17, 38]
    full, info_full = run_host(records_cpu, hw, path, tmp_path, wanted=wanted)
    got, info = run_host(records_cpu, hw, path, tmp_path, wanted=wanted, keyframes=[0, 12, 24, 36])
    got = got.reshape(-1, gold.shape[1])
    assert len(got) == 48 and info["frames"] == 48 and info["decoded"] + info["skipped"] == 48
    for r in wanted:
        assert np.array_equal(got[r], gold[r]), "wanted row %d" % r
    for r in range(48):
        assert np.array_equal(got[r], gold[r]) or not got[r].any()
    assert not got[24:36].any(), "nothing of the GOP without a wanted row is decoded"
    assert info["decoded"] < info_full["decoded"], (info, info_full)
    if bframes == 0:
        assert info["decoded"] == 2 + 6 + 0 + 3 and info_full["decoded"] == 39, (info, info_full)
    # without the GOP boundaries: everything up to the last wanted position (+ reorder depth), nothing behind it
    full = full.reshape(-1, gold.shape[1])
    assert not full[38 + 6:].any() and info_full["decoded"] <= 38 + 6


@pytest.mark.parametrize("name", ["open_gop_bb", "open_gop_tools"])
def test_decoding_from_a_cra_picture(records_cpu, hw, oracle, tmp_path, name):
    """Random access into an open GOP (what a seek into an x265 stream is): decoding starts at the CRA picture, its
    leading pictures refer to pictures that were never decoded - 8.3.3 makes them up - and are output all the same
    (one frame per sample: the automata counts).  Everything behind the CRA picture in output order is the sequential
    decode's; the leading pictures are whatever the made-up references give, the same in the oracle and here."""
    kw = dict(dict(HEVC_MATRIX)[name])
    path = str(tmp_path / "o.mp4")
    gold = make_hevc_clip(path, **kw)
    idx = hw.index_video(path)
    gop, lead = kw["gop"], kw["bframes"]
    assert list(idx.keyframe_indices()) == list(range(0, kw["frames"], gop))     # the CRA pictures are sync samples
    for first in range(gop, kw["frames"], gop):
        got, info = run_host(records_cpu, hw, path, tmp_path, first=first)
        got = got.reshape(-1, gold.shape[1])
        want, ctr = oracle.decode_file_hevc(path, idx, first=first, count=idx.frames() - first)
        assert len(got) == len(want) == idx.frames() - first
        assert ctr["md5_mismatch"] == lead                      # the oracle's own check: only the leading pictures are off
        for i in range(len(got)):
            assert np.array_equal(got[i], want[i][0]), "from sample %d: output %d" % (first, i)
            if i >= lead:
                assert np.array_equal(got[i], gold[first + i]), "from sample %d: output %d against the sequential decode" % (first, i)
        assert any(not np.array_equal(got[i], gold[first + i]) for i in range(lead))

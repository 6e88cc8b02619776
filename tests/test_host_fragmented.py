"""Container variants of SURVEY 8(f) rank 2: fragmented MP4 (mvex / trex, one
moof + mdat per GOP, three tfhd / trun flag combinations), co64 and stz2 -
written by the corpus muxer, indexed by the product and by the REFERENCE's
indexer compiled in place (oracle/_ref/ref_mp4_index), including the pull
protocol's (offset, size) trace.

Reference quirk pinned here rather than copied: for a fragment whose base is
the moof itself (tfhd default-base-is-moof, or no base at all) the reference
counts the moof's position INSIDE the fed chunk twice
(mp4_index_creator.cpp:598-606, `offset_ + before_moof_offset`), so a moof that
does not start the chunk it is parsed from - the first one, which shares the
first KiB with ftyp and moov - gets sample offsets that point past its data.
The product follows ISO/IEC 14496-12 8.8.7; the test asserts that the
difference is exactly that in-chunk position and that frames decode from the
product's index."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

from tests.corpus_util import make_clip
from tests.util import ROOT

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "ref_mp4_index")


def top_level_boxes(path):
    data = open(path, "rb").read()
    out, off = [], 0
    while off + 8 <= len(data):
        size, kind = struct.unpack(">I4s", data[off:off + 8])
        if size == 1:
            size = struct.unpack(">Q", data[off + 8:off + 16])[0]
        out.append((kind.decode(), off, size))
        off += size
    return out


def pull_trace(hw, path):
    size = os.path.getsize(path)
    ind = hw.MP4IndexCreator(size)
    trace, off, n = [], 0, min(1024, size)
    with open(path, "rb") as f:
        while not ind.is_done():
            trace.append([off, n])
            f.seek(off)
            _, off, n = ind.feed(f.read(n), n)
    assert not ind.is_error(), ind.error_message()
    return trace


@pytest.mark.parametrize("variant", [1, 2, 3])
def test_fragmented_index(hw, oracle, tmp_path, variant):
    path = str(tmp_path / ("frag%d.mp4" % variant))
    gold = make_clip(path, frames=30, gop=8, bframes=2, fragmented=variant)
    idx = hw.index_video(path)
    boxes = top_level_boxes(path)
    moofs = [(off, size) for kind, off, size in boxes if kind == "moof"]
    mdats = [(off, size) for kind, off, size in boxes if kind == "mdat"]
    assert len(moofs) == len(mdats) == 4 and boxes[0][0] == "ftyp" and boxes[1][0] == "moov"
    assert idx.frames() == 30 and idx.keyframe_indices() == [0, 8, 16, 24]
    assert (idx.frame_width(), idx.frame_height(), idx.format()) == (320, 240, "avc1")
    # every sample lies inside its fragment's mdat, back to back
    offs, sizes = idx.sample_offsets(), idx.sample_sizes()
    for g, (moff, msize) in enumerate(mdats):
        first, last = 8 * g, min(30, 8 * g + 8)
        assert offs[first] == moff + 8
        for s in range(first + 1, last):
            assert offs[s] == offs[s - 1] + sizes[s - 1]
        assert offs[last - 1] + sizes[last - 1] == moff + msize
    # the planner never merges across fragments (a moof sits between the GOPs)
    plan = hw.slice_into_video_intervals(idx, [0, 9, 29])
    assert [iv for iv, _ in plan] == [(0, 8), (8, 16), (24, 30)]
    # frames decode from this index
    frames, c = oracle.decode_file(path, idx)
    assert c["slices_clean_end"] == c["slices"] and len(frames) == 30
    for i in (0, 7, 8, 29):
        assert np.array_equal(frames[i][0], gold[i])
    if not os.path.exists(REF_BIN):
        pytest.skip("reference indexer not built here")
    ref = json.loads(subprocess.check_output([REF_BIN, path]))
    assert not ref["error"]
    assert ref["sample_sizes"] == sizes and ref["keyframes"] == idx.keyframe_indices()
    assert ref["metadata"] == idx.metadata_bytes().hex()
    assert (ref["timescale"], ref["duration"], ref["width"], ref["height"], ref["format"]) == (
        idx.timescale(), idx.duration(), idx.frame_width(), idx.frame_height(), idx.format())
    assert ref["trace"] == pull_trace(hw, path)
    if variant == 2:          # explicit base_data_offset: identical
        assert ref["sample_offsets"] == offs
    else:                     # base = the moof: the reference adds the moof's in-chunk position once more
        seen_quirk = False
        for g, (moff, _) in enumerate(moofs):
            chunk_start = [o for o, n in ref["trace"] if o <= moff < o + n][-1]
            seen_quirk |= moff != chunk_start
            for s in range(8 * g, min(30, 8 * g + 8)):
                assert ref["sample_offsets"][s] - offs[s] == moff - chunk_start, (g, s)
        assert seen_quirk


@pytest.mark.parametrize("kw", [dict(co64=1), dict(stz2=1), dict(co64=1, stz2=1)], ids=["co64", "stz2", "co64_stz2"])
def test_co64_and_stz2_match_the_reference(hw, oracle, tmp_path, kw):
    path = str(tmp_path / "v.mp4")
    gold = make_clip(path, frames=12, gop=6, bframes=1, **kw)
    raw = open(path, "rb").read()
    assert (b"co64" in raw) == bool(kw.get("co64")) and (b"stz2" in raw) == bool(kw.get("stz2"))
    idx = hw.index_video(path)
    frames, _ = oracle.decode_file(path, idx)
    assert len(frames) == 12 and np.array_equal(frames[11][0], gold[11])
    if not os.path.exists(REF_BIN):
        pytest.skip("reference indexer not built here")
    ref = json.loads(subprocess.check_output([REF_BIN, path]))
    assert not ref["error"]
    assert ref["sample_offsets"] == idx.sample_offsets() and ref["sample_sizes"] == idx.sample_sizes()
    assert ref["keyframes"] == idx.keyframe_indices() and ref["trace"] == pull_trace(hw, path)


LAYOUTS = [
    ("moov_last", dict(moov_last=1)),
    ("free_and_uuid_boxes", dict(free_boxes=1)),
    ("mdat_64bit_size", dict(large_mdat=1)),
    ("audio_track_first", dict(audio_track=1)),
    ("all_of_them_co64", dict(moov_last=1, free_boxes=1, large_mdat=1, audio_track=1, co64=1)),
    ("audio_track_stz2", dict(audio_track=1, stz2=1, free_boxes=1)),
    # chunks of 3-4 / 5-6 samples (several stsc runs), the sound track's chunks interleaved: one offset per chunk, with gaps
    ("chunks4", dict(chunk_samples=4)),
    ("chunks3_interleaved_audio", dict(chunk_samples=3, audio_track=1)),
    ("chunks5_interleaved_moov_last_co64_stz2", dict(chunk_samples=5, audio_track=1, moov_last=1, co64=1, stz2=1)),
]


@pytest.mark.parametrize("name,kw", LAYOUTS, ids=[l[0] for l in LAYOUTS])
def test_file_layouts_match_the_reference(hw, oracle, tmp_path, name, kw):
    """Files as muxers leave them when nobody "fast-starts" them (moov behind mdat), with padding and vendor boxes,
    a 64-bit mdat size, a sound track in front of the video track: index, extradata and the pull protocol's
    (offset, size) trace equal the reference indexer's, and frames decode from the index."""
    path = str(tmp_path / "v.mp4")
    gold = make_clip(path, frames=12, gop=6, bframes=1, **kw)
    idx = hw.index_video(path)
    assert idx.frames() == 12 and idx.keyframe_indices() == [0, 6]
    offs, sizes = idx.sample_offsets(), idx.sample_sizes()
    if kw.get("audio_track") and kw.get("chunk_samples"):  # the sound chunks leave gaps between the video chunks
        assert any(offs[i + 1] != offs[i] + sizes[i] for i in range(11))
    assert (idx.frame_width(), idx.frame_height(), idx.format()) == (320, 240, "avc1")
    frames, c = oracle.decode_file(path, idx)
    assert c["slices_clean_end"] == c["slices"] and len(frames) == 12
    assert np.array_equal(frames[11][0], gold[11]) and np.array_equal(frames[0][0], gold[0])
    if not os.path.exists(REF_BIN):
        pytest.skip("reference indexer not built here")
    ref = json.loads(subprocess.check_output([REF_BIN, path]))
    assert not ref["error"]
    assert ref["sample_offsets"] == idx.sample_offsets() and ref["sample_sizes"] == idx.sample_sizes()
    assert ref["keyframes"] == idx.keyframe_indices() and ref["metadata"] == idx.metadata_bytes().hex()
    assert (ref["timescale"], ref["duration"], ref["width"], ref["height"], ref["format"]) == (
        idx.timescale(), idx.duration(), idx.frame_width(), idx.frame_height(), idx.format())
    assert ref["trace"] == pull_trace(hw, path)


def test_all_intra_track_without_stss(hw, oracle, tmp_path):
    """No sync sample box: every sample is a keyframe (mp4_index_creator.cpp:413-417) - and every frame its own interval."""
    path = str(tmp_path / "intra.mp4")
    gold = make_clip(path, frames=7, gop=1, bframes=0, no_stss=1, chunk_samples=3)
    assert b"stss" not in open(path, "rb").read()
    idx = hw.index_video(path)
    assert idx.keyframe_indices() == list(range(7))
    plan = hw.slice_into_video_intervals(idx, [2, 5])
    assert [iv for iv, _ in plan] == [(2, 3), (5, 6)]
    frames, _ = oracle.decode_file(path, idx)
    assert np.array_equal(frames[6][0], gold[6])
    if not os.path.exists(REF_BIN):
        pytest.skip("reference indexer not built here")
    ref = json.loads(subprocess.check_output([REF_BIN, path]))
    assert not ref["error"] and ref["keyframes"] == idx.keyframe_indices()
    assert ref["sample_offsets"] == idx.sample_offsets() and ref["trace"] == pull_trace(hw, path)

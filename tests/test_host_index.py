"""MP4 indexer, VideoIndex wire format and seek planner vs golden vectors.

Golden values were captured from the REFERENCE binaries in the build
container (mp4_index_creator.cpp and
video_index.cpp compiled from /root/reference, Python protobuf for the wire
bytes).  Where oracle/_ref/ref_mp4_index exists (reference indexer compiled in
place by `make -C oracle ref`) the product indexer is also diffed against it
live on synthetic files."""
import json
import os
import subprocess

import pytest

from tests.util import COCKATOO, REALSHORT, ROOT, have_file

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "host_golden.json")))


@pytest.mark.parametrize("name,path", [("realshort", REALSHORT), ("cockatoo", COCKATOO)])
def test_index_real_files(hw, name, path):
    if not have_file(path):
        pytest.skip("sample file not in this image")
    g = GOLD["index"][name]
    idx = hw.index_video(path)
    assert idx.timescale() == g["timescale"]
    assert idx.duration() == g["duration"]
    assert (idx.frame_width(), idx.frame_height()) == tuple(g["size"])
    assert idx.format() == g["format"]
    assert idx.frames() == g["frames"]
    assert idx.keyframe_indices() == g["keyframes"]
    first = list(zip(idx.sample_offsets()[:5], idx.sample_sizes()[:5]))
    assert [list(t) for t in first] == g["first_samples"]
    assert idx.metadata_bytes().hex() == g["metadata"]
    assert abs(idx.fps() - g["fps"]) < 1e-6


def test_index_pull_protocol_call_count(hw):
    if not have_file(REALSHORT):
        pytest.skip("sample file not in this image")
    size = os.path.getsize(REALSHORT)
    ind = hw.MP4IndexCreator(size)
    calls, off, n = 0, 0, 1024
    with open(REALSHORT, "rb") as f:
        while not ind.is_done():
            f.seek(off)
            ret, off, n2 = ind.feed(f.read(n), n)
            n = n2
            calls += 1
    assert calls == GOLD["index"]["realshort"]["feed_calls"]
    assert not ind.is_error()


def test_proto_wire_golden(hw):
    g = GOLD["proto"]
    idx = hw.VideoIndex(timescale=g["timescale"], duration=g["duration"], width=g["width"],
                        height=g["height"], format=g["format"], sample_offsets=g["offsets"],
                        sample_sizes=g["sizes"], keyframe_indices=g["keyframes"],
                        metadata=bytes.fromhex(g["metadata"]))
    assert idx.serialize().hex() == g["wire"]
    back = hw.VideoIndex.deserialize(bytes.fromhex(g["wire"]))
    assert back.sample_offsets() == g["offsets"]
    assert back.sample_sizes() == g["sizes"]
    assert back.keyframe_indices() == g["keyframes"]
    assert back.format() == g["format"] and back.timescale() == g["timescale"]
    assert back.duration() == g["duration"] and back.metadata_bytes().hex() == g["metadata"]
    assert back.serialize().hex() == g["wire"]


def test_proto_rejects_garbage(hw):
    with pytest.raises(RuntimeError):
        hw.VideoIndex.deserialize(b"\x1a\xff\xff\xff\xff\x0f")  # length past the end


def _contiguous_index(hw, n=50, gop=10, gap=0):
    offs, sizes, o = [], [], 100
    for i in range(n):
        if gap and i % gop == 0:
            o += gap
        offs.append(o)
        sizes.append(10 + i)
        o += 10 + i
    return hw.VideoIndex(timescale=1000, duration=n * 1000, width=16, height=16, format="avc1",
                         sample_offsets=offs, sample_sizes=sizes,
                         keyframe_indices=list(range(0, n, gop)), metadata=b"")


def test_planner_golden_contiguous(hw):
    idx = _contiguous_index(hw)
    for case in GOLD["planner_contiguous"]:
        got = hw.slice_into_video_intervals(idx, case["rows"])
        assert [[list(iv), v] for iv, v in got] == case["plan"], case["rows"]


def test_planner_golden_gapped(hw):
    idx = _contiguous_index(hw, n=40, gop=10, gap=7)
    got = hw.slice_into_video_intervals(idx, list(range(40)))
    assert [list(iv) for iv, _ in got] == [[0, 10], [10, 20], [20, 30], [30, 40]]
    assert [v for _, v in got] == [list(range(a, a + 10)) for a in (0, 10, 20, 30)]


def test_planner_golden_cockatoo(hw):
    if not have_file(COCKATOO):
        pytest.skip("sample file not in this image")
    idx = hw.index_video(COCKATOO)
    for case in GOLD["planner_cockatoo"]:
        got = hw.slice_into_video_intervals(idx, case["rows"])
        assert [[list(iv), v] for iv, v in got] == case["plan"]


def test_planner_row_out_of_range(hw):
    idx = _contiguous_index(hw)
    with pytest.raises(RuntimeError):
        hw.slice_into_video_intervals(idx, [50])


def test_planner_matches_python_restatement(hw, oracle):
    """Property test against the pure-Python restatement in oracle/."""
    import random
    from oracle import host_oracle
    rnd = random.Random(7)
    for trial in range(200):
        n = rnd.randint(5, 120)
        gop = rnd.randint(1, 15)
        gap = rnd.choice([0, 0, 3])
        idx = _contiguous_index(hw, n=n, gop=gop, gap=gap)
        k = rnd.randint(0, n)
        rows = sorted(rnd.sample(range(n), k))
        want = host_oracle.slice_into_video_intervals(
            idx.sample_offsets(), idx.sample_sizes(), idx.keyframe_indices(), rows)
        got = hw.slice_into_video_intervals(idx, rows)
        assert [(tuple(iv), list(v)) for iv, v in got] == want


REF_BIN = os.path.join(ROOT, "oracle", "_ref", "ref_mp4_index")


@pytest.mark.skipif(not os.path.exists(REF_BIN), reason="reference indexer not built here")
@pytest.mark.parametrize("path", [REALSHORT, COCKATOO])
def test_indexer_matches_reference_binary(hw, path):
    if not have_file(path):
        pytest.skip("sample file not in this image")
    ref = json.loads(subprocess.check_output([REF_BIN, path]))
    idx = hw.index_video(path)
    assert not ref["error"]
    assert idx.sample_offsets() == ref["sample_offsets"]
    assert idx.sample_sizes() == ref["sample_sizes"]
    assert idx.keyframe_indices() == ref["keyframes"]
    assert idx.metadata_bytes().hex() == ref["metadata"]
    assert (idx.timescale(), idx.duration(), idx.frame_width(), idx.frame_height(), idx.format()) == (
        ref["timescale"], ref["duration"], ref["width"], ref["height"], ref["format"])
    # pull protocol: same (offset, size) request sequence
    size = os.path.getsize(path)
    ind = hw.MP4IndexCreator(size)
    trace, off, n = [], 0, min(1024, size)
    with open(path, "rb") as f:
        while not ind.is_done():
            trace.append([off, n])
            f.seek(off)
            _, off, n = ind.feed(f.read(n), n)
    assert trace == ref["trace"]

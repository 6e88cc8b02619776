"""DecoderAutomata protocol, pinned with the STAMP fixture backend (decodes
nothing; stamps each frame with the fed sample's first four bytes), the way
SURVEY.md Appendix B-3 drove the reference automata with a counting fake.

Invariant from the reference's only results test
(decoder_automata_test.cpp:262-342): the frames returned for sparse rows are
the frames a sequential decode yields at those positions."""
import random
import struct

import pytest


def make_clip(hw, n, gop, gap=0, w=4, h=2):
    samples = [struct.pack("<I", i) + b"x" * (6 + i % 5) for i in range(n)]
    offs, sizes, blob, o = [], [], bytearray(), 0
    for i, s in enumerate(samples):
        if gap and i % gop == 0 and i:
            blob += b"\0" * gap
            o += gap
        offs.append(o)
        sizes.append(len(s))
        blob += s
        o += len(s)
    idx = hw.VideoIndex(timescale=1000, duration=n * 1000, width=w, height=h, format="avc1",
                        sample_offsets=offs, sample_sizes=sizes,
                        keyframe_indices=list(range(0, n, gop)), metadata=b"")
    return idx, bytes(blob)


def encoded_data(hw, idx, blob, plan):
    """Builds EncodedData exactly as python/hwang/decoder.py:36-63 does."""
    offs, sizes = idx.sample_offsets(), idx.sample_sizes()
    offs.append(offs[-1] + sizes[-1])
    sizes.append(0)
    out = []
    for (s, e), valid in plan:
        d = hw.EncodedData()
        d.width, d.height, d.format = idx.frame_width(), idx.frame_height(), idx.format()
        d.start_keyframe, d.end_keyframe = s, e
        start = offs[s]
        d.encoded_video = blob[start:offs[e] + sizes[e]]
        d.sample_offsets = [o - start for o in offs[s:e]]
        d.sample_sizes = sizes[s:e]
        d.valid_frames = valid
        d.keyframes = [k for k in idx.keyframe_indices() if s <= k <= e]
        out.append(d)
    return out


def stamps(frames):
    return [int.from_bytes(f.tobytes()[:4], "little") for f in frames]


def stamp_trace(hw):
    """Call counters of the STAMP fixture (process-wide; reset by fresh_automata)."""
    import ctypes
    t = (ctypes.c_int32 * 8)()
    hw.stamp_fixture.hwang_stamp_trace(t)
    keys = ("configures", "feeds", "keyframe_feeds", "empty_feeds", "flushes", "gets", "discards", "hints")
    return dict(zip(keys, list(t)))


def fresh_automata(hw):
    hw.stamp_fixture.hwang_stamp_reset()
    return hw.DecoderAutomata(hw.DeviceHandle(hw.DeviceType.CPU, 0), 1, hw.STAMP)


def test_reference_protocol_example(hw):
    # 4 touched GOPs of 10, wanted {3,4,27,55,58,99}
    idx, blob = make_clip(hw, 100, 10)
    rows = [3, 4, 27, 55, 58, 99]
    plan = hw.slice_into_video_intervals(idx, rows)
    a = fresh_automata(hw)
    a.initialize(encoded_data(hw, idx, blob, plan), b"")
    got = a.get_frames(idx, 4) + a.get_frames(idx, 2)
    assert stamps(got) == rows
    t = stamp_trace(hw)
    assert t["configures"] == 1
    assert t["gets"] == 6
    assert t["keyframe_feeds"] == 4          # one IDR per touched interval
    assert t["hints"] == 4
    # never more work than the reference (40 feeds, 34 discards), and an
    # end-of-interval marker + flush after every interval that was left
    assert t["feeds"] <= 40 and t["discards"] <= 34
    assert 3 <= t["flushes"] <= 4 and t["empty_feeds"] == t["flushes"]


@pytest.mark.parametrize("seed", range(6))
def test_sparse_equals_sequential(hw, seed):
    rnd = random.Random(seed)
    n, gop = rnd.randint(20, 150), rnd.randint(1, 17)
    idx, blob = make_clip(hw, n, gop, gap=rnd.choice([0, 5]))
    rows = sorted(rnd.sample(range(n), rnd.randint(1, min(n, 25))))
    plan = hw.slice_into_video_intervals(idx, rows)
    a = fresh_automata(hw)
    a.initialize(encoded_data(hw, idx, blob, plan), b"")
    got, left = [], len(rows)
    while left:
        k = rnd.randint(1, left)          # arbitrary get_frames chunking
        got += a.get_frames(idx, k)
        left -= k
    assert stamps(got) == rows
    assert a.stats()["frames_returned"] == len(rows)


def test_reinitialize_and_per_interval_loop(hw):
    # the reference's Python loop re-initializes per interval (decoder.py:65)
    idx, blob = make_clip(hw, 60, 6)
    rows = [0, 5, 6, 30, 59]
    a = fresh_automata(hw)
    got = []
    for seg in encoded_data(hw, idx, blob, hw.slice_into_video_intervals(idx, rows)):
        a.initialize([seg], b"")
        got += a.get_frames(idx, len(seg.valid_frames))
    assert stamps(got) == rows


def test_errors_are_results_not_aborts(hw):
    idx, blob = make_clip(hw, 20, 10)
    a = fresh_automata(hw)
    with pytest.raises(RuntimeError):
        a.get_frames(idx, 1)              # before initialize
    segs = encoded_data(hw, idx, blob, hw.slice_into_video_intervals(idx, [3]))
    a.initialize(segs, b"")
    assert stamps(a.get_frames(idx, 1)) == [3]
    with pytest.raises(RuntimeError):
        a.get_frames(idx, 1)              # nothing left
    segs[0].sample_offsets[2] = 10 ** 9   # sample outside encoded_video
    a.initialize(segs, b"")
    with pytest.raises(RuntimeError):
        a.get_frames(idx, 1)


def test_unsupported_backends_fail_loudly(hw):
    # no FFmpeg / NVDEC / CPU fallback in this build
    for t in (hw.VideoDecoderType.SOFTWARE, hw.VideoDecoderType.NVIDIA, hw.VideoDecoderType.INTEL):
        with pytest.raises(RuntimeError):
            hw.DecoderAutomata(hw.DeviceHandle(hw.DeviceType.CPU, 0), 1, t)
    with pytest.raises(RuntimeError):
        hw.DecoderAutomata(hw.DeviceHandle(hw.DeviceType.CPU, 0), 1, hw.VideoDecoderType.AMD)

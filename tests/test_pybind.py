"""The pybind11 module `hwang._python` (the reference's Python boundary,
hwang/hwang_python.cpp:102-167) over this repo's C++ classes, and the `hwang`
package on top of it (python/hwang/*.py).  CPU part: names, signatures, host
components against the same goldens as the C ABI, the automata protocol with
the STAMP fixture.  GPU part (-m gpu): the reference's Decoder.retrieve loop
with VideoDecoderType.AMD, frames bit-exact against the oracle."""
import io
import os
import struct

import numpy as np
import pytest

from tests.corpus_util import make_clip
from tests.util import REALSHORT, have_file


@pytest.fixture(scope="module")
def ref_api(built, hw):   # `hw` registers the STAMP fixture with the library both modules share
    import hwang
    return hwang


def test_module_surface(ref_api):
    m = ref_api._python
    assert m.__name__ == "hwang._python"
    for cls, methods in {
        "VideoIndex": ["deserialize", "serialize", "timescale", "duration", "fps", "frame_width", "frame_height", "format",
                       "frames", "sample_offsets", "sample_sizes", "keyframe_indices", "metadata_bytes"],
        "MP4IndexCreator": ["feed", "is_done", "is_error", "error_message", "get_video_index"],
        "DecoderAutomata": ["initialize", "get_frames"],
        "EncodedData": ["encoded_video", "width", "height", "format", "start_keyframe", "end_keyframe", "sample_offsets",
                        "sample_sizes", "keyframes", "valid_frames"],
        "DeviceHandle": ["type", "id"],
    }.items():
        for name in methods:
            assert hasattr(getattr(m, cls), name), (cls, name)
    assert int(m.DeviceType.CPU) == 0 and int(m.DeviceType.GPU) == 1
    # the reference's enumerators keep their values; AMD is appended (video_decoder_factory.h:22-26)
    assert [int(m.VideoDecoderType.SOFTWARE), int(m.VideoDecoderType.NVIDIA), int(m.VideoDecoderType.INTEL),
            int(m.VideoDecoderType.AMD)] == [0, 1, 2, 3]
    assert m.VideoDecoderType.AMD in m.supported_decoder_types()
    assert hasattr(ref_api.VideoIndex, "from_file") and hasattr(ref_api.VideoIndex, "to_file")


def test_no_cpu_fallback(ref_api):
    with pytest.raises(RuntimeError):
        ref_api.Decoder(io.BytesIO(b""), video_index=object(), device_type=ref_api.DeviceType.CPU)


@pytest.mark.skipif(not have_file(REALSHORT), reason="realshort.mp4 not in this image")
def test_index_and_planner_match_the_c_abi(ref_api, hw):
    a, b = ref_api.index_video(REALSHORT), hw.index_video(REALSHORT)
    assert a.serialize() == b.serialize()
    assert (a.frames(), a.keyframe_indices(), a.fps()) == (36, [0, 30], b.fps())
    buf = io.BytesIO()
    a.to_file(buf)
    buf.seek(0)
    assert ref_api.VideoIndex.from_file(buf).sample_offsets() == b.sample_offsets()
    for rows in ([0], [1, 2, 31], list(range(36)), [35]):
        got = ref_api.slice_into_video_intervals(a, rows)
        assert [((s, e), list(v)) for (s, e), v in got] == hw.slice_into_video_intervals(b, rows)
    with pytest.raises(RuntimeError):
        ref_api.slice_into_video_intervals(a, [36])


def test_reference_retrieve_loop_with_the_stamp_backend(ref_api, tmp_path):
    """python/hwang/decoder.py:30-69 over DecoderAutomata: k-th array == frame rows[k]."""
    n, gop, w, h = 240, 30, 6, 4
    samples = [struct.pack("<I", i) + b"s" * (5 + i % 7) for i in range(n)]
    path = str(tmp_path / "stamp.bin")
    gap = b"\xee" * 3   # bytes between GOPs: intervals never merge
    offs, blob = [], b""
    for i, s in enumerate(samples):
        if i % gop == 0:
            blob += gap
        offs.append(len(blob))
        blob += s
    open(path, "wb").write(blob)
    import hwang_amd
    seed = hwang_amd.VideoIndex(timescale=30000, duration=n * 1001, width=w, height=h, format="avc1", sample_offsets=offs,
                                sample_sizes=[len(s) for s in samples], keyframe_indices=list(range(0, n, gop)), metadata=b"")
    index = ref_api.VideoIndex.deserialize(seed.serialize())
    dec = ref_api.Decoder(path, video_index=index, device_type=ref_api.DeviceType.CPU,
                          decoder_type=ref_api.VideoDecoderType(100))
    rows = [0, 1, 29, 30, 95, 96, 200, 239]
    frames = dec.retrieve(rows)
    assert len(frames) == len(rows)
    for r, f in zip(rows, frames):
        assert f.shape == (h, w, 3) and f.dtype == np.uint8
        assert struct.unpack("<I", f.tobytes()[:4])[0] == r
    # frames of one call share one buffer and outlive the decoder
    del dec
    assert struct.unpack("<I", frames[3].tobytes()[:4])[0] == 30

    # An error in the middle of a request (here: at the last group of intervals but one, with the last one already waiting
    # in the queue) must come out of retrieve() at once: the helper thread that reads ahead is told to stop even while it
    # is waiting to hand over the end marker (it used to block there on a full queue, and retrieve() hung in its join -
    # found by the GPU fuzz test).
    import threading
    import time
    for group_bytes in (0, 64):
        dec = ref_api.Decoder(path, video_index=index, device_type=ref_api.DeviceType.CPU,
                              decoder_type=ref_api.VideoDecoderType(100), interval_group_bytes=group_bytes)
        inner, calls = dec._decoder, [0]
        ngroups = sum(1 for _ in dec._groups(rows))
        assert ngroups >= 3

        class FailsLater(object):
            def initialize(self, *a):
                return inner.initialize(*a)

            def get_frames(self, *a):
                calls[0] += 1
                if calls[0] >= ngroups - 1:
                    time.sleep(0.3)   # (the helper thread gets to the end of the plan meanwhile)
                    raise RuntimeError("decoder says no")
                return inner.get_frames(*a)
        dec._decoder = FailsLater()
        result = []

        def run():
            try:
                dec.retrieve(rows)
                result.append("returned")
            except RuntimeError as e:
                result.append(str(e))
        t = threading.Thread(target=run, daemon=True)
        t.start()
        t.join(20)
        assert not t.is_alive(), "retrieve() hangs after an error (interval_group_bytes=%d)" % group_bytes
        assert result == ["decoder says no"]


@pytest.mark.gpu
def test_reference_decoder_selects_the_amd_backend(ref_api, oracle, tmp_path):
    """What a hwang user runs after switching: hwang.Decoder(path, device_type=GPU).retrieve(rows)."""
    import torch
    assert torch.cuda.is_available()
    path = str(tmp_path / "clip.mp4")
    gold = make_clip(path, frames=36, gop=12, bframes=2, slices=2)
    dec = ref_api.Decoder(path, device_type=ref_api.DeviceType.GPU)
    rows = [0, 3, 11, 12, 13, 25, 35]
    frames = dec.retrieve(rows)
    for k, r in enumerate(rows):
        assert frames[k].shape == (240, 320, 3)
        assert np.array_equal(frames[k], oracle.i420_to_rgb(gold[r], 320, 240)), "row %d" % r
    # and the index the pybind indexer builds equals the file's
    assert ref_api.index_video(path).frames() == 36

"""Python surface of the reference's pybind11 module `hwang._python`
(hwang/hwang_python.cpp:102-167), re-hosted on the C ABI of libhwang_amd.so.

Same class / method / function names and argument meaning, so code written
against `hwang` runs against `hwang_amd` unchanged; the one addition is
`VideoDecoderType.AMD`; decoder types >= 100 are backends registered at run
time through `hwang_register_decoder_backend` (the test suite's STAMP fixture).
"""
import ctypes as C
import enum

import numpy as np

from . import _capi
from ._capi import lib


class DeviceType(enum.IntEnum):
    CPU = 0
    GPU = 1


class VideoDecoderType(enum.IntEnum):
    SOFTWARE = 0
    NVIDIA = 1
    INTEL = 2
    AMD = 3


class DeviceHandle(object):
    def __init__(self, type=DeviceType.CPU, id=0):
        self.type = type
        self.id = id


class VideoIndex(object):
    """hwang.VideoIndex (hwang_python.cpp:106-119)."""

    def __init__(self, _handle=None, timescale=0, duration=0, width=0, height=0,
                 format="", sample_offsets=(), sample_sizes=(),
                 keyframe_indices=(), metadata=b""):
        if _handle is not None:
            self._h = _handle
            return
        offs = _capi.u64_array(sample_offsets)
        sizes = _capi.u64_array(sample_sizes)
        kfs = _capi.u64_array(keyframe_indices)
        if len(offs) != len(sizes):
            raise ValueError("sample_offsets and sample_sizes differ in length")
        self._h = lib.hwang_video_index_create(
            timescale, duration, width, height, format.encode(), offs, sizes,
            len(offs), kfs, len(kfs), bytes(metadata), len(metadata))
        if not self._h:
            raise ValueError("VideoIndex: bad arguments")

    def problem(self):
        """'' when the index can be decoded from; otherwise what is wrong with it
        (such an index can be held and serialized -- the reference's own wire
        golden is one -- but the planner and the decoders refuse it)."""
        return lib.hwang_video_index_problem(self._h).decode()

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.hwang_video_index_destroy(h)

    @staticmethod
    def deserialize(data):
        data = bytes(data)
        h = lib.hwang_video_index_deserialize(data, len(data))
        if not h:
            raise RuntimeError("VideoIndex.deserialize: malformed data")
        return VideoIndex(_handle=h)

    def serialize(self):
        n = lib.hwang_video_index_serialize(self._h, None, 0)
        buf = (C.c_uint8 * n)()
        lib.hwang_video_index_serialize(self._h, buf, n)
        return bytes(buf)

    # hwang/python/hwang/video_index.py:5-15
    @staticmethod
    def from_file(f):
        return VideoIndex.deserialize(f.read())

    def to_file(self, f):
        f.write(self.serialize())

    def timescale(self):
        return lib.hwang_video_index_timescale(self._h)

    def duration(self):
        return lib.hwang_video_index_duration(self._h)

    def fps(self):
        return lib.hwang_video_index_fps(self._h)

    def frame_width(self):
        return lib.hwang_video_index_frame_width(self._h)

    def frame_height(self):
        return lib.hwang_video_index_frame_height(self._h)

    def format(self):
        return lib.hwang_video_index_format(self._h).decode()

    def frames(self):
        return lib.hwang_video_index_frames(self._h)

    def _u64(self, ptr, n):
        if n == 0:
            return np.zeros(0, dtype=np.uint64)
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy()

    def sample_offsets_np(self):
        return self._u64(lib.hwang_video_index_sample_offsets(self._h), self.frames())

    def sample_sizes_np(self):
        return self._u64(lib.hwang_video_index_sample_sizes(self._h), self.frames())

    def sample_offsets(self):
        return self.sample_offsets_np().tolist()

    def sample_sizes(self):
        return self.sample_sizes_np().tolist()

    def keyframe_indices(self):
        n = lib.hwang_video_index_num_keyframes(self._h)
        return self._u64(lib.hwang_video_index_keyframe_indices(self._h), n).tolist()

    def metadata_bytes(self):
        n = lib.hwang_video_index_metadata_size(self._h)
        if n == 0:
            return b""
        return C.string_at(lib.hwang_video_index_metadata_bytes(self._h), n)


class MP4IndexCreator(object):
    """hwang.MP4IndexCreator (hwang_python.cpp:121-127)."""

    def __init__(self, file_size):
        self._h = lib.hwang_mp4_indexer_create(file_size)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.hwang_mp4_indexer_destroy(h)

    def feed(self, data, size):
        data = bytes(data)
        off = C.c_uint64(0)
        sz = C.c_uint64(0)
        ret = lib.hwang_mp4_indexer_feed(self._h, data, min(size, len(data)),
                                         C.byref(off), C.byref(sz))
        return bool(ret), off.value, sz.value

    def is_done(self):
        return bool(lib.hwang_mp4_indexer_is_done(self._h))

    def is_error(self):
        return bool(lib.hwang_mp4_indexer_is_error(self._h))

    def error_message(self):
        return lib.hwang_mp4_indexer_error_message(self._h).decode()

    def get_video_index(self):
        return VideoIndex(_handle=lib.hwang_mp4_indexer_get_video_index(self._h))


def slice_into_video_intervals(index, rows):
    """-> [((start, end), [frames...]), ...] (hwang_python.cpp:38-49)."""
    arr = _capi.u64_array(rows)
    h = lib.hwang_slice_into_video_intervals(index._h, arr, len(arr))
    if not h:
        raise RuntimeError("slice_into_video_intervals: row out of range")
    try:
        out = []
        for i in range(lib.hwang_video_intervals_count(h)):
            s, e = C.c_uint64(), C.c_uint64()
            lib.hwang_video_intervals_range(h, i, C.byref(s), C.byref(e))
            n = lib.hwang_video_intervals_num_valid(h, i)
            p = lib.hwang_video_intervals_valid(h, i)
            out.append(((s.value, e.value), [p[k] for k in range(n)]))
        return out
    finally:
        lib.hwang_video_intervals_destroy(h)


class EncodedData(object):
    """hwang.EncodedData (hwang_python.cpp:144-161)."""

    def __init__(self):
        self.encoded_video = b""
        self.width = 0
        self.height = 0
        self.format = ""
        self.start_keyframe = 0
        self.end_keyframe = 0
        self.sample_offsets = []
        self.sample_sizes = []
        self.keyframes = []
        self.valid_frames = []


class DecoderAutomata(object):
    """hwang.DecoderAutomata (hwang_python.cpp:163-166)."""

    def __init__(self, handle, num_devices, decoder_type):
        self._h = lib.hwang_automata_create(int(handle.type), int(handle.id),
                                            int(num_devices), int(decoder_type))
        if not self._h:
            raise RuntimeError(
                "DecoderAutomata: backend %r is not available on device "
                "(%r, %d) -- no fallback exists" %
                (decoder_type, handle.type, handle.id))

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.hwang_automata_destroy(h)

    def initialize(self, encoded_data, extradata):
        n = len(encoded_data)
        arr = (_capi.EncodedDataStruct * n)()
        keep = []
        for i, d in enumerate(encoded_data):
            video = bytes(d.encoded_video)
            offs = _capi.u64_array(d.sample_offsets)
            sizes = _capi.u64_array(d.sample_sizes)
            kfs = _capi.u64_array(d.keyframes)
            valid = _capi.u64_array(d.valid_frames)
            if len(offs) != len(sizes):
                raise RuntimeError("EncodedData: sample_offsets/sample_sizes mismatch")
            fmt = d.format.encode()
            keep += [video, offs, sizes, kfs, valid, fmt]
            s = arr[i]
            s.encoded_video = C.cast(C.c_char_p(video), C.c_void_p)
            s.encoded_video_size = len(video)
            s.width, s.height = d.width, d.height
            s.start_keyframe, s.end_keyframe = d.start_keyframe, d.end_keyframe
            s.format = fmt
            s.sample_offsets, s.sample_sizes, s.num_samples = offs, sizes, len(offs)
            s.keyframes, s.num_keyframes = kfs, len(kfs)
            s.valid_frames, s.num_valid_frames = valid, len(valid)
        extradata = bytes(extradata)
        if lib.hwang_automata_initialize(self._h, arr, n, extradata, len(extradata)):
            raise RuntimeError(lib.hwang_automata_last_error(self._h).decode())

    def get_frames(self, index, num_frames):
        h, w = index.frame_height(), index.frame_width()
        buf = np.empty((num_frames, h, w, 3), dtype=np.uint8)
        if lib.hwang_automata_get_frames(self._h, buf.ctypes.data, num_frames):
            raise RuntimeError(lib.hwang_automata_last_error(self._h).decode())
        return [buf[i] for i in range(num_frames)]

    # -- not in the reference ------------------------------------------------
    def stats(self):
        s = (C.c_int64 * 4)()
        lib.hwang_automata_stats(self._h, s)
        return dict(samples_fed=s[0], frames_discarded=s[1],
                    frames_returned=s[2], flushes=s[3])

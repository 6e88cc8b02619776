"""hwang.Decoder (python/hwang/decoder.py:5-69) on the MI355X backend.

`retrieve(rows)` keeps the reference's meaning -- the k-th returned array is
frame rows[k] as H x W x 3 uint8 RGB -- but a GPU Decoder hands the whole
request to the library's parallel retriever (keyframe-anchored intervals
fanned out over lanes whose pictures are batched into wide GPU launches)
instead of looping one
DecoderAutomata over the intervals.  `retrieve_serial` walks the intervals
through one automata the way the reference does, for parity tests of that
path.
"""
import ctypes as C
import mmap

import numpy as np

from . import _capi
from ._capi import lib
from .api import (DecoderAutomata, DeviceHandle, DeviceType, EncodedData,
                  VideoDecoderType, slice_into_video_intervals)


def _host_view(f):
    """The whole file as (ctypes source, length, keep-alive)."""
    try:
        view = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    except (AttributeError, OSError, ValueError):
        f.seek(0, 0)
        view = f.read()
    if isinstance(view, bytes):
        return view, len(view), view
    arr = np.frombuffer(view, dtype=np.uint8)
    return C.c_void_p(arr.ctypes.data), len(view), (arr, view)


class BatchRetriever(object):
    """Many clips, one call (BASELINE config C3: 128 clips, 1 % of each): the
    GOP work items of all requests share one queue that the lanes of every
    device pull from.  `device_ids` lists the GPUs of this process; with one
    process per GPU pass one id and `shard=(rank, world)`."""

    def __init__(self, device_ids=(0,), device_type=DeviceType.GPU, decoder_type=None, num_lanes=0, shard=None):
        if decoder_type is None:
            decoder_type = VideoDecoderType.AMD if device_type == DeviceType.GPU else VideoDecoderType.SOFTWARE
        if not lib.hwang_decoder_type_supported(int(decoder_type)):
            raise RuntimeError("hwang_amd.BatchRetriever: decoder type %r is not built into libhwang_amd.so "
                               "(there is no CPU fallback)" % (decoder_type,))
        ids = (C.c_int32 * len(device_ids))(*device_ids)
        self._h = lib.hwang_retriever_create_multi(int(device_type), ids, len(device_ids), int(decoder_type), num_lanes)
        if not self._h:
            raise RuntimeError("hwang_amd.BatchRetriever: cannot open backend %r on devices %r" %
                               (decoder_type, list(device_ids)))
        self.num_devices = len(device_ids)
        self._views = {}   # file object -> its mapping: one mmap for the retriever's lifetime (a fresh one per
                           # call would fault every touched page in again)
        if shard is not None and lib.hwang_retriever_set_shard(self._h, int(shard[0]), int(shard[1])):
            raise RuntimeError(lib.hwang_retriever_last_error(self._h).decode())

    def num_lanes(self):
        return lib.hwang_retriever_num_lanes(self._h)

    def retrieve_into(self, requests, yuv=False):
        """requests: [(video_index, file object or path, rows, out_ptr, out_capacity)]"""
        arr = (_capi.RetrieveRequestStruct * len(requests))()
        keep = []
        for i, (index, f, rows, out_ptr, cap) in enumerate(requests):
            if isinstance(f, str):
                key = ("path", f)
                fobj = self._views.get(key)
                if fobj is None:   # open only on a miss (setdefault(open(...)) opened - and leaked - a handle per call)
                    fobj = self._views[key] = open(f, "rb")
                f = fobj
            view = self._views.get(id(f))
            if view is None:
                view = self._views[id(f)] = (f,) + _host_view(f)
            _, src, n, k = view
            rows_arr = _capi.u64_array(rows)
            keep.append((f, k, rows_arr, src))
            arr[i].index = index._h
            arr[i].file = C.cast(src, C.c_void_p) if not isinstance(src, C.c_void_p) else src
            arr[i].file_size = n
            arr[i].rows = rows_arr
            arr[i].num_rows = len(rows_arr)
            arr[i].out = out_ptr
            arr[i].out_capacity = cap
        rc = lib.hwang_retriever_retrieve_batch(self._h, arr, len(requests), 1 if yuv else 0)
        del keep
        if rc:
            raise RuntimeError(lib.hwang_retriever_last_error(self._h).decode())

    def retrieve(self, requests):
        """requests: [(video_index, file or path, rows)] -> per request a list of H x W x 3 arrays."""
        bufs, full = [], []
        for index, f, rows in requests:
            rows = list(rows)
            buf = np.zeros((len(rows), index.frame_height(), index.frame_width(), 3), dtype=np.uint8)
            bufs.append(buf)
            full.append((index, f, rows, buf.ctypes.data, buf.nbytes))
        self.retrieve_into(full)
        return [[b[i] for i in range(b.shape[0])] for b in bufs]

    def host_counters(self):
        """What the lane threads of the last call executed (hwang_retriever_host_counters: all zeros but cpu_ns unless
        HWANG_HOST_COUNTERS=1 was set and the kernel lets the process count its own events)."""
        s = (C.c_int64 * 4)()
        lib.hwang_retriever_host_counters(self._h, s)
        return dict(zip(("instructions", "cycles", "branch_misses", "cpu_ns"), list(s)))

    def stats(self):
        s = (C.c_int64 * 10)()
        lib.hwang_retriever_stats(self._h, s)
        keys = ("intervals", "samples_fed", "pictures_reconstructed", "pictures_skipped", "frames_returned",
                "bytes_uploaded", "entropy_ns", "wall_ns", "stall_ns", "lane_threads")
        return dict(zip(keys, list(s)))

    def device_stats(self):
        out = []
        for d in range(self.num_devices):
            s = (C.c_int64 * 4)()
            lib.hwang_retriever_device_stats(self._h, d, s)
            out.append(dict(zip(("lanes", "items", "frames_returned", "samples_fed"), list(s))))
        return out

    def close(self):
        """Destroys the native retriever and lets go of the files opened by path (mappings are dropped with
        their last view)."""
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.hwang_retriever_destroy(h)
        views, self._views = getattr(self, "_views", {}), {}
        for key, v in views.items():
            if isinstance(key, tuple) and key[:1] == ("path",):
                try:
                    v.close()
                except (OSError, ValueError, BufferError):
                    pass

    def __del__(self):
        self.close()


class Decoder(object):
    def __init__(self, f_or_path, video_index=None, device_type=DeviceType.GPU,
                 device_id=0, decoder_type=None, num_lanes=0, device_ids=None, shard=None):
        import hwang_amd
        if video_index is None:
            video_index = hwang_amd.index_video(f_or_path)
        self.video_index = video_index
        if isinstance(f_or_path, str):
            f = open(f_or_path, 'rb')
        else:
            f = f_or_path
        self.f = f

        if decoder_type is None:
            # The reference hard-wires GPU -> NVIDIA, CPU -> SOFTWARE
            # (decoder.py:24-27).  Here GPU -> AMD; a CPU handle has no
            # backend in this build and fails below.
            decoder_type = (VideoDecoderType.AMD if device_type == DeviceType.GPU
                            else VideoDecoderType.SOFTWARE)
        self.device_type = device_type
        self.device_id = device_id
        # several GPUs of this process (one request's GOP items fan out over all of them), and / or this
        # process's share of a job that runs one process per GPU
        self.device_ids = list(device_ids) if device_ids is not None else [device_id]
        self.shard = shard
        self.decoder_type = decoder_type
        self.num_lanes = num_lanes
        self._automata = None
        self._retriever = None
        if not lib.hwang_decoder_type_supported(int(decoder_type)):
            raise RuntimeError(
                "hwang_amd.Decoder: decoder type %r is not built into "
                "libhwang_amd.so (only AMD on a GPU device; there is no CPU "
                "fallback)" % (decoder_type,))

    # -- one DecoderAutomata, one interval at a time ---------------------------
    def _get_automata(self):
        if self._automata is None:
            self._automata = DecoderAutomata(DeviceHandle(self.device_type, self.device_id), 1, self.decoder_type)
        return self._automata

    def _interval(self, bounds, first, last, wanted):
        """EncodedData of the planner interval [first, last): its bytes, its samples relative to them, the sync
        samples inside it (the interval's end included, as the planner counts) and the frames wanted of it."""
        begin, end = int(bounds[first]), int(bounds[last])
        self.f.seek(begin)
        piece = EncodedData()
        piece.encoded_video = self.f.read(end - begin)
        piece.width, piece.height = self.video_index.frame_width(), self.video_index.frame_height()
        piece.format = self.video_index.format()
        piece.start_keyframe, piece.end_keyframe = first, last
        piece.sample_offsets = [int(o) - begin for o in bounds[first:last]]
        piece.sample_sizes = self._sizes[first:last]
        piece.keyframes = [k for k in self._keyframes if first <= k <= last]
        piece.valid_frames = wanted
        return piece

    def retrieve_serial(self, rows):
        """The shape of the reference's Decoder.retrieve (python/hwang/decoder.py:30-69) on the ctypes surface: the
        seek plan's intervals go through ONE automata one after the other, each with an initialize() and a
        get_frames() of its own.  Kept for the parity tests of the automata path; `retrieve` is the product."""
        index = self.video_index
        offsets = index.sample_offsets()
        self._sizes = index.sample_sizes()
        self._keyframes = index.keyframe_indices()
        bounds = list(offsets) + [offsets[-1] + self._sizes[-1]]   # bounds[k]: where sample k starts / the file's samples end
        extradata = index.metadata_bytes()
        automata = self._get_automata()
        out = []
        for (first, last), wanted in slice_into_video_intervals(index, rows):
            if not wanted:   # (the planner answers an empty request with one interval that wants nothing)
                continue
            automata.initialize([self._interval(bounds, first, last, wanted)], extradata)
            out.extend(automata.get_frames(index, len(wanted)))
        return out

    # -- parallel path ---------------------------------------------------------
    def _get_retriever(self):
        if self._retriever is None:
            ids = (C.c_int32 * len(self.device_ids))(*self.device_ids)
            h = lib.hwang_retriever_create_multi(int(self.device_type), ids, len(self.device_ids),
                                                 int(self.decoder_type), self.num_lanes)
            if not h:
                raise RuntimeError(
                    "hwang_amd.Decoder: cannot open backend %r on devices %r" %
                    (self.decoder_type, self.device_ids))
            if self.shard is not None and lib.hwang_retriever_set_shard(h, int(self.shard[0]), int(self.shard[1])):
                raise RuntimeError(lib.hwang_retriever_last_error(h).decode())
            self._retriever = h
        return self._retriever

    def _file_view(self):
        # one mapping for the Decoder's lifetime: a fresh mmap per call would
        # fault every touched page in again (120 k faults per 1 %-sparse pass
        # over a 1.8 GB clip)
        view = getattr(self, "_view", None)
        if view is None:
            try:
                view = mmap.mmap(self.f.fileno(), 0, access=mmap.ACCESS_READ)
            except (AttributeError, OSError, ValueError):
                self.f.seek(0, 0)
                view = self.f.read()
            self._view = view
        return view

    def retrieve_into(self, rows, out_ptr, out_capacity, yuv=False):
        """Decode `rows` (sorted) into caller memory (host or HBM address)."""
        r = self._get_retriever()
        view = self._file_view()
        n = len(view)
        if isinstance(view, bytes):
            src, keep = view, view
        else:
            keep = (C.c_char * n).from_buffer_copy(view) if False else None
            arr = np.frombuffer(view, dtype=np.uint8)
            src, keep = C.c_void_p(arr.ctypes.data), arr
        rows_arr = _capi.u64_array(rows)
        rc = lib.hwang_retriever_retrieve(r, self.video_index._h, src, n, rows_arr,
                                          len(rows_arr), out_ptr, out_capacity,
                                          1 if yuv else 0)
        del keep
        if rc:
            raise RuntimeError(lib.hwang_retriever_last_error(r).decode())

    def retrieve(self, rows):
        rows = list(rows)
        h, w = self.video_index.frame_height(), self.video_index.frame_width()
        buf = np.empty((len(rows), h, w, 3), dtype=np.uint8)
        if rows:
            self.retrieve_into(rows, buf.ctypes.data, buf.nbytes)
        return [buf[i] for i in range(len(rows))]

    def retrieve_yuv(self, rows):
        """Planar I420 frames (the normative decoder output), one 1-D array each."""
        rows = list(rows)
        h, w = self.video_index.frame_height(), self.video_index.frame_width()
        fs = w * h + 2 * ((w + 1) // 2) * ((h + 1) // 2)
        buf = np.empty((len(rows), fs), dtype=np.uint8)
        if rows:
            self.retrieve_into(rows, buf.ctypes.data, buf.nbytes, yuv=True)
        return [buf[i] for i in range(len(rows))]

    def stats(self):
        s = (C.c_int64 * 10)()
        lib.hwang_retriever_stats(self._get_retriever(), s)
        keys = ("intervals", "samples_fed", "pictures_reconstructed",
                "pictures_skipped", "frames_returned", "bytes_uploaded",
                "entropy_ns", "wall_ns", "stall_ns", "lane_threads")
        return dict(zip(keys, list(s)))

    def device_stats(self):
        out = []
        for d in range(len(self.device_ids)):
            s = (C.c_int64 * 4)()
            lib.hwang_retriever_device_stats(self._get_retriever(), d, s)
            out.append(dict(zip(("lanes", "items", "frames_returned", "samples_fed"), list(s))))
        return out

    def __del__(self):
        h, self._retriever = getattr(self, "_retriever", None), None
        if h:
            lib.hwang_retriever_destroy(h)

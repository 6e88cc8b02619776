"""TEST INFRASTRUCTURE: ctypes wrapper of the CPU oracle decoder
(oracle/_build/liboracle_h264.so).  Only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle_h264.so")


def load():
    if not os.path.exists(_LIB):
        raise ImportError("oracle not built: run `make -C oracle`")
    lib = C.CDLL(_LIB)
    lib.h264o_open.restype = C.c_void_p
    lib.h264o_close.argtypes = [C.c_void_p]
    lib.h264o_configure.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.h264o_decode.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.h264o_flush.argtypes = [C.c_void_p]
    lib.h264o_frames_ready.argtypes = [C.c_void_p]
    lib.h264o_frame_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.h264o_pop_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.h264o_error.argtypes = [C.c_void_p]
    lib.h264o_error.restype = C.c_char_p
    lib.h264o_counters.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.h264o_yuv420_to_rgb24.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return lib


class OracleDecoder(object):
    def __init__(self, extradata):
        self.lib = load()
        self.h = self.lib.h264o_open()
        extradata = bytes(extradata)
        if self.lib.h264o_configure(self.h, extradata, len(extradata)):
            raise RuntimeError("oracle configure: " + self.error())

    def error(self):
        return self.lib.h264o_error(self.h).decode()

    def close(self):
        if self.h:
            self.lib.h264o_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _drain(self, out):
        w, h, poc = C.c_int(), C.c_int(), C.c_int()
        while self.lib.h264o_frames_ready(self.h) > 0:
            self.lib.h264o_frame_info(self.h, C.byref(w), C.byref(h), C.byref(poc))
            n = w.value * h.value + 2 * ((w.value + 1) // 2) * ((h.value + 1) // 2)
            buf = np.empty(n, dtype=np.uint8)
            self.lib.h264o_pop_frame(self.h, buf.ctypes.data, n)
            out.append((buf, w.value, h.value, poc.value))

    def decode(self, sample):
        """Feed one sample; returns the list of frames that became ready,
        each (i420 ndarray, width, height, poc)."""
        sample = bytes(sample)
        if self.lib.h264o_decode(self.h, sample, len(sample)):
            raise RuntimeError("oracle decode: " + self.error())
        out = []
        self._drain(out)
        return out

    def flush(self):
        if self.lib.h264o_flush(self.h):
            raise RuntimeError("oracle flush: " + self.error())
        out = []
        self._drain(out)
        return out

    def counters(self):
        c = (C.c_int64 * 10)()
        self.lib.h264o_counters(self.h, c)
        keys = ("slices", "slices_clean_end", "macroblocks", "pictures", "i_slices",
                "p_slices", "b_slices", "intra_mbs", "skip_mbs", "t8x8_mbs")
        return dict(zip(keys, list(c)))


def i420_to_rgb(i420, w, h):
    lib = load()
    rgb = np.empty((h, w, 3), dtype=np.uint8)
    i420 = np.ascontiguousarray(i420)
    lib.h264o_yuv420_to_rgb24(i420.ctypes.data, w, h, rgb.ctypes.data)
    return rgb


def decode_file(path, index, first=0, count=None):
    """Sequentially decodes samples [first, first+count) of an indexed MP4;
    returns frames in display order."""
    offs, sizes = index.sample_offsets(), index.sample_sizes()
    n = index.frames() if count is None else count
    dec = OracleDecoder(index.metadata_bytes())
    frames = []
    with open(path, "rb") as f:
        for i in range(first, first + n):
            f.seek(offs[i])
            frames += dec.decode(f.read(sizes[i]))
    frames += dec.flush()
    return frames, dec.counters()

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
    lib.h264o_set_parse_only.argtypes = [C.c_void_p, C.c_int]
    lib.h264o_coverage_enable.argtypes = [C.c_void_p]
    lib.h264o_coverage.argtypes = [C.c_void_p, C.POINTER(C.c_int64), C.c_int]
    lib.h264o_yuv420_to_rgb24.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    return lib


# coverage histogram sections: name -> (base, cells); the layout is documented in h264_oracle.h
COV_SECTIONS = {}
_off = 0
for _name, _n in (("i4", 144), ("i8", 144), ("i16", 16), ("ic", 16), ("luma_mc", 48), ("chroma_mc", 64), ("wp", 9),
                  ("bs", 60), ("filt", 12), ("mbtype_i", 26), ("mbtype_p", 32), ("mbtype_b", 50), ("sub_p", 4),
                  ("sub_b", 13), ("coeff_token", 340), ("total_zeros", 240), ("total_zeros_dc", 12),
                  ("run_before", 105), ("escape", 5), ("dequant", 7), ("direct", 8), ("ctx", 7 * 460 * 2)):
    COV_SECTIONS[_name] = (_off, _n)
    _off += _n
COV_CELLS = _off


class OracleDecoder(object):
    def __init__(self, extradata, coverage=False, parse_only=False):
        self.lib = load()
        self.h = self.lib.h264o_open()
        if coverage:
            self.lib.h264o_coverage_enable(self.h)
        if parse_only:   # syntax and entropy decoding only, nothing reconstructed; accepts 4:4:4 (h264_oracle.h)
            self.lib.h264o_set_parse_only(self.h, int(parse_only))   # (2: without the old-x264 emulation)
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


def _coverage(self):
    """The coverage histogram as an int64 array of COV_CELLS cells (see COV_SECTIONS)."""
    out = np.zeros(COV_CELLS, dtype=np.int64)
    n = self.lib.h264o_coverage(self.h, out.ctypes.data_as(C.POINTER(C.c_int64)), COV_CELLS)
    assert n == COV_CELLS, "oracle_py.COV_SECTIONS is out of step with h264_oracle.h"
    return out


OracleDecoder.coverage = _coverage


def i420_to_rgb(i420, w, h):
    lib = load()
    rgb = np.empty((h, w, 3), dtype=np.uint8)
    i420 = np.ascontiguousarray(i420)
    lib.h264o_yuv420_to_rgb24(i420.ctypes.data, w, h, rgb.ctypes.data)
    return rgb


def decode_file(path, index, first=0, count=None, coverage=None, parse_only=False):
    """Sequentially decodes samples [first, first+count) of an indexed MP4;
    returns frames in display order.  `coverage`: an int64 array of COV_CELLS
    cells that this decode's histogram is added to."""
    offs, sizes = index.sample_offsets(), index.sample_sizes()
    n = index.frames() if count is None else count
    dec = OracleDecoder(index.metadata_bytes(), coverage=coverage is not None, parse_only=parse_only)
    frames = []
    with open(path, "rb") as f:
        for i in range(first, first + n):
            f.seek(offs[i])
            frames += dec.decode(f.read(sizes[i]))
    frames += dec.flush()
    if coverage is not None:
        coverage += dec.coverage()
    return frames, dec.counters()


# ---------------------------------------------------------------------------------------------------------- HEVC
_HEVC_LIB = os.path.join(_HERE, "_build", "liboracle_hevc.so")


def load_hevc():
    if not os.path.exists(_HEVC_LIB):
        raise ImportError("HEVC oracle not built: run `make -C oracle`")
    lib = C.CDLL(_HEVC_LIB)
    lib.hevco_open.restype = C.c_void_p
    lib.hevco_close.argtypes = [C.c_void_p]
    lib.hevco_configure.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.hevco_decode.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.hevco_flush.argtypes = [C.c_void_p]
    lib.hevco_frames_ready.argtypes = [C.c_void_p]
    lib.hevco_frame_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.hevco_pop_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    lib.hevco_error.argtypes = [C.c_void_p]
    lib.hevco_error.restype = C.c_char_p
    lib.hevco_counters.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.hevco_set_parse_only.argtypes = [C.c_void_p, C.c_int]
    lib.hevco_residual_block.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.hevco_intra_block.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    lib.hevco_inter_block.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p]
    lib.hevco_deblock_luma.argtypes = [C.c_void_p] + [C.c_int] * 7
    lib.hevco_deblock_chroma.argtypes = [C.c_void_p] + [C.c_int] * 4
    lib.hevco_sao_block.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return lib


class HevcOracleDecoder(object):
    """The H.264 wrapper's interface over oracle/hevc_*.c (hvcC extradata, length-prefixed samples)."""

    def __init__(self, extradata, parse_only=False):
        self.lib = load_hevc()
        self.h = self.lib.hevco_open()
        if parse_only:
            self.lib.hevco_set_parse_only(self.h, 1)
        extradata = bytes(extradata)
        if self.lib.hevco_configure(self.h, extradata, len(extradata)):
            raise RuntimeError("HEVC oracle configure: " + self.error())

    def error(self):
        return self.lib.hevco_error(self.h).decode()

    def close(self):
        if self.h:
            self.lib.hevco_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _drain(self, out):
        w, h, poc = C.c_int(), C.c_int(), C.c_int()
        while self.lib.hevco_frames_ready(self.h) > 0:
            self.lib.hevco_frame_info(self.h, C.byref(w), C.byref(h), C.byref(poc))
            n = w.value * h.value + 2 * ((w.value + 1) // 2) * ((h.value + 1) // 2)
            buf = np.empty(n, dtype=np.uint8)
            self.lib.hevco_pop_frame(self.h, buf.ctypes.data, n)
            out.append((buf, w.value, h.value, poc.value))

    def decode(self, sample):
        sample = bytes(sample)
        if self.lib.hevco_decode(self.h, sample, len(sample)):
            raise RuntimeError("HEVC oracle decode: " + self.error())
        out = []
        self._drain(out)
        return out

    def flush(self):
        if self.lib.hevco_flush(self.h):
            raise RuntimeError("HEVC oracle flush: " + self.error())
        out = []
        self._drain(out)
        return out

    def counters(self):
        c = (C.c_int64 * 12)()
        self.lib.hevco_counters(self.h, c)
        keys = ("slices", "slices_clean_end", "coding_units", "pictures", "i_slices", "p_slices", "b_slices", "intra_cus",
                "skip_cus", "md5_checked", "md5_mismatch", "transform_units")
        return dict(zip(keys, list(c)))


def decode_file_hevc(path, index, first=0, count=None, parse_only=False):
    """Sequentially decodes samples [first, first+count) of an indexed MP4 with an HEVC track; frames in display order."""
    offs, sizes = index.sample_offsets(), index.sample_sizes()
    n = index.frames() if count is None else count
    dec = HevcOracleDecoder(index.metadata_bytes(), parse_only=parse_only)
    frames = []
    with open(path, "rb") as f:
        for i in range(first, first + n):
            f.seek(offs[i])
            frames += dec.decode(f.read(sizes[i]))
    frames += dec.flush()
    return frames, dec.counters()

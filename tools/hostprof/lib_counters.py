#!/usr/bin/env python3
"""Development aid: the host parser of one or more parser libraries (build_variant.sh, or the product library) on a
C1-shaped clip, one thread: wall time per picture and - where the kernel lets a process count its own events -
instructions / cycles / branch misses per picture kind (HWANG_PROBE_TIMES, capi_probe.cpp).
usage: lib_counters.py clip.mp4 lib_a.so [lib_b.so ...]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["HWANG_PROBE_FAST"] = "1"
os.environ["HWANG_PROBE_TIMES"] = "1"
import hwang_amd as hw  # noqa: E402

path = sys.argv[1]
idx = hw.index_video(path)
data = np.fromfile(path, dtype=np.uint8)
offs = idx.sample_offsets_np().astype(np.uint64)
sizes = idx.sample_sizes_np().astype(np.uint64)
avcc = bytes(idx.metadata_bytes())
gop = 60
# every GOP up to its last P picture, B pictures skipped (what C1 mostly decodes) - or, ALL=1, every picture
wanted = np.arange(gop - 3, idx.frames(), gop, dtype=np.uint64) if not os.environ.get("ALL") else np.arange(idx.frames(), dtype=np.uint64)
if os.environ.get("WANTED") == "keys":  # the keyframes alone (what follows each in its GOP serves nobody and is not parsed)
    wanted = np.arange(0, idx.frames(), gop, dtype=np.uint64)
if os.environ.get("WANTED") == "p":  # the last P picture of every GOP: no B picture at all
    wanted = np.arange(gop - 3, idx.frames(), gop, dtype=np.uint64)
if os.environ.get("WANTED") == "b":  # one B picture per GOP, the last: every P picture before it and that one B picture
    wanted = np.arange(gop - 2, idx.frames(), gop, dtype=np.uint64)
keys = np.arange(0, idx.frames(), gop, dtype=np.uint64)  # (the clip's closed GOPs; only used with WANTED set)
for libpath in sys.argv[2:]:
    lib = C.CDLL(os.path.abspath(libpath))
    lib.hwang_h264_probe_workers.restype = C.c_int
    lib.hwang_h264_probe_workers.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                             C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_int64), C.c_char_p, C.c_size_t]
    print("==", libpath, flush=True)
    best = 1e9
    for rep in range(int(os.environ.get("REPS", "3"))):
        out = (C.c_int64 * 9)()
        err = C.create_string_buffer(256)
        t = time.time()
        nk = len(keys) if os.environ.get("WANTED") else 0
        rc = lib.hwang_h264_probe_workers(avcc, len(avcc), data.ctypes.data, offs.ctypes.data, sizes.ctypes.data, len(offs), wanted.ctypes.data,
                                          len(wanted), keys.ctypes.data if nk else None, nk, 0, out, err, 256)
        dt = time.time() - t
        if rc:
            raise SystemExit(err.value)
        best = min(best, dt)
    print("   %d pictures decoded: best %.3f ms per picture (digest %x)" % (out[2], 1e3 * best / out[2], out[7]), flush=True)

#!/usr/bin/env python3
"""Development aid: scaling of the host-side H.264 parser alone (no GPU):
N threads, each entropy-decoding GOPs of the probe clip through
hwang_h264_probe; prints pictures/s per N."""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hwang_amd as hw  # noqa: E402
from hwang_amd._capi import lib  # noqa: E402

path = sys.argv[1]
idx = hw.index_video(path)
data = np.fromfile(path, dtype=np.uint8)
offs = idx.sample_offsets_np().astype(np.uint64)
sizes = idx.sample_sizes_np().astype(np.uint64)
kf = list(idx.keyframe_indices()) + [idx.frames()]
avcc = bytes(idx.metadata_bytes())
os.environ["HWANG_PROBE_FAST"] = "1"
GOPS = int(os.environ.get("GOPS", 2))


def work(tid, res):
    out = (C.c_int64 * 8)()
    err = C.create_string_buffer(256)
    # ONE probe call (one parser, one set of buffers) over GOPS consecutive GOPs
    k = (tid * GOPS) % max(1, len(kf) - GOPS)
    a, b = kf[k], kf[k + GOPS]
    o = np.ascontiguousarray(offs[a:b])
    s = np.ascontiguousarray(sizes[a:b])
    rc = lib.hwang_h264_probe(avcc, len(avcc), data.ctypes.data_as(C.POINTER(C.c_uint8)),
                              o.ctypes.data_as(C.POINTER(C.c_uint64)), s.ctypes.data_as(C.POINTER(C.c_uint64)),
                              b - a, None, 0, out, err, 256)
    if rc:
        raise RuntimeError(err.value)
    n = b - a
    res[tid] = n


for nthreads in [int(x) for x in os.environ.get("THREADS", "1,16,32,64,128").split(",")]:
    res = [0] * nthreads
    th = [threading.Thread(target=work, args=(i, res)) for i in range(nthreads)]
    t = time.time()
    for x in th:
        x.start()
    for x in th:
        x.join()
    dt = time.time() - t
    print("threads %3d: %6d pictures in %.3f s -> %8.0f pics/s, %.2f ms per picture per thread" % (
        nthreads, sum(res), dt, sum(res) / dt, 1e3 * dt * nthreads / sum(res)))

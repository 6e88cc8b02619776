#!/usr/bin/env python3
"""Development aid: lib_counters.py's measurement on N threads at once (each thread its own parser over the same clip, its
own hardware counters): what sharing the machine costs the parser in cycles per picture.
usage: lib_counters_mt.py clip.mp4 lib.so N [N ...]"""
import ctypes as C
import os
import re
import subprocess
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 4 or os.environ.get("MT_CHILD") is None:
    # one child process per thread count: the probe prints its per-kind lines to stderr at the end of every call
    for n in sys.argv[3:]:
        env = dict(os.environ, MT_CHILD="1", HWANG_PROBE_FAST="1", HWANG_PROBE_TIMES="1")
        r = subprocess.run([sys.executable, __file__, sys.argv[1], sys.argv[2], n], env=env, capture_output=True, text=True)
        agg = {}
        for m in re.finditer(r"probe: (\d+) (\S+) pictures, ([\d.]+) ms .*?; (\d+) instructions, (\d+) cycles, (\d+) branch misses each", r.stderr):
            k = m.group(2)
            a = agg.setdefault(k, [0, 0.0, 0, 0, 0])
            c = int(m.group(1))
            a[0] += c
            a[1] += float(m.group(3)) * c
            a[2] += int(m.group(4)) * c
            a[3] += int(m.group(5)) * c
            a[4] += int(m.group(6)) * c
        for k, a in agg.items():
            print("threads %3s  %-10s %6d pictures: %.3f ms, %.2f M instructions, %.2f M cycles (IPC %.2f), %.0f branch misses each" % (
                n, k, a[0], a[1] / a[0], a[2] / a[0] / 1e6, a[3] / a[0] / 1e6, a[2] / max(1, a[3]), a[4] / a[0]), flush=True)
    sys.exit(0)
import hwang_amd as hw  # noqa: E402
path, libpath, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
idx = hw.index_video(path)
data = np.fromfile(path, dtype=np.uint8)
offs = idx.sample_offsets_np().astype(np.uint64)
sizes = idx.sample_sizes_np().astype(np.uint64)
avcc = bytes(idx.metadata_bytes())
wanted = np.arange(57, idx.frames(), 60, dtype=np.uint64)
lib = C.CDLL(os.path.abspath(libpath))
lib.hwang_h264_probe.restype = C.c_int
lib.hwang_h264_probe.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t,
                                 C.POINTER(C.c_int64), C.c_char_p, C.c_size_t]


def work():
    for _ in range(3):
        out = (C.c_int64 * 8)()
        err = C.create_string_buffer(256)
        lib.hwang_h264_probe(avcc, len(avcc), data.ctypes.data, offs.ctypes.data, sizes.ctypes.data, len(offs), wanted.ctypes.data, len(wanted),
                             out, err, 256)


th = [threading.Thread(target=work) for _ in range(n)]
for t in th:
    t.start()
for t in th:
    t.join()

#!/usr/bin/env python3
"""Development aid (GPU box): latency of ONE planner interval through the reference's per-interval loop
(hwang.Decoder(..., num_devices=1, interval_group_bytes=0)): rows at the END of GOPs of the bench clip, i.e. a keyframe and
its whole chain of P pictures per returned frame.  Prints ms per interval and the chain launches' share.
usage: chain_probe.py [n=12] [pos=59] [frames=1200]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import hwang  # noqa: E402
import hwang_amd  # noqa: E402
from hwang_amd._capi import lib  # noqa: E402

args = dict(a.split("=") for a in sys.argv[1:])
n, pos, frames = int(args.get("n", 9)), int(args.get("pos", 59)), int(args.get("frames", 1200))
path = bench.make_corpus(frames)
idx = hwang_amd.index_video(path)
ridx = hwang.VideoIndex.deserialize(idx.serialize())
dec = hwang.Decoder(path, video_index=ridx, device_type=hwang.DeviceType.GPU, num_devices=int(args.get("threads", 1)), interval_group_bytes=0)
dec.retrieve([pos])
rows = [[bench.GOP * (2 * k + 1) + pos] for k in range(n)]   # every other GOP: never byte-adjacent, one interval each
lib.hwang_prof_reset()
lib.hwang_prof_enable(1)
t0 = time.perf_counter()
for r in rows:
    out = dec.retrieve(r)
t1 = time.perf_counter()
lib.hwang_prof_enable(0)
print("%d intervals, wanted frame at GOP position %d: %.2f ms per interval" % (n, pos, 1000 * (t1 - t0) / n))
for k in range(lib.hwang_prof_num_kernels()):
    o = (C.c_double * 3)()
    lib.hwang_prof_read(k, o)
    if o[0]:
        print("   %-18s launches %5d  avg %8.1f us  total %7.3f ms per interval" % (lib.hwang_prof_kernel_name(k).decode(), o[0], o[1] / o[0], o[1] / 1e3 / n))

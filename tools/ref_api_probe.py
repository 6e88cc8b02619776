#!/usr/bin/env python3
"""Development aid: the drop-in path (hwang.Decoder over pybind11, ONE DecoderAutomata, ONE AMD backend with
num_devices entropy workers) on the bench corpus; prints rate and CPU use per setting.
usage: ref_api_probe.py [frames=108000] [rows=270] [threads=1,4,16] [group_mb=64]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import hwang  # noqa: E402
import hwang_amd  # noqa: E402
import ctypes as C  # noqa: E402
from hwang_amd._capi import lib  # noqa: E402

args = dict(a.split("=") for a in sys.argv[1:])
frames = int(args.get("frames", 108000))
nrows = int(args.get("rows", 270))
threads = [int(x) for x in args.get("threads", "1,4,16").split(",")]
group = int(args.get("group_mb", 64)) << 20
path = bench.make_corpus(frames)
idx = hwang_amd.index_video(path)
ridx = hwang.VideoIndex.deserialize(idx.serialize())
rows = sorted(np.random.default_rng(bench.SEED).choice(idx.frames(), max(1, round(0.01 * idx.frames())), replace=False).tolist())[:nrows]
# what else the bench process has going when it runs this path (value_reference_api): a torch HIP context (torch=1),
# a BatchRetriever that has just served the request (retr=1)
if int(args.get("torch", 0)) or int(args.get("retr", 0)):
    import torch
    torch.cuda.synchronize()
if int(args.get("retr", 0)):
    from hwang_amd import BatchRetriever
    br = BatchRetriever(device_ids=[0], num_lanes=int(args.get("lanes", 48)))
    fobj = open(path, "rb")
    out_t = torch.empty((len(rows), bench.HEIGHT, bench.WIDTH, 3), dtype=torch.uint8, device="cuda:0")
    for _ in range(2):
        br.retrieve_into([(idx, fobj, rows, out_t.data_ptr(), out_t.numel())])
    torch.cuda.synchronize()
    if int(args.get("retr", 0)) == 2:   # ... and is gone again
        del br
for t in threads:
    dec = hwang.Decoder(path, video_index=ridx, device_type=hwang.DeviceType.GPU, num_devices=t, interval_group_bytes=group)
    dec.retrieve(rows[:4])
    tm = {"initialize": 0.0, "get_frames": 0.0}
    inner = dec._decoder

    class Timed(object):   # where retrieve() spends its time: building / copying the intervals vs decoding
        def initialize(self, *a):
            t = time.perf_counter()
            r = inner.initialize(*a)
            tm["initialize"] += time.perf_counter() - t
            return r

        def get_frames(self, *a):
            t = time.perf_counter()
            r = inner.get_frames(*a)
            tm["get_frames"] += time.perf_counter() - t
            return r
    dec._decoder = Timed()
    prof = bool(int(args.get("prof", 0)))
    if prof:
        lib.hwang_prof_reset()
        lib.hwang_prof_enable(1)
    c0, t0 = os.times(), time.perf_counter()
    out = dec.retrieve(rows)
    t1, c1 = time.perf_counter(), os.times()
    if prof:
        lib.hwang_prof_enable(0)
        for k in range(lib.hwang_prof_num_kernels()):
            o = (C.c_double * 3)()
            lib.hwang_prof_read(k, o)
            if o[0]:
                print("   %-18s launches %5d  avg %8.1f us  total %6.3f s" % (lib.hwang_prof_kernel_name(k).decode(), o[0], o[1] / o[0], o[1] / 1e6))
    print("num_devices %2d: %6.1f frames/s  (%d rows in %.2f s, %.2f CPUs busy; initialize %.2f s, get_frames %.2f s, "
          "Python around them %.2f s)" %
          (t, len(out) / (t1 - t0), len(out), t1 - t0, ((c1.user - c0.user) + (c1.system - c0.system)) / (t1 - t0),
           tm["initialize"], tm["get_frames"], t1 - t0 - tm["initialize"] - tm["get_frames"]), flush=True)
    del out, dec

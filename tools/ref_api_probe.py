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

args = dict(a.split("=") for a in sys.argv[1:])
frames = int(args.get("frames", 108000))
nrows = int(args.get("rows", 270))
threads = [int(x) for x in args.get("threads", "1,4,16").split(",")]
group = int(args.get("group_mb", 64)) << 20
path = bench.make_corpus(frames)
idx = hwang_amd.index_video(path)
ridx = hwang.VideoIndex.deserialize(idx.serialize())
rows = sorted(np.random.default_rng(bench.SEED).choice(idx.frames(), max(1, round(0.01 * idx.frames())), replace=False).tolist())[:nrows]
for t in threads:
    dec = hwang.Decoder(path, video_index=ridx, device_type=hwang.DeviceType.GPU, num_devices=t, interval_group_bytes=group)
    dec.retrieve(rows[:4])
    c0, t0 = os.times(), time.perf_counter()
    out = dec.retrieve(rows)
    t1, c1 = time.perf_counter(), os.times()
    print("num_devices %2d: %6.1f frames/s  (%d rows in %.2f s, %.2f CPUs busy)" %
          (t, len(out) / (t1 - t0), len(out), t1 - t0, ((c1.user - c0.user) + (c1.system - c0.system)) / (t1 - t0)), flush=True)
    del out, dec

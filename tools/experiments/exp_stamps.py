#!/usr/bin/env python3
"""Round 4 experiment (GPU box): where one wave's time goes inside k_mv / k_deblock.  Needs a library built with
-DEXP_STAMPS=1|2 -DEXP_STAMP_ROW=r (tools/experiments/build_variant.sh) and HWANG_AMD_LIB pointing at it.  Decodes a
1080p clip with ONE lane (launches of one picture), then prints the s_memtime differences between the stamps of the
last launch's row r, macroblock by macroblock, and their averages.
usage: HWANG_AMD_LIB=_ab/lib_x.so exp_stamps.py"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hwang_amd as hw  # noqa: E402
from hwang_amd._capi import lib  # noqa: E402

path = "/tmp/stamps_1080.mp4"
if not os.path.exists(path):
    subprocess.check_call([os.path.join(ROOT, "corpus/_build/h264_encoder"), "--width", "1920", "--height", "1080", "--frames", "120",
                           "--gop", "60", "--bframes", "2", "--threads", "16", "--qp", "26", "--fps", "30", "--out", path], stdout=subprocess.DEVNULL)
idx = hw.index_video(path)
import torch  # noqa: E402
rows = [int(os.environ.get("ROW", 57))]   # 57: the P picture three before the second keyframe (a long chain of P pictures, the last launch is one of them); 60: a keyframe
dec = hw.Decoder(path, video_index=idx, num_lanes=1)
lib.hwang_prof_reset()
lib.hwang_prof_enable(1)
out = torch.empty((len(rows), 1080, 1920, 3), dtype=torch.uint8, device="cuda")
dec.retrieve_into(rows, out.data_ptr(), out.numel())
torch.cuda.synchronize()
lib.hwang_prof_enable(0)
for k in range(lib.hwang_prof_num_kernels()):
    o = (C.c_double * 3)()
    lib.hwang_prof_read(k, o)
    if o[0]:
        print("%-18s launches %4d avg %8.1f us" % (lib.hwang_prof_kernel_name(k).decode(), int(o[0]), o[1] / o[0]))
buf = (C.c_uint64 * 2048)()
lib.hwang_debug_stamps.argtypes = [C.c_void_p]
assert lib.hwang_debug_stamps(buf) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.int64)
# s_memtime ticks: calibrate against the span of the row
valid = a[:, 0] > 0
n = int(valid.sum())
print("macroblocks stamped:", n)
t = a[valid]
nslots = 7
for i in range(min(n, 120)):
    d = [int(t[i, k + 1] - t[i, k]) if t[i, k + 1] and t[i, k] else -1 for k in range(nslots - 1)]
    nxt = int(t[i + 1, 0] - t[i, 0]) if i + 1 < n else -1
    print("mb %3d flags %04x  seg %s  to-next %d" % (i, int(t[i, 7]), d, nxt))
steps = np.diff(t[:, 0])
print("mean ticks per macroblock step: %.1f; total ticks row: %d" % (steps.mean(), int(t[-1, 0] - t[0, 0])))
for k in range(nslots - 1):
    ok = (t[:, k] > 0) & (t[:, k + 1] > 0)
    if ok.any():
        print("segment %d->%d mean %.1f ticks" % (k, k + 1, (t[ok, k + 1] - t[ok, k]).mean()))

#!/usr/bin/env python3
"""Development aid (GPU box): throughput of Decoder.retrieve on a synthetic
1080p clip for several lane counts, plus per-kernel event timings."""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hwang_amd as hw  # noqa: E402
from hwang_amd._capi import lib  # noqa: E402

W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
FRAMES = int(os.environ.get("FRAMES", 600))
REPEAT = int(os.environ.get("REPEAT", 6))
GOP = int(os.environ.get("GOP", 60))
QP = int(os.environ.get("QP", 26))
FPS = int(os.environ.get("FPS", 30))
path = "/tmp/probe_%dx%d_%d_%d_%d_qp%d.mp4" % (W, H, FRAMES, REPEAT, GOP, QP)
if not os.path.exists(path):
    t = time.time()
    subprocess.check_call([os.path.join(ROOT, "corpus/_build/h264_encoder"), "--width", str(W), "--height", str(H),
                           "--frames", str(FRAMES), "--gop", str(GOP), "--bframes", "2", "--threads", "64", "--qp", str(QP), "--fps", str(FPS),
                           "--repeat", str(REPEAT), "--out", path])
    print("corpus: %.1fs, %.1f MB" % (time.time() - t, os.path.getsize(path) / 1e6))
idx = hw.index_video(path)
n = idx.frames()
rng = np.random.default_rng(20240601)
rows = sorted(rng.choice(n, max(1, round(0.01 * n)), replace=False).tolist())
if os.environ.get("ROWS_MODE") == "keyframes":   # intra-only path in isolation
    kf = idx.keyframe_indices()
    rows = sorted(rng.choice(kf, min(len(kf), len(rows)), replace=False).tolist())
elif os.environ.get("ROWS_MODE") == "stride10":  # BASELINE config C2: every 10th frame, every GOP touched
    rows = list(range(0, n, 10))
elif os.environ.get("ROWS_MODE") == "dense":      # BASELINE config C4: every frame
    rows = list(range(n))
elif os.environ.get("ROWS_MODE") == "gopend":    # longest P chains
    kf = idx.keyframe_indices()
    rows = sorted(set(int(k) - 3 for k in rng.choice(kf[1:], min(len(kf) - 1, len(rows)), replace=False)))
print("frames", n, "rows", len(rows))
import torch
if os.environ.get("HOST_OUT"):   # PCIe-inclusive variant: frames land in pinned host memory
    out = torch.empty((len(rows), H, W, 3), dtype=torch.uint8).pin_memory()
else:
    out = torch.empty((len(rows), H, W, 3), dtype=torch.uint8, device="cuda")
for lanes in [int(x) for x in os.environ.get("LANES", "1,4,16,32,64").split(",")]:
    dec = hw.Decoder(path, video_index=idx, num_lanes=lanes)
    best = None
    cpu_best = 0.0
    smp = None
    if os.environ.get("SAMPLE"):   # process-wide PC sampling of the last iteration -> gpurun_out/samples_<lanes>.txt
        smp = C.CDLL(os.path.join(ROOT, "tools/hostprof/libsampler.so"))
    for it in range(12 if smp else 3):
        if smp and it == 2:
            smp.hw_sampler_start()
        t = time.time()
        c0 = os.times()
        dec.retrieve_into(rows, out.data_ptr(), out.numel())
        dt = time.time() - t
        c1 = os.times()
        if best is None or dt < best:
            best, cpu_best = dt, (c1.user - c0.user) + (c1.system - c0.system)
    if smp:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        smp.hw_sampler_dump(os.path.join(ROOT, "gpurun_out", "samples_%d.txt" % lanes).encode())
    st = dec.stats()
    if os.environ.get("CPU_ACCOUNT"):
        # CPU seconds per thread name since process start (lane threads are re-created per retrieve)
        import collections, glob
        acc = collections.Counter()
        for f in glob.glob("/proc/self/task/*/stat"):
            try:
                txt = open(f).read()
            except OSError:
                continue
            name = txt[txt.index("(") + 1:txt.rindex(")")]
            fields = txt[txt.rindex(")") + 2:].split()
            acc[name] += (int(fields[11]) + int(fields[12])) / os.sysconf("SC_CLK_TCK")
        t = os.times()
        print("   process cpu user %.1f s sys %.1f s; live threads by name: %s" % (t.user, t.system, dict(acc.most_common(8))))
    print("lanes %3d: %.3fs  returned %.1f fps  reconstructed %.0f pics/s  fed %d skipped %d entropy %.2fs stall %.2fs (sum over lanes) process cpu %.2fs upload %.1f MB" % (
        lanes, best, len(rows) / best, st["pictures_reconstructed"] / best, st["samples_fed"], st["pictures_skipped"],
        st["entropy_ns"] / 1e9, st["stall_ns"] / 1e9, cpu_best, st["bytes_uploaded"] / 1e6))
    del dec
if os.environ.get("SERIAL"):
    # the reference-shaped path: ONE DecoderAutomata, intervals one after the other (python/hwang/decoder.py:30-69)
    k = int(os.environ["SERIAL"])
    dec = hw.Decoder(path, video_index=idx, num_lanes=1)
    dec.retrieve_serial(rows[:4])
    t = time.time()
    c0 = os.times()
    got = dec.retrieve_serial(rows[:k])
    dt = time.time() - t
    c1 = os.times()
    print("serial (one DecoderAutomata, reference loop): %d rows in %.2fs = %.1f returned fps, process cpu %.2fs" % (
        len(got), dt, len(got) / dt, (c1.user - c0.user) + (c1.system - c0.system)))
    del dec
# per-kernel timings with one lane (events serialise nothing else)
lib.hwang_prof_enable(1)
lib.hwang_prof_reset()
dec = hw.Decoder(path, video_index=idx, num_lanes=int(os.environ.get("PROF_LANES", "8")))
dec.retrieve_into(rows, out.data_ptr(), out.numel())
lib.hwang_prof_enable(0)
for k in range(lib.hwang_prof_num_kernels()):
    o = (C.c_double * 3)()
    lib.hwang_prof_read(k, o)
    if o[0]:
        print("%-16s launches %6d  avg %8.1f us  alg %.2f GB/s" % (lib.hwang_prof_kernel_name(k).decode(), o[0], o[1] / o[0], o[2] / (o[1] * 1e-6) / 1e9))
del dec  # before interpreter exit: module globals outlive the library's atexit shutdown

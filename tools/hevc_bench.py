"""HEVC leg of the measurements: a 1080p30 HEVC Main clip - `--gops` unique closed GOPs encoded in
parallel by corpus/hevc_encoder (one process each), their samples concatenated and repeated to `--frames` - read with
the configuration the headline bench uses for H.264: a uniform-random 1 % of the frames, one batched call, GOP work
items over the lanes of one GPU, frames returned as RGB24 in host memory.  Prints one JSON object; bench.py embeds it
as `hevc` (python bench.py --hevc) and tools/experiments/r06_*.sh profile it with rocprofv3.

The clip is a blob of samples plus a VideoIndex built in memory (hwang_amd.VideoIndex takes offsets and sizes as they
are): the indexer is not what is measured here."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_clip(workdir, width, height, gops, gop, frames, qp, bframes=2):
    import numpy as np
    import hwang_amd as hw
    enc = os.path.join(ROOT, "corpus", "_build", "hevc_encoder")
    os.makedirs(workdir, exist_ok=True)
    blob_path = os.path.join(workdir, "hevc_%dx%d_%d_%d_%d_qp%d.blob" % (width, height, gops, gop, frames, qp))
    meta_path = blob_path + ".json"
    if not (os.path.exists(blob_path) and os.path.exists(meta_path)):
        procs = []
        for g in range(gops):
            path = os.path.join(workdir, "gop%d.mp4" % g)
            procs.append((path, subprocess.Popen([enc, "--width", str(width), "--height", str(height), "--frames", str(gop), "--gop", str(gop),
                                                  "--bframes", str(bframes), "--refs", "2", "--ctb", "64", "--sao", "1", "--sdh", "1", "--amp", "1",
                                                  "--qp", str(qp), "--seed", str(100 + g), "--out", path], stderr=subprocess.DEVNULL)))
        chunks, extradata = [], None
        for path, p in procs:
            if p.wait():
                raise RuntimeError("hevc_encoder failed")
            idx = hw.index_video(path)
            data = open(path, "rb").read()
            chunks.append([data[o:o + s] for o, s in zip(idx.sample_offsets(), idx.sample_sizes())])
            extradata = bytes(idx.metadata_bytes())
            os.unlink(path)
        sizes, keys = [], []
        with open(blob_path, "wb") as f:
            n = 0
            while n < frames:
                for ch in chunks:
                    if n >= frames:
                        break
                    keys.append(n)
                    for smp in ch:
                        f.write(smp)
                        sizes.append(len(smp))
                    n += len(ch)
        json.dump({"sizes": sizes, "keys": keys, "extradata": extradata.hex()}, open(meta_path, "w"))
    meta = json.load(open(meta_path))
    sizes = meta["sizes"]
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).tolist()
    idx = hw.VideoIndex(timescale=30000, duration=len(sizes) * 1000, width=width, height=height, format="hev1", sample_offsets=offs,
                        sample_sizes=sizes, keyframe_indices=meta["keys"], metadata=bytes.fromhex(meta["extradata"]))
    return blob_path, idx


def run(width=1920, height=1080, gops=8, gop=24, frames=7200, qp=30, lanes=16, steps=3, fraction=0.01, workdir="/tmp/hwang_hevc_bench"):
    import ctypes as C
    import numpy as np
    import hwang_amd as hw
    import hwang_amd._capi
    t0 = time.perf_counter()
    blob, idx = build_clip(workdir, width, height, gops, gop, frames, qp)
    t_build = time.perf_counter() - t0
    n = idx.frames()
    rows = sorted(np.random.default_rng(20240601).choice(n, max(1, round(fraction * n)), replace=False).tolist())
    dec = hw.Decoder(blob, video_index=idx, num_lanes=lanes)
    dec.retrieve(rows[:8])     # first-use costs outside
    times = []
    for _ in range(steps):
        t = time.perf_counter()
        out = dec.retrieve(rows)
        times.append(time.perf_counter() - t)
        assert len(out) == len(rows) and out[0].shape == (height, width, 3)
        st = dec.stats()
        hc = (C.c_int64 * 4)()
        hw._capi.lib.hwang_retriever_host_counters(dec._get_retriever(), hc)
        del out
    best = min(times)
    if os.environ.get("HEVC_BENCH_THREADS"):   # where the process's CPU time went, thread by thread (lane threads end with their call)
        rows_ = []
        for tid in os.listdir("/proc/self/task"):
            try:
                comm = open("/proc/self/task/%s/comm" % tid).read().strip()
                f = open("/proc/self/task/%s/stat" % tid).read().rsplit(")", 1)[1].split()
                rows_.append((int(f[11]) + int(f[12]), comm, tid))
            except OSError:
                pass
        tick = os.sysconf("SC_CLK_TCK")
        for t, comm, tid in sorted(rows_, reverse=True)[:12]:
            print("thread %-16s %7.2f s" % (comm, t / tick), file=sys.stderr)
        ru = os.times()
        print("process user %.2f s system %.2f s; calls took %s" % (ru.user, ru.system, times), file=sys.stderr)
    return {"what": "HEVC Main %dx%d, %d frames (%d unique GOPs of %d, CTB 64, 2 B pictures, SAO, AMP, sign hiding, QP %d), %.0f %% uniform-random "
                    "sample, one batched call, %d lanes of one GPU, RGB24 into host memory" % (width, height, n, gops, gop, qp, 100 * fraction, lanes),
            "rows": len(rows), "value": round(len(rows) / best, 2), "unit": "frames/s", "seconds_per_call": [round(t, 3) for t in times],
            "pictures_reconstructed": st["pictures_reconstructed"], "pictures_skipped": st["pictures_skipped"],
            "pictures_reconstructed_per_s": round(st["pictures_reconstructed"] / best, 1),
            "host_parse_ms_per_picture": round(st["entropy_ns"] / 1e6 / max(1, st["pictures_reconstructed"]), 3),
            "lane_cpu_ms_per_picture": round(hc[3] / 1e6 / max(1, st["pictures_reconstructed"]), 3),
            "lane_wall_ms_per_picture": round(lanes * times[-1] * 1e3 / max(1, st["pictures_reconstructed"]), 3),
            "staging_stall_ms_per_picture": round(st["stall_ns"] / 1e6 / max(1, st["pictures_reconstructed"]), 3),
            "upload_MB_per_picture": round(st["bytes_uploaded"] / 1e6 / max(1, st["pictures_reconstructed"]), 3),
            "bytes_per_sample": round(sum(idx.sample_sizes()) / n), "clip_build_s": round(t_build, 1),
            "parity": "unpinned: no HEVC decoder or foreign stream exists on either machine (DESIGN.md section 5a)"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=7200)
    ap.add_argument("--gops", type=int, default=8)
    ap.add_argument("--gop", type=int, default=24)
    ap.add_argument("--lanes", type=int, default=16)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--fraction", type=float, default=0.01)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    a = ap.parse_args()
    print(json.dumps(run(width=a.width, height=a.height, gops=a.gops, gop=a.gop, frames=a.frames, lanes=a.lanes, steps=a.steps, fraction=a.fraction)))

#!/usr/bin/env python3
"""BASELINE.md "CPU baseline procedure" step 1: look for an FFmpeg on this
machine (shared libavcodec / libswscale / libavutil, or an `ffmpeg` binary).

The reference's pixel path is libavcodec + libswscale
(/root/reference/hwang/impls/software/software_video_decoder.cpp:349-457), so a
loadable FFmpeg is the only thing that can pin pixel parity and the only
honest "FFmpeg CPU" baseline.  This probe never installs or downloads
anything; it reports what it finds as one JSON object (stdout, and
gpurun_out/ffmpeg_probe.json when run from the repo root).

Used by bench.py (`cpu_baseline.ffmpeg_probe`) and by
tests/golden/make_ffmpeg_golden.py (only if something is found)."""
import ctypes
import ctypes.util
import glob
import json
import os
import shutil
import subprocess
import sys

LIBS = {
    "avcodec": range(56, 64),
    "swscale": range(3, 10),
    "avutil": range(54, 62),
    "avformat": range(56, 64),
}
SEARCH_DIRS = [
    "/usr/lib", "/usr/lib64", "/usr/lib/x86_64-linux-gnu", "/usr/local/lib", "/usr/local/lib64",
    "/opt/conda/lib", "/opt/rocm/lib", "/opt/rocm/lib64", "/opt/ffmpeg/lib", "/opt/lib",
]


def _try_load(name):
    try:
        ctypes.CDLL(name)
        return True
    except OSError:
        return False


def probe(deep=True):
    found = {"binary": None, "binary_version": None, "libs": {}, "files": [], "python_modules": []}
    exe = shutil.which("ffmpeg") or shutil.which("avconv") or shutil.which("ffprobe")
    if exe:
        found["binary"] = exe
        try:
            out = subprocess.run([exe, "-version"], capture_output=True, text=True, timeout=20).stdout
            found["binary_version"] = out.splitlines()[0] if out else None
        except Exception as e:  # noqa: BLE001
            found["binary_version"] = "error: %r" % (e,)
    for lib, versions in LIBS.items():
        hit = None
        for v in versions:
            so = "lib%s.so.%d" % (lib, v)
            if _try_load(so):
                hit = so
                break
        if hit is None:
            n = ctypes.util.find_library(lib)
            if n and _try_load(n):
                hit = n
        found["libs"][lib] = hit
    if deep:
        pats = []
        for d in SEARCH_DIRS:
            pats += glob.glob(os.path.join(d, "libavcodec*")) + glob.glob(os.path.join(d, "libswscale*"))
            pats += glob.glob(os.path.join(d, "**", "libavcodec*.so*"), recursive=True)[:8]
        # python wheels that bundle an FFmpeg (PyAV, imageio-ffmpeg, opencv, torchvision's video ops)
        for sp in set(p for p in sys.path if p and os.path.isdir(p)):
            pats += glob.glob(os.path.join(sp, "*", "libavcodec*"))
            pats += glob.glob(os.path.join(sp, "*.libs", "libavcodec*"))
            pats += glob.glob(os.path.join(sp, "imageio_ffmpeg", "binaries", "*"))
        found["files"] = sorted(set(pats))[:32]
        for mod in ("av", "cv2", "imageio_ffmpeg", "torchcodec", "torchvision.io", "decord"):
            try:
                __import__(mod)
                found["python_modules"].append(mod)
            except Exception:  # noqa: BLE001
                pass
    found["available"] = bool(found["binary"] or found["libs"].get("avcodec") or found["files"])
    found["summary"] = (
        "FFmpeg found: %s" % (found["binary"] or found["libs"].get("avcodec") or found["files"][0])
        if found["available"] else
        "FFmpeg oracle unavailable on this machine: no ffmpeg/avconv binary in PATH, "
        "no loadable libavcodec.so.{56..63}/libswscale, no bundled copies in site-packages")
    return found


if __name__ == "__main__":
    r = probe()
    r["host"] = {"nproc_visible": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
    try:
        r["host"]["cpu_max"] = open("/sys/fs/cgroup/cpu.max").read().strip()
    except OSError:
        pass
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                r["host"]["cpu_model"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    txt = json.dumps(r, indent=1)
    print(txt)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "ffmpeg_probe.json"), "w") as f:
        f.write(txt + "\n")

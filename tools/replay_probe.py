#!/usr/bin/env python3
"""GPU box: the reconstruction kernels with the host out of the way.  Decodes one C1 request with the
retriever (48 lanes) while the executor keeps the descriptors of the last pictures, then replays those pictures - the
records and surfaces are still in HBM - in launches of a given width, back to back.  Prints JSON: pictures/s and the
per-kernel spans (average launch time, algorithmic GB/s, fraction of the 8 TB/s HBM roofline) per width, for the wide
kernels and for chain launches.
usage: replay_probe.py [frames=10800] [widths=8,24,48,96,160] [launches=40] [out=path.json]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import hwang_amd as hw  # noqa: E402
from hwang_amd._capi import lib  # noqa: E402
from hwang_amd.decoder import BatchRetriever  # noqa: E402

args = dict(a.split("=") for a in sys.argv[1:])
frames = int(args.get("frames", 10800))
widths = [int(x) for x in args.get("widths", "8,24,48,96,160").split(",")]
launches = int(args.get("launches", 40))
lanes = int(args.get("lanes", 48))
path = bench.make_corpus(frames)
idx = hw.index_video(path)
rows = bench.request_rows(np, idx.frames(), 0)
import torch  # noqa: E402
out = torch.empty((len(rows), bench.HEIGHT, bench.WIDTH, 3), dtype=torch.uint8, device="cuda:0")
br = BatchRetriever(device_ids=[0], num_lanes=lanes)
fobj = open(path, "rb")
lib.hwang_replay_keep(0, 4 * lanes + 64)
br.retrieve_into([(idx, fobj, rows, out.data_ptr(), out.numel())])
torch.cuda.synchronize()


def spans():
    d = {}
    for k in range(lib.hwang_prof_num_kernels()):
        o = (C.c_double * 3)()
        lib.hwang_prof_read(k, o)
        if o[0]:
            d[lib.hwang_prof_kernel_name(k).decode()] = {"launches": int(o[0]), "avg_us": round(o[1] / o[0], 1),
                                                          "alg_GBps": round(o[2] / (o[1] * 1e-6) / 1e9, 1),
                                                          "hbm_frac": round(o[2] / (o[1] * 1e-6) / 1e9 / bench.HBM_PEAK_GBS, 4)}
    return d


result = {"workload": "%s, %d lanes; the last pictures of a C1 request replayed from HBM" % (os.path.basename(path), lanes), "runs": []}
for chain in (0, 1):
    for w in widths:
        o = (C.c_double * 4)()
        err = C.create_string_buffer(256)
        lib.hwang_replay_run(0, w, 3, chain, o, err, 256)   # warm
        lib.hwang_prof_reset()
        lib.hwang_prof_enable(1)
        rc = lib.hwang_replay_run(0, w, launches, chain, o, err, 256)
        lib.hwang_prof_enable(0)
        if rc:
            result["runs"].append({"kernels": "chain" if chain else "wide", "width": w, "error": err.value.decode()})
            continue
        k = spans()
        main = ("chain",) if chain else ("inter_pred", "intra_pred", "deblock")
        main_us = sum(k[n]["avg_us"] for n in main if n in k)
        result["runs"].append({"kernels": "chain launch" if chain else "wide kernels", "pictures_available": int(o[0]), "width": int(o[1]),
                               "launches": int(o[2]), "wall_s": round(o[3], 4),
                               "pictures_per_s_wall": round(o[1] * o[2] / o[3], 1),
                               "pictures_per_s_sample_stage": round(o[1] / (main_us * 1e-6), 1) if main_us else None,
                               "spans": k})
        if int(o[1]) < w:
            break
print(json.dumps(result, indent=1))
if "out" in args:
    with open(args["out"], "w") as f:
        json.dump(result, f, indent=1)

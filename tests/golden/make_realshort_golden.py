#!/usr/bin/env python3
"""Generates tests/golden/realshort_md5.json: per-frame MD5 of the I420
output of the CPU ORACLE on imageio's realshort.mp4 (x264, High, CABAC,
8x8 transform, I+P, 320x240, 36 frames).  The clip itself is not committed
(licence unknown); only these derived digests are.  They are a regression
pin of the oracle, not an independent golden: no FFmpeg exists in this image
to produce one ("parity unpinned")."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hwang_amd  # noqa: E402
from oracle import oracle_py  # noqa: E402
from tests.util import REALSHORT  # noqa: E402

idx = hwang_amd.index_video(REALSHORT)
frames, counters = oracle_py.decode_file(REALSHORT, idx)
out = {"counters": counters, "pocs": [f[3] for f in frames],
       "md5": [hashlib.md5(f[0].tobytes()).hexdigest() for f in frames]}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "realshort_md5.json"), "w"), indent=1)
print(counters)

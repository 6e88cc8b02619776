#!/usr/bin/env python3
"""Generates tests/golden/corpus_md5.json and tests/golden/corpus_coverage.json.

corpus_md5.json: for every corpus case the parity tests use (tests/corpus_util.py: MATRIX and BAND_SIZES) the MD5
of the encoded .mp4 and, per frame in display order, the MD5 of the Y, U and V plane of the CPU ORACLE's output.
The GPU parity tests compare the HIP path with these committed digests (and, live, with the encoder's
reconstruction), the CPU tests compare the oracle with them: a change of the oracle, of the corpus encoder or of the
HIP path between rounds shows up as a digest mismatch instead of moving all three together.

They are a regression pin, not an independent golden: no FFmpeg exists in this image to produce one (DESIGN.md,
"parity unpinned").

corpus_coverage.json: the oracle's coverage histogram (oracle/h264_oracle.h, H264O_COV_*) summed over all cases:
per section the list of cells the corpus does not reach.  tests/test_corpus_roundtrip.py recomputes it and holds the
list of cells that are unreachable by construction.

Run from the repo root:  python tests/golden/make_corpus_golden.py"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import hwang_amd  # noqa: E402
from oracle import oracle_py  # noqa: E402
from tests.corpus_util import all_cases, make_clip, plane_md5s  # noqa: E402


def main():
    cases = {}
    cov = np.zeros(oracle_py.COV_CELLS, dtype=np.int64)
    with tempfile.TemporaryDirectory() as tmp:
        for name, w, h, kw in all_cases():
            path = os.path.join(tmp, name + ".mp4")
            gold = make_clip(path, width=w, height=h, **kw)
            idx = hwang_amd.index_video(path)
            frames, counters = oracle_py.decode_file(path, idx, coverage=cov)
            assert counters["slices"] == counters["slices_clean_end"] > 0, name
            assert len(frames) == len(gold), name
            for i, f in enumerate(frames):
                assert np.array_equal(f[0], gold[i]), (name, i)
            cases[name] = {"width": w, "height": h, "mp4_md5": hashlib.md5(open(path, "rb").read()).hexdigest(),
                           "frames": [plane_md5s(f[0], w, h) for f in frames]}
            os.unlink(path)
            print("%-32s %3d frames" % (name, len(frames)), flush=True)
    gdir = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(gdir, "corpus_md5.json"), "w") as f:
        json.dump({"note": "per frame: MD5 of the Y, U, V plane of the CPU oracle's output; see make_corpus_golden.py",
                   "cases": cases}, f, indent=0, sort_keys=True)
    holes = {}
    for sec, (base, n) in oracle_py.COV_SECTIONS.items():
        holes[sec] = {"cells": n, "reached": int((cov[base:base + n] > 0).sum()),
                      "unreached": [int(i) for i in np.nonzero(cov[base:base + n] == 0)[0]]}
    with open(os.path.join(gdir, "corpus_coverage.json"), "w") as f:
        json.dump({"note": "cells of the oracle's coverage histogram (oracle/h264_oracle.h) that no corpus case reaches",
                   "sections": holes}, f, indent=0, sort_keys=True)
    for sec, hinfo in holes.items():
        print("%-16s %5d / %5d reached" % (sec, hinfo["reached"], hinfo["cells"]))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Development aid (GPU box): decode a clip through the product backend and
through the oracle, report the first differing macroblock per frame."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hwang_amd as hw  # noqa: E402
from oracle import oracle_py  # noqa: E402


def compare(path, rows=None, verbose=True):
    idx = hw.index_video(path)
    w, h = idx.frame_width(), idx.frame_height()
    n = idx.frames()
    rows = list(range(n)) if rows is None else rows
    dec = hw.Decoder(path, video_index=idx, num_lanes=1)
    got = dec.retrieve_yuv(rows)
    ref, _ = oracle_py.decode_file(path, idx)
    bad = 0
    for k, r in enumerate(rows):
        a, b = got[k], ref[r][0]
        if np.array_equal(a, b):
            continue
        bad += 1
        if verbose and bad <= 6:
            ya, yb = a[:w * h].reshape(h, w), b[:w * h].reshape(h, w)
            d = np.argwhere(ya != yb)
            cw, ch = (w + 1) // 2, (h + 1) // 2
            ua, ub = a[w * h:w * h + cw * ch].reshape(ch, cw), b[w * h:w * h + cw * ch].reshape(ch, cw)
            va, vb = a[w * h + cw * ch:].reshape(ch, cw), b[w * h + cw * ch:].reshape(ch, cw)
            du, dv = np.argwhere(ua != ub), np.argwhere(va != vb)
            msg = "frame %d: luma diffs %d" % (r, len(d))
            if len(d):
                y0, x0 = d[0]
                msg += " first at (x=%d,y=%d) mb=(%d,%d) got %d want %d" % (x0, y0, x0 // 16, y0 // 16, ya[y0, x0], yb[y0, x0])
                mbs = sorted({(int(y) // 16, int(x) // 16) for y, x in d})
                msg += " | %d MBs differ, first few %s" % (len(mbs), mbs[:5])
            msg += " | U diffs %d V diffs %d" % (len(du), len(dv))
            if len(du):
                msg += " firstU (x=%d,y=%d)" % (du[0][1], du[0][0])
            print(msg)
    print("%s: %d/%d frames match" % (os.path.basename(path), len(rows) - bad, len(rows)))
    return bad == 0


if __name__ == "__main__":
    ok = True
    for p in sys.argv[1:]:
        ok &= compare(p)
    sys.exit(0 if ok else 1)

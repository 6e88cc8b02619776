#!/usr/bin/env python3
"""Development aid (GPU box): a corpus clip through the backend with and without chain launches (HWANG_CHAIN is read when
a lane registers, so each decode runs in a process of its own) against the encoder's reconstruction; prints which
macroblocks of the first wrong frames differ.
usage: chain_debug.py [width=320] [height=240] [lanes=2] [key=value for the encoder ...]"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(path, lanes):
    import ctypes
    import threading
    import hwang_amd as hw
    from hwang_amd._capi import lib

    def watchdog():   # a launch that never ends: say what its waves were doing, then leave
        import time
        time.sleep(float(os.environ.get("CHAIN_DEBUG_WATCHDOG", 15)))
        buf = (ctypes.c_uint32 * 2048)()
        fn = lib.hwang_debug_chain_state
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int]
        rc = fn(buf, 2048)
        names = {0: "-", 1: "need inter (row, mb)", 2: "need intra (row, count)", 3: "need reference (row, mbs)", 4: "inter unit (job, run)",
                 5: "intra unit (job, band)", 6: "deblock unit (job, band)", 7: "unit done", 8: "inter waves done", 9: "row waves done"}
        print("WATCHDOG rc=%d" % rc, flush=True)
        from collections import Counter
        c = Counter()
        for wg in range(256):
            for w in range(8):
                v = buf[wg * 8 + w]
                if v:
                    c[(names.get(v >> 24, str(v >> 24)), "rows" if w < 4 else "inter")] += 1
        for k, n in sorted(c.items()):
            print("   %4d waves: %s %s" % (n, k[1], k[0]), flush=True)
        shown = 0
        for wg in range(256):
            for w in range(8):
                v = buf[wg * 8 + w]
                if v and (v >> 24) in (1, 2, 3, 5, 6) and shown < 60:
                    print("   wg %3d wave %d: %s a=%d b=%d" % (wg, w, names[v >> 24], (v >> 12) & 0xfff, v & 0xfff), flush=True)
                    shown += 1
        os._exit(3)
    threading.Thread(target=watchdog, daemon=True).start()
    idx = hw.index_video(path)
    dec = hw.Decoder(path, video_index=idx, num_lanes=lanes)
    got = dec.retrieve_yuv(list(range(idx.frames())))
    np.stack(got).tofile(path + ".out.%s" % os.environ.get("HWANG_CHAIN", "x"))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        return child(sys.argv[2], int(sys.argv[3]))
    from tests.corpus_util import make_clip
    kw = dict(a.split("=") for a in sys.argv[1:])
    w, h, lanes = int(kw.pop("width", 320)), int(kw.pop("height", 240)), int(kw.pop("lanes", 2))
    kw = {k: int(v) for k, v in kw.items()} or dict(frames=10, gop=10, bframes=0, cabac=1, t8x8=1, refs=2)
    path = "/tmp/chain_debug.mp4"
    gold = make_clip(path, width=w, height=h, **kw)
    fs = w * h * 3 // 2
    for mode in ("0", "1"):
        env = dict(os.environ, HWANG_CHAIN=mode)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", path, str(lanes)], env=env, capture_output=True, text=True)
        if r.returncode:
            print("HWANG_CHAIN=%s: decode failed: %s" % (mode, (r.stdout + r.stderr)[-6000:]))
            continue
        got = np.fromfile(path + ".out." + mode, dtype=np.uint8).reshape(-1, fs)
        bad = [i for i in range(len(gold)) if not np.array_equal(got[i], gold[i])]
        print("HWANG_CHAIN=%s %s: %d frames, wrong: %s" % (mode, json.dumps(kw), len(gold), bad))
        err = [l for l in r.stderr.splitlines() if "chain launches" in l or "fault" in l.lower()]
        for l in err[:6]:
            print("   ", l)
        for i in bad[:3]:
            ya, yb = got[i][:w * h].reshape(h, w), gold[i][:w * h].reshape(h, w)
            cw, ch = w // 2, h // 2
            ua, ub = got[i][w * h:w * h + cw * ch].reshape(ch, cw), gold[i][w * h:w * h + cw * ch].reshape(ch, cw)
            va, vb = got[i][w * h + cw * ch:].reshape(ch, cw), gold[i][w * h + cw * ch:].reshape(ch, cw)
            mbh, mbw = (h + 15) // 16, (w + 15) // 16
            print("  frame %d: luma samples wrong %d, Cb %d, Cr %d; macroblock map (L luma, c chroma, * both):" %
                  (i, int((ya != yb).sum()), int((ua != ub).sum()), int((va != vb).sum())))
            for my in range(mbh):
                row = ""
                for mx in range(mbw):
                    l = (ya[my * 16:my * 16 + 16, mx * 16:mx * 16 + 16] != yb[my * 16:my * 16 + 16, mx * 16:mx * 16 + 16]).any()
                    c = (ua[my * 8:my * 8 + 8, mx * 8:mx * 8 + 8] != ub[my * 8:my * 8 + 8, mx * 8:mx * 8 + 8]).any() or \
                        (va[my * 8:my * 8 + 8, mx * 8:mx * 8 + 8] != vb[my * 8:my * 8 + 8, mx * 8:mx * 8 + 8]).any()
                    row += "*" if l and c else "L" if l else "c" if c else "."
                print("    " + row)
            d = np.argwhere(ya != yb)
            if len(d):
                y0, x0 = d[0]
                print("    first luma difference at x=%d y=%d (in its macroblock: %d, %d): got %d want %d" %
                      (x0, y0, x0 % 16, y0 % 16, ya[y0, x0], yb[y0, x0]))
                ys = sorted({int(y) % 16 for y, x in d})
                xs = sorted({int(x) % 16 for y, x in d})
                print("    rows inside macroblocks that differ: %s; columns: %s" % (ys, xs))


if __name__ == "__main__":
    main()

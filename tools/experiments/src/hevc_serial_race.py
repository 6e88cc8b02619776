"""Development aid: the automata path (retrieve_serial) against the retriever path, many times over - a frame that
differs now and then is a race between the feeder thread and the fetching thread."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.corpus_util import HEVC_MATRIX, make_hevc_clip
import hwang_amd as hw
from oracle import oracle_py
for name, rowsets in (("open_gop_bb", ([24, 25, 26, 31], [3, 11, 12, 30])), ("radl_bb_tmvp", ([3, 11, 12, 30], [13, 14, 25])), ("p_refs2_long", ([3, 11, 12, 30],))):
    kw = dict(dict(HEVC_MATRIX)[name]) if name != "p_refs2_long" else dict(frames=36, gop=12, bframes=2, refs=2)
    path = "/tmp/race_%s.mp4" % name
    gold = make_hevc_clip(path, **kw)
    idx = hw.index_video(path)
    n = idx.frames()
    gold_rgb = [oracle_py.i420_to_rgb(gold[j], 128, 96) for j in range(n)]
    if rowsets is None:
        rowsets = ([0, 1, n - 2, n - 1], [n // 2, n - 1], list(range(0, n, 3)))
    dec = hw.Decoder(path, video_index=idx, num_lanes=2)
    for rows in rowsets:
        ref = [oracle_py.i420_to_rgb(f, 128, 96) for f in dec.retrieve_yuv(rows)]
        bad = {}
        for it in range(200):
            got = dec.retrieve_serial(rows)
            for k, r in enumerate(rows):
                if not np.array_equal(got[k], ref[k]):
                    d = np.nonzero((got[k] != ref[k]).any(axis=2))
                    same_as = [j for j in range(n) if np.array_equal(got[k], gold_rgb[j])]
                    frac = float((got[k] != ref[k]).any(axis=2).mean())
                    bad.setdefault(r, []).append((it, same_as, round(frac, 3), int(d[0].min()), int(d[0].max())))
        print(name, rows, "mismatching rows:", {r: (len(v), v[:3]) for r, v in bad.items()})

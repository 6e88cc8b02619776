"""Motion-vector prediction (8.4.1.3) and the P_Skip inference (8.4.1.1) checked against a restatement that shares nothing
with the oracle or the kernels.

`--global-mv 7` (corpus/h264_encoder.c) writes P pictures in which every macroblock is one 16x16 partition, or two 16x8 or
8x16 ones, or four quadrants with sub-macroblock partitions down to 4x4, whose reference indices and vector DIFFERENCES follow from its address and the picture's number; every fourth macroblock is P_Skip, some
have reference 0 and vector 0 outright (the neighbour the P_Skip rule looks for).  Nothing
else is coded and the in-loop filter is off.  The vectors themselves are then whatever the median prediction makes of the
differences - in the encoder (which uses the oracle's function), in the oracle, and in the HIP path's k_mv, which gets
the differences and derives the vectors on the GPU.  This test derives them a third time, in Python, from the clause:
neighbours A (left), B (above), C (above-right, or D above-left where C does not exist), the rule that a single
neighbour with the same reference index wins, the median otherwise, "only A available", the neighbour a 16x8 / 8x16
partition looks at first (and that its second partition sees the first), and the P_Skip conditions - and
builds the picture from the two reference pictures with the interpolation of tests/test_inter_independent.py.  Compared
with the oracle's decode (CPU) and the HIP path's (GPU)."""
import numpy as np
import pytest

from tests.corpus_util import make_clip
from tests.test_inter_independent import (B_FRAMES, global_mv_of, implicit_weights, planes, shift_chroma, shift_luma, shifted,
                                          weighted_bi)

W, H = 112, 80   # 7 x 5 macroblocks
FRAMES = 14


def median(a, b, c):
    return a + b + c - min(a, b, c) - max(a, b, c)


class Motion(object):
    """What the macroblocks decoded so far ended up with, per 4x4 block of the picture: reference index (-1: not decoded
    yet) and vector."""
    def __init__(self, mbw, mbh, slice_mbs=0):
        self.w4, self.h4 = mbw * 4, mbh * 4
        self.mbw, self.slice_mbs, self.cur = mbw, slice_mbs, 0   # cur: address of the macroblock being decoded
        self.ref = [[-1] * self.w4 for _ in range(self.h4)]
        self.mv = [[(0, 0)] * self.w4 for _ in range(self.h4)]
        self.decoded = [[False] * self.w4 for _ in range(self.h4)]

    def at(self, x, y):
        """None: no such block (outside the picture, or not decoded yet); else (reference index, vector) - the index is
        -1 and the vector zero where the block does not use this list."""
        if x < 0 or y < 0 or x >= self.w4 or y >= self.h4 or not self.decoded[y][x]:
            return None
        if self.slice_mbs and ((y // 4) * self.mbw + x // 4) // self.slice_mbs != self.cur // self.slice_mbs:
            return None   # (a macroblock of another slice is not available: 6.4.4)
        return self.ref[y][x], self.mv[y][x]

    def put(self, x0, y0, w, h, ref, mv):
        for y in range(y0, y0 + h):
            for x in range(x0, x0 + w):
                self.ref[y][x], self.mv[y][x], self.decoded[y][x] = ref, (mv if ref >= 0 else (0, 0)), True


def predict(m, x, y, bw, ref, direction=None):
    """8.4.1.3 for the partition whose top-left 4x4 block is (x, y) (picture coordinates, 4x4 units) and which is bw
    blocks wide.  direction: None, or the neighbour a 16x8 / 8x16 partition looks at first ("A", "B", "C")."""
    a, b = m.at(x - 1, y), m.at(x, y - 1)
    c = m.at(x + bw, y - 1)
    if c is None:
        c = m.at(x - 1, y - 1)   # (8.4.1.3.2: C is replaced by D when C is not available)
    first = {"A": a, "B": b, "C": c}.get(direction)
    if first is not None and first[0] == ref:
        return first[1]          # (8.4.1.3: the directional rules of 16x8 and 8x16 partitions)
    if b is None and c is None and a is not None:
        return a[1]              # only A is available
    cand = [(n[0], n[1]) if n is not None else (-1, (0, 0)) for n in (a, b, c)]
    same = [n for n in cand if n[0] == ref]
    if len(same) == 1:
        return same[0][1]
    return (median(cand[0][1][0], cand[1][1][0], cand[2][1][0]), median(cand[0][1][1], cand[1][1][1], cand[2][1][1]))


def skip_vector(m, x, y):
    """8.4.1.1: zero when A or B does not exist, or is (reference 0, vector 0); the prediction for reference 0 otherwise."""
    a, b = m.at(x - 1, y), m.at(x, y - 1)
    if a is None or b is None or a == (0, (0, 0)) or b == (0, (0, 0)):
        return (0, 0)
    return predict(m, x, y, 4, 0)


def expected(dec, k, w, h, slice_mbs=0):
    """Picture k from the decoded pictures before it: list 0 of a P picture = the previous pictures, nearest first.
    slice_mbs: a new slice every so many macroblocks (--slice-mbs).
    Returns the picture, the Motion it implies and, per macroblock, what kind it was."""
    mbw, mbh = w // 16, h // 16
    nref = min(2, k)
    m = Motion(mbw, mbh, slice_mbs)
    kinds = []
    out = [np.zeros((h, w), np.uint8), np.zeros((h // 2, w // 2), np.uint8), np.zeros((h // 2, w // 2), np.uint8)]
    cache = {}

    def paint(x4, y4, w4, h4, r, mv):   # a partition, in 4x4 units of the picture
        m.put(x4, y4, w4, h4, r, mv)
        if (r, mv) not in cache:
            y, u, v = planes(dec[k - 1 - r], w, h)
            cache[(r, mv)] = (shift_luma(y, *mv), shift_chroma(u, *mv), shift_chroma(v, *mv))
        for pl, s in ((0, 4), (1, 2), (2, 2)):
            out[pl][y4 * s:(y4 + h4) * s, x4 * s:(x4 + w4) * s] = cache[(r, mv)][pl][y4 * s:(y4 + h4) * s, x4 * s:(x4 + w4) * s]
    for addr in range(mbw * mbh):
        mx, my = addr % mbw, addr // mbw
        x, y = mx * 4, my * 4
        m.cur = addr
        shape = (addr + k) % 3
        if addr % 4 == 3:
            kinds.append("skip")
            paint(x, y, 4, 4, 0, skip_vector(m, x, y))
        elif addr % 7 == 5:
            kinds.append("zero")
            paint(x, y, 4, 4, 0, (0, 0))   # (the encoder codes whatever difference that takes)
        elif (addr * 3 + k) % 5 == 4:
            kinds.append("8x8")   # P_8x8: a reference per quadrant, sub-macroblock partitions in decoding order
            for q in range(4):
                st, r = (addr + q * 3 + k) % 4, (addr + q + k) % nref
                pw, ph = (2 if st in (0, 1) else 1), (2 if st in (0, 2) else 1)   # 0 8x8, 1 8x4, 2 4x8, 3 4x4
                n = 0
                for sy in range(0, 2, ph):
                    for sx in range(0, 2, pw):
                        bx, by = x + (q & 1) * 2 + sx, y + (q >> 1) * 2 + sy
                        px, py = predict(m, bx, by, pw, r)
                        paint(bx, by, pw, ph, r, (px + (addr * 5 + k * 3 + q * 4 + n) % 13 - 6, py + (addr * 7 + k + q * 2 + n * 3) % 9 - 4))
                        n += 1
        elif shape:
            kinds.append("16x8" if shape == 1 else "8x16")
            for part in range(2):
                r = (addr * 3 + k + part) % nref
                dx, dy = (addr * 5 + k * 3 + part * 4) % 13 - 6, (addr * 7 + k + part * 2) % 9 - 4
                if shape == 1:
                    px, py = predict(m, x, y + 2 * part, 4, r, "B" if part == 0 else "A")
                    paint(x, y + 2 * part, 4, 2, r, (px + dx, py + dy))
                else:
                    px, py = predict(m, x + 2 * part, y, 2, r, "A" if part == 0 else "C")
                    paint(x + 2 * part, y, 2, 4, r, (px + dx, py + dy))
        else:
            kinds.append("16x16")
            r = (addr * 3 + k) % nref
            px, py = predict(m, x, y, 4, r)
            paint(x, y, 4, 4, r, (px + (addr * 5 + k * 3) % 13 - 6, py + (addr * 7 + k) % 9 - 4))
    return np.concatenate([p.ravel() for p in out]), m, kinds


def clip_args(slice_mbs=0):
    return dict(frames=FRAMES, gop=FRAMES, bframes=0, refs=2, global_mv=7, deblock=1, slice_mbs=slice_mbs)


SLICE_MBS = [0, 11, 5]   # one slice; slices that end in the middle of rows


def test_the_prediction_rules_all_come_into_play():
    """(over the pictures of the clip: every macroblock kind, both P_Skip outcomes, macroblocks at the right border - C
    replaced by D - and in the top row - only A)"""
    mbw, mbh = W // 16, H // 16
    seen = {}
    dec = [np.zeros(W * H * 3 // 2, np.uint8)] * FRAMES
    for k in range(1, FRAMES):
        _, m, kinds = expected(dec, k, W, H)
        for addr, kind in enumerate(kinds):
            mx, my = addr % mbw, addr // mbw
            if kind == "skip":
                kind = "skip_zero" if m.mv[my * 4][mx * 4] == (0, 0) else "skip_predicted"
            seen[kind] = seen.get(kind, 0) + 1
            if my > 0 and mx + 1 == mbw:
                seen["right border"] = seen.get("right border", 0) + 1
            if my == 0 and mx > 0:
                seen["top row"] = seen.get("top row", 0) + 1
    for kind in ("16x16", "16x8", "8x16", "8x8", "zero", "skip_zero", "skip_predicted", "right border", "top row"):
        assert seen.get(kind, 0) > 5, seen


@pytest.mark.parametrize("cabac", [1, 0])
@pytest.mark.parametrize("slice_mbs", SLICE_MBS)
def test_oracle_predicts_vectors_as_the_clause_says(hw, oracle, tmp_path, cabac, slice_mbs):
    path = str(tmp_path / "mvp.mp4")
    gold = make_clip(path, width=W, height=H, cabac=cabac, **clip_args(slice_mbs))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == FRAMES
    dec = [f[0] for f in frames]
    for k in range(1, FRAMES):
        assert np.array_equal(dec[k], expected(dec, k, W, H, slice_mbs)[0]), "picture %d" % k
        assert np.array_equal(dec[k], gold[k])


@pytest.mark.gpu
@pytest.mark.parametrize("cabac", [1, 0])
@pytest.mark.parametrize("slice_mbs", SLICE_MBS)
def test_hip_path_predicts_vectors_as_the_clause_says(hw, tmp_path, cabac, slice_mbs):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = str(tmp_path / "mvp.mp4")
    make_clip(path, width=W, height=H, yuv=False, cabac=cabac, **clip_args(slice_mbs))
    dec = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(FRAMES)))
    for k in range(1, FRAMES):
        assert np.array_equal(dec[k], expected(dec, k, W, H, slice_mbs)[0]), "picture %d" % k


# ---- spatial direct prediction (8.4.1.2.2) ---------------------------------------------------------------------------------
# --global-mv 10: B pictures as a checkerboard of coded macroblocks (list 0, list 1 or both by address and picture number,
# one vector per list and picture) and B_Skip macroblocks.  A B_Skip macroblock takes, per list, the smallest non-negative
# reference index among its neighbours A, B, C (both negative: reference 0 in both lists and zero vectors), the median
# prediction for it as its vector - or zero where the reference index is 0 and the co-located macroblock of the anchor
# behind "does not move" (reference 0, both vector components within +-1).
def min_positive(a, b):
    return min(a, b) if (a >= 0 and b >= 0) else max(a, b)


def anchor_mv(k, w, h):
    """corpus/h264_encoder.c, --global-mv 10: every third anchor hardly moves."""
    return (1, -1) if k % 9 == 0 else global_mv_of(k, w, h)


def expected_spatial(dec, d, w, h, implicit, nrefs=1, motion_out=None):
    """nrefs 2: the lists of a B picture between the anchors a and a + 3 are [a, a + 3] and [a + 3, a] (8.2.4.2.3), and
    the coded macroblocks use index (addr / 2 + d) % 2 in list 0, the other one in list 1."""
    if d % 3 == 0:
        return np.concatenate([p.ravel() for p in shifted(dec[d - 3], anchor_mv(d, w, h), w, h)])
    mbw, mbh, a = w // 16, h // 16, d - d % 3
    lists = [Motion(mbw, mbh), Motion(mbw, mbh)]
    mv_col = anchor_mv(a + 3, w, h)               # every macroblock of the anchor behind: reference 0, this vector
    col_zero = abs(mv_col[0]) <= 1 and abs(mv_col[1]) <= 1
    gm = (global_mv_of(d, w, h), global_mv_of(d + 17, w, h))
    pics = ([a, a + 3], [a + 3, a])               # picture behind (list, reference index)
    out = [np.zeros((h, w), np.uint8), np.zeros((h // 2, w // 2), np.uint8), np.zeros((h // 2, w // 2), np.uint8)]
    cache = {}

    def pred(lst, use):
        pic = pics[lst][use[0]]
        if (pic, use[1]) not in cache:
            cache[(pic, use[1])] = shifted(dec[pic], use[1], w, h)
        return cache[(pic, use[1])]
    for addr in range(mbw * mbh):
        mx, my = addr % mbw, addr // mbw
        x, y = mx * 4, my * 4
        if (mx + my) & 1:
            ref = []
            for m in lists:
                nb = [m.at(x - 1, y), m.at(x, y - 1), m.at(x + 4, y - 1)]
                if nb[2] is None:
                    nb[2] = m.at(x - 1, y - 1)
                r = [n[0] if n is not None else -1 for n in nb]
                ref.append(min_positive(r[0], min_positive(r[1], r[2])))
            if ref[0] < 0 and ref[1] < 0:
                use = [(0, (0, 0)), (0, (0, 0))]
            else:
                use = []
                for lst, m in enumerate(lists):
                    if ref[lst] < 0:
                        use.append((-1, (0, 0)))
                    elif ref[lst] == 0 and col_zero:
                        use.append((0, (0, 0)))
                    else:
                        use.append((ref[lst], predict(m, x, y, 4, ref[lst])))
        else:
            kind = 1 + (d + addr) % 3   # 1 list 0, 2 list 1, 3 both (corpus/h264_encoder.c: decide_b)
            r0, r1 = (addr // 2 + d) % nrefs, (addr // 2 + d + 1) % nrefs
            use = [(r0, gm[0]) if kind != 2 else (-1, (0, 0)), (r1, gm[1]) if kind != 1 else (-1, (0, 0))]
        for lst in range(2):
            lists[lst].put(x, y, 4, 4, use[lst][0], use[lst][1])
        if motion_out is not None:   # which pictures the macroblock predicts from, with which vectors
            motion_out.append(sorted((pics[lst][use[lst][0]], use[lst][1]) for lst in range(2) if use[lst][0] >= 0))
        for pl, s in ((0, 16), (1, 8), (2, 8)):
            reg = (slice(my * s, (my + 1) * s), slice(mx * s, (mx + 1) * s))
            if use[0][0] >= 0 and use[1][0] >= 0:
                p0, p1 = pred(0, use[0])[pl][reg], pred(1, use[1])[pl][reg]
                w0, w1 = implicit_weights(2 * d, 2 * pics[0][use[0][0]], 2 * pics[1][use[1][0]]) if implicit else (32, 32)
                out[pl][reg] = weighted_bi(p0, p1, 5, w0, w1, 0, 0) if implicit else ((p0.astype(np.int64) + p1 + 1) >> 1).astype(np.uint8)
            else:
                lst = 0 if use[0][0] >= 0 else 1
                out[pl][reg] = pred(lst, use[lst])[pl][reg]
    return np.concatenate([p.ravel() for p in out])


def spatial_clip_args(implicit, nrefs=1):
    # (filter off: neighbouring macroblocks differ in which lists they use, their edges would get bS 1)
    return dict(frames=B_FRAMES, gop=B_FRAMES, bframes=2, refs=nrefs, global_mv=10, implicit_wp=1 if implicit else 0, deblock=1)


@pytest.mark.parametrize("cabac", [1, 0])
@pytest.mark.parametrize("implicit", [False, True])
@pytest.mark.parametrize("nrefs", [1])   # (the corpus encoder keeps one active reference per list in B slices)
def test_oracle_derives_spatial_direct_motion_as_the_clause_says(hw, oracle, tmp_path, cabac, implicit, nrefs):
    path = str(tmp_path / "sd.mp4")
    gold = make_clip(path, width=W, height=H, cabac=cabac, **spatial_clip_args(implicit, nrefs))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == B_FRAMES
    dec = [f[0] for f in frames]
    for d in range(1, B_FRAMES):
        assert np.array_equal(dec[d], expected_spatial(dec, d, W, H, implicit, nrefs)), "picture %d (%s)" % (d, "P" if d % 3 == 0 else "B")
        assert np.array_equal(dec[d], gold[d])


@pytest.mark.gpu
@pytest.mark.parametrize("implicit", [False, True])
@pytest.mark.parametrize("nrefs", [1])
def test_hip_path_derives_spatial_direct_motion_as_the_clause_says(hw, tmp_path, implicit, nrefs):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = str(tmp_path / "sd.mp4")
    make_clip(path, width=W, height=H, yuv=False, **spatial_clip_args(implicit, nrefs))
    dec = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(B_FRAMES)))
    for d in range(1, B_FRAMES):
        assert np.array_equal(dec[d], expected_spatial(dec, d, W, H, implicit, nrefs)), "picture %d (%s)" % (d, "P" if d % 3 == 0 else "B")


# ---- boundary strength 1 between differently predicted macroblocks of a B picture (8.7.2.1) --------------------------------
# The same pictures with the filter on.  Nothing is coded, so an edge between two macroblocks has bS 1 when they predict
# from different reference PICTURES (whatever the lists and indices that name them) or from a different number of them, or
# when a vector differs by four or more quarter samples from its counterpart for the same picture; otherwise 0.
def expected_spatial_filtered(dec, d, w, h, qp, deblock):
    from tests.test_deblock_independent import chroma_qp, filter_line, thresholds
    motion = []
    unfiltered = expected_spatial(dec, d, w, h, False, 1, motion)
    if d % 3 == 0:
        return unfiltered   # (anchors: one vector, no edge has a strength)
    mbw, mbh = w // 16, h // 16

    def strength(p, q):
        if [m[0] for m in p] != [m[0] for m in q]:
            return 1         # different pictures, or a different number of them
        return 1 if any(abs(a - b) >= 4 for mp, mq in zip(p, q) for a, b in zip(mp[1], mq[1])) else 0
    planes_ = [pl.astype(np.int64) for pl in planes(unfiltered, w, h)]
    th = [thresholds(qp, 0, 0), thresholds(chroma_qp(qp, 0), 0, 0), thresholds(chroma_qp(qp, 0), 0, 0)]
    for my in range(mbh):
        for mx in range(mbw):
            for vertical in (True, False):
                if (mx == 0) if vertical else (my == 0):
                    continue
                other = motion[my * mbw + mx - 1] if vertical else motion[(my - 1) * mbw + mx]
                if not strength(other, motion[my * mbw + mx]):
                    continue
                for a, (alpha, beta, tc0), mb, chroma in zip(planes_, th, (16, 8, 8), (False, True, True)):
                    reach = 2 if chroma else 4
                    for i in range(mb):
                        px = [0] * 8
                        if vertical:
                            px[4 - reach:4 + reach] = [int(t) for t in a[my * mb + i, mx * mb - reach:mx * mb + reach]]
                            filter_line(px, chroma, alpha, beta, tc0)
                            a[my * mb + i, mx * mb - reach:mx * mb + reach] = px[4 - reach:4 + reach]
                        else:
                            px[4 - reach:4 + reach] = [int(t) for t in a[my * mb - reach:my * mb + reach, mx * mb + i]]
                            filter_line(px, chroma, alpha, beta, tc0)
                            a[my * mb - reach:my * mb + reach, mx * mb + i] = px[4 - reach:4 + reach]
    return np.concatenate([np.clip(pl, 0, 255).astype(np.uint8).ravel() for pl in planes_])


@pytest.mark.parametrize("cabac", [1, 0])
@pytest.mark.parametrize("qp", [30, 40])
def test_oracle_rates_b_picture_edges_as_the_clause_says(hw, oracle, tmp_path, cabac, qp):
    path = str(tmp_path / "sdf.mp4")
    gold = make_clip(path, width=W, height=H, cabac=cabac, qp=qp, **dict(spatial_clip_args(False), deblock=0))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == B_FRAMES
    dec = [f[0] for f in frames]
    moved = 0
    for d in range(1, B_FRAMES):
        want = expected_spatial_filtered(dec, d, W, H, qp, 0)
        assert np.array_equal(dec[d], want), "picture %d (%s)" % (d, "P" if d % 3 == 0 else "B")
        moved += int((want != expected_spatial(dec, d, W, H, False)).sum())
    assert moved > 0


@pytest.mark.gpu
@pytest.mark.parametrize("qp", [30, 40])
def test_hip_path_rates_b_picture_edges_as_the_clause_says(hw, tmp_path, qp):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = str(tmp_path / "sdf.mp4")
    make_clip(path, width=W, height=H, yuv=False, qp=qp, **dict(spatial_clip_args(False), deblock=0))
    dec = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(B_FRAMES)))
    for d in range(1, B_FRAMES):
        assert np.array_equal(dec[d], expected_spatial_filtered(dec, d, W, H, qp, 0)), "picture %d (%s)" % (d, "P" if d % 3 == 0 else "B")


@pytest.mark.gpu
def test_hip_path_predicts_vectors_in_a_larger_picture_as_the_clause_says(hw, tmp_path):
    """640 x 368 = 40 x 23 macroblocks: k_mv's row pipeline over six bands (line buffers handed from band to band)."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    w, h = 640, 368
    path = str(tmp_path / "mvpl.mp4")
    make_clip(path, width=w, height=h, yuv=False, **dict(clip_args(37), frames=3, gop=3))
    dec = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(3)))
    for k in range(1, 3):   # (a minute of Python: a whole-picture interpolation per distinct vector)
        assert np.array_equal(dec[k], expected(dec, k, w, h, 37)[0]), "picture %d" % k


# ---- vector prediction in B pictures, list by list (--global-mv 12) --------------------------------------------------------
# B macroblocks of one 16x16 or two 16x8 / 8x16 partitions; which lists a partition uses and the vector difference per list
# follow from address, partition and picture number.  Each list is predicted on its own: a neighbour that does not use
# the list counts as reference -1 with a zero vector.  (On this path the HOST derives the vectors: B pictures do not go
# through k_mv.)
def expected_b_partitions(dec, d, w, h):
    if d % 3 == 0:
        return np.concatenate([p.ravel() for p in shifted(dec[d - 3], global_mv_of(d, w, h), w, h)])
    mbw, mbh, a = w // 16, h // 16, d - d % 3
    lists = [Motion(mbw, mbh), Motion(mbw, mbh)]
    refs = (dec[a], dec[a + 3])
    out = [np.zeros((h, w), np.uint8), np.zeros((h // 2, w // 2), np.uint8), np.zeros((h // 2, w // 2), np.uint8)]
    cache = {}

    def pred(lst, mv):
        if (lst, mv) not in cache:
            cache[(lst, mv)] = shifted(refs[lst], mv, w, h)
        return cache[(lst, mv)]
    for addr in range(mbw * mbh):
        mx, my = addr % mbw, addr // mbw
        x, y = mx * 4, my * 4
        shape = (addr + d) % 3   # 0 16x16, 1 16x8, 2 8x16
        for part in range(2 if shape else 1):
            use = (addr * 2 + d + part) % 3   # 0 list 0, 1 list 1, 2 both
            if shape == 0:
                rect, first = (x, y, 4, 4), None
            elif shape == 1:
                rect, first = (x, y + 2 * part, 4, 2), ("B" if part == 0 else "A")
            else:
                rect, first = (x + 2 * part, y, 2, 4), ("A" if part == 0 else "C")
            mv = [None, None]
            for lst in range(2):
                if use == 2 or use == lst:
                    px, py = predict(lists[lst], rect[0], rect[1], rect[2], 0, first)
                    mv[lst] = (px + (addr * 5 + d * 3 + part * 4 + lst * 2) % 13 - 6, py + (addr * 7 + d + part * 2 + lst * 5) % 9 - 4)
                    lists[lst].put(rect[0], rect[1], rect[2], rect[3], 0, mv[lst])
                else:
                    lists[lst].put(rect[0], rect[1], rect[2], rect[3], -1, (0, 0))
            for pl, s in ((0, 4), (1, 2), (2, 2)):
                reg = (slice(rect[1] * s, (rect[1] + rect[3]) * s), slice(rect[0] * s, (rect[0] + rect[2]) * s))
                if mv[0] is not None and mv[1] is not None:
                    out[pl][reg] = ((pred(0, mv[0])[pl][reg].astype(np.int64) + pred(1, mv[1])[pl][reg] + 1) >> 1).astype(np.uint8)
                else:
                    lst = 0 if mv[0] is not None else 1
                    out[pl][reg] = pred(lst, mv[lst])[pl][reg]
    return np.concatenate([p.ravel() for p in out])


def b_partition_args():
    return dict(frames=19, gop=19, bframes=2, refs=1, global_mv=12, deblock=1)


@pytest.mark.parametrize("cabac", [1, 0])
def test_oracle_predicts_b_partition_vectors_as_the_clause_says(hw, oracle, tmp_path, cabac):
    path = str(tmp_path / "bp.mp4")
    gold = make_clip(path, width=W, height=H, cabac=cabac, **b_partition_args())
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == 19
    dec = [f[0] for f in frames]
    for d in range(1, 19):
        assert np.array_equal(dec[d], expected_b_partitions(dec, d, W, H)), "picture %d (%s)" % (d, "P" if d % 3 == 0 else "B")
        assert np.array_equal(dec[d], gold[d])


@pytest.mark.gpu
def test_hip_path_predicts_b_partition_vectors_as_the_clause_says(hw, tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = str(tmp_path / "bp.mp4")
    make_clip(path, width=W, height=H, yuv=False, **b_partition_args())
    dec = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(19)))
    for d in range(1, 19):
        assert np.array_equal(dec[d], expected_b_partitions(dec, d, W, H)), "picture %d (%s)" % (d, "P" if d % 3 == 0 else "B")

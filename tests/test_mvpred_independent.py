"""Motion-vector prediction (8.4.1.3) and the P_Skip inference (8.4.1.1) checked against a restatement that shares nothing
with the oracle or the kernels.

`--global-mv 7` (corpus/h264_encoder.c) writes P pictures in which every macroblock is one 16x16 partition whose reference
index and vector DIFFERENCE follow from its address and the picture's number; every fourth macroblock is P_Skip, some
have reference 0 and vector 0 outright (the neighbour the P_Skip rule looks for).  Nothing
else is coded and the in-loop filter is off.  The vectors themselves are then whatever the median prediction makes of the
differences - in the encoder (which uses the oracle's function), in the oracle, and in the HIP path's k_mv, which gets
the differences and derives the vectors on the GPU.  This test derives them a third time, in Python, from the clause:
neighbours A (left), B (above), C (above-right, or D above-left where C does not exist), the rule that a single
neighbour with the same reference index wins, the median otherwise, "only A available" and the P_Skip conditions - and
builds the picture from the two reference pictures with the interpolation of tests/test_inter_independent.py.  Compared
with the oracle's decode (CPU) and the HIP path's (GPU)."""
import numpy as np
import pytest

from tests.corpus_util import make_clip
from tests.test_inter_independent import planes, shift_chroma, shift_luma

W, H = 112, 80   # 7 x 5 macroblocks
FRAMES = 14


def median(a, b, c):
    return a + b + c - min(a, b, c) - max(a, b, c)


def predict(mvs, refs, mx, my, mbw, ref):
    """8.4.1.3 for a 16x16 partition of the macroblock at (mx, my): mvs / refs hold what the macroblocks decoded so far
    ended up with (refs: -1 = not decoded / outside)."""
    def nb(x, y):
        if x < 0 or y < 0 or x >= mbw or refs[y][x] < 0:
            return None
        return refs[y][x], mvs[y][x]
    a, b = nb(mx - 1, my), nb(mx, my - 1)
    c = nb(mx + 1, my - 1)
    if c is None:
        c = nb(mx - 1, my - 1)   # (8.4.1.3.2: C is replaced by D when C is not available)
    if b is None and c is None and a is not None:
        return a[1]              # only A is available
    cand = [(n[0], n[1]) if n is not None else (-1, (0, 0)) for n in (a, b, c)]
    same = [n for n in cand if n[0] == ref]
    if len(same) == 1:
        return same[0][1]
    return (median(cand[0][1][0], cand[1][1][0], cand[2][1][0]), median(cand[0][1][1], cand[1][1][1], cand[2][1][1]))


def skip_vector(mvs, refs, mx, my, mbw):
    """8.4.1.1: zero when A or B does not exist, or is (reference 0, vector 0); the prediction for reference 0 otherwise."""
    def nb(x, y):
        if x < 0 or y < 0 or x >= mbw or refs[y][x] < 0:
            return None
        return refs[y][x], mvs[y][x]
    a, b = nb(mx - 1, my), nb(mx, my - 1)
    if a is None or b is None or a == (0, (0, 0)) or b == (0, (0, 0)):
        return (0, 0)
    return predict(mvs, refs, mx, my, mbw, 0)


def expected(dec, k, w, h):
    """Picture k from the decoded pictures before it: list 0 of a P picture = the previous pictures, nearest first."""
    mbw, mbh = w // 16, h // 16
    nref = min(2, k)
    mvs = [[(0, 0)] * mbw for _ in range(mbh)]
    refs = [[-1] * mbw for _ in range(mbh)]
    out = [np.zeros((h, w), np.uint8), np.zeros((h // 2, w // 2), np.uint8), np.zeros((h // 2, w // 2), np.uint8)]
    cache = {}
    for addr in range(mbw * mbh):
        mx, my = addr % mbw, addr // mbw
        if addr % 4 == 3:
            r, mv = 0, skip_vector(mvs, refs, mx, my, mbw)
        elif addr % 7 == 5:
            r, mv = 0, (0, 0)   # (the encoder codes whatever difference that takes)
        else:
            r = (addr * 3 + k) % nref
            px, py = predict(mvs, refs, mx, my, mbw, r)
            mv = (px + (addr * 5 + k * 3) % 13 - 6, py + (addr * 7 + k) % 9 - 4)
        mvs[my][mx], refs[my][mx] = mv, r
        if (r, mv) not in cache:
            y, u, v = planes(dec[k - 1 - r], w, h)
            cache[(r, mv)] = (shift_luma(y, *mv), shift_chroma(u, *mv), shift_chroma(v, *mv))
        for pl, mb in ((0, 16), (1, 8), (2, 8)):
            out[pl][my * mb:(my + 1) * mb, mx * mb:(mx + 1) * mb] = cache[(r, mv)][pl][my * mb:(my + 1) * mb, mx * mb:(mx + 1) * mb]
    return np.concatenate([p.ravel() for p in out]), mvs, refs


def clip_args():
    return dict(frames=FRAMES, gop=FRAMES, bframes=0, refs=2, global_mv=7, deblock=1)


def test_the_prediction_rules_all_come_into_play():
    """(over the pictures of the clip: the single-match rule, the median, C replaced by D, only A, both P_Skip outcomes)"""
    mbw, mbh = W // 16, H // 16
    single = med = c_from_d = only_a = skip_zero = skip_pred = 0
    dec = [np.zeros(W * H * 3 // 2, np.uint8)] * FRAMES
    for k in range(1, FRAMES):
        _, mvs, refs = expected(dec, k, W, H)
        for addr in range(mbw * mbh):
            mx, my = addr % mbw, addr // mbw
            if my > 0 and mx + 1 >= mbw:
                c_from_d += 1
            if my == 0 and mx > 0:
                only_a += 1
            if addr % 4 == 3:
                if mvs[my][mx] == (0, 0):
                    skip_zero += 1
                else:
                    skip_pred += 1
            elif my > 0 and mx > 0:
                r = refs[my][mx]
                nbs = [refs[my][mx - 1], refs[my - 1][mx], refs[my - 1][mx + 1] if mx + 1 < mbw else refs[my - 1][mx - 1]]
                if sum(1 for n in nbs if n == r) == 1:
                    single += 1
                else:
                    med += 1
    assert min(single, med, c_from_d, only_a, skip_zero, skip_pred) > 5, (single, med, c_from_d, only_a, skip_zero, skip_pred)


@pytest.mark.parametrize("cabac", [1, 0])
def test_oracle_predicts_vectors_as_the_clause_says(hw, oracle, tmp_path, cabac):
    path = str(tmp_path / "mvp.mp4")
    gold = make_clip(path, width=W, height=H, cabac=cabac, **clip_args())
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == FRAMES
    dec = [f[0] for f in frames]
    for k in range(1, FRAMES):
        assert np.array_equal(dec[k], expected(dec, k, W, H)[0]), "picture %d" % k
        assert np.array_equal(dec[k], gold[k])


@pytest.mark.gpu
@pytest.mark.parametrize("cabac", [1, 0])
def test_hip_path_predicts_vectors_as_the_clause_says(hw, tmp_path, cabac):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = str(tmp_path / "mvp.mp4")
    make_clip(path, width=W, height=H, yuv=False, cabac=cabac, **clip_args())
    dec = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(FRAMES)))
    for k in range(1, FRAMES):
        assert np.array_equal(dec[k], expected(dec, k, W, H)[0]), "picture %d" % k

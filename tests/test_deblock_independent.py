"""The in-loop filter (8.7) for boundary strength 1 checked against a restatement that shares nothing with the oracle or
the kernels - the tables included: they are typed a second time here.

`--global-mv 2` (corpus/h264_encoder.c) writes P pictures whose macroblocks take one of two vectors in a checkerboard,
reference 0, and code nothing.  So the picture before filtering is known (tests/test_inter_independent.py: a mosaic of
two shifted copies of the previous picture), every macroblock edge inside the picture separates different motion and has
bS = 1, no inner edge is filtered, and every macroblock has the slice's QP.  What the filter makes of that is computed
here macroblock by macroblock in decoding order (left edge, then top edge; luma, then both chroma planes) from 8.7.2.3 /
8.7.2.4 with the thresholds of Tables 8-16 / 8-17 and the chroma QP of Table 8-15, for several QPs, filter offsets and a
chroma QP offset, and compared with the oracle's decode (CPU) and the HIP path's (GPU)."""
import numpy as np
import pytest

from tests.corpus_util import make_clip
from tests.test_inter_independent import global_mv_of, planes, shift_chroma, shift_luma

W, H = 176, 144
FRAMES = 20

# Table 8-16 (alpha', beta' by indexA / indexB) and the bS = 1 column of Table 8-17 (tC0' by indexA)
ALPHA = [0] * 16 + [4, 4, 5, 6, 7, 8, 9, 10, 12, 13, 15, 17, 20, 22, 25, 28, 32, 36, 40, 45, 50, 56, 63, 71, 80, 90, 101, 113, 127, 144,
                    162, 182, 203, 226, 255, 255]
BETA = [0] * 16 + [2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13, 14, 14, 15, 15, 16, 16, 17, 17,
                   18, 18]
TC0_BS1 = [0] * 23 + [1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13]
# Table 8-15: QPc for qPI = 30 .. 51 (below 30: QPc = qPI)
QPC_TAIL = [29, 30, 31, 32, 32, 33, 34, 34, 35, 35, 36, 36, 37, 37, 37, 38, 38, 38, 39, 39, 39, 39]
assert len(ALPHA) == len(BETA) == len(TC0_BS1) == 52 and len(QPC_TAIL) == 22


def clip3(lo, hi, x):
    return max(lo, min(hi, x))


def filter_line_strong(px, chroma, alpha, beta):
    """One line across an edge with bS = 4 (8.7.2.4): px = [p3 p2 p1 p0 q0 q1 q2 q3] (ints), filtered in place."""
    p3, p2, p1, p0, q0, q1, q2, q3 = px
    if not (abs(p0 - q0) < alpha and abs(p1 - p0) < beta and abs(q1 - q0) < beta):
        return
    if chroma:
        px[3] = (2 * p1 + p0 + q1 + 2) >> 2
        px[4] = (2 * q1 + q0 + p1 + 2) >> 2
        return
    small = abs(p0 - q0) < ((alpha >> 2) + 2)
    if abs(p2 - p0) < beta and small:
        px[3] = (p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3
        px[2] = (p2 + p1 + p0 + q0 + 2) >> 2
        px[1] = (2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3
    else:
        px[3] = (2 * p1 + p0 + q1 + 2) >> 2
    if abs(q2 - q0) < beta and small:
        px[4] = (p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3
        px[5] = (p0 + q0 + q1 + q2 + 2) >> 2
        px[6] = (2 * q3 + 3 * q2 + q1 + q0 + p0 + 4) >> 3
    else:
        px[4] = (2 * q1 + q0 + p1 + 2) >> 2


def filter_line(px, chroma, alpha, beta, tc0):
    """One line of samples across an edge, bS = 1 (8.7.2.3): px = [p3 p2 p1 p0 q0 q1 q2 q3] (ints), filtered in place.
    tc0 None: bS = 4."""
    if tc0 is None:
        return filter_line_strong(px, chroma, alpha, beta)
    p2, p1, p0, q0, q1, q2 = px[1], px[2], px[3], px[4], px[5], px[6]
    if not (abs(p0 - q0) < alpha and abs(p1 - p0) < beta and abs(q1 - q0) < beta):
        return
    if chroma:
        tc = tc0 + 1
    else:
        ap, aq = abs(p2 - p0), abs(q2 - q0)
        tc = tc0 + (1 if ap < beta else 0) + (1 if aq < beta else 0)
        if ap < beta:
            px[2] = p1 + clip3(-tc0, tc0, (p2 + ((p0 + q0 + 1) >> 1) - (p1 << 1)) >> 1)
        if aq < beta:
            px[5] = q1 + clip3(-tc0, tc0, (q2 + ((p0 + q0 + 1) >> 1) - (q1 << 1)) >> 1)
    delta = clip3(-tc, tc, (((q0 - p0) << 2) + (p1 - q1) + 4) >> 3)
    px[3] = clip3(0, 255, p0 + delta)
    px[4] = clip3(0, 255, q0 - delta)


def deblock_plane(img, mb, chroma, alpha, beta, tc0, slice_mbs=0):
    """All macroblock edges inside the plane, bS = 1, macroblock by macroblock in raster order: the left edge's lines,
    then the top edge's.  `mb`: macroblock size in this plane (16 luma, 8 chroma).  slice_mbs > 0: a new slice every so
    many macroblocks and disable_deblocking_filter_idc = 2 - an edge shared with another slice is left alone (8.7)."""
    a = img.astype(np.int64)
    h, w = a.shape
    mbw = w // mb
    reach = 2 if chroma else 4   # samples of either side a line may look at (chroma lines are shorter: p1 .. q1)
    same_slice = lambda p, q: not slice_mbs or p // slice_mbs == q // slice_mbs   # noqa: E731
    for my in range(h // mb):
        for mx in range(w // mb):
            x0, y0 = mx * mb, my * mb
            addr = my * mbw + mx
            if mx > 0 and not same_slice(addr, addr - 1):
                pass
            elif mx > 0:
                for y in range(y0, y0 + mb):
                    px = [0] * 8
                    px[4 - reach:4 + reach] = [int(v) for v in a[y, x0 - reach:x0 + reach]]
                    filter_line(px, chroma, alpha, beta, tc0)
                    a[y, x0 - reach:x0 + reach] = px[4 - reach:4 + reach]
            if my > 0 and same_slice(addr, addr - mbw):
                for x in range(x0, x0 + mb):
                    px = [0] * 8
                    px[4 - reach:4 + reach] = [int(v) for v in a[y0 - reach:y0 + reach, x]]
                    filter_line(px, chroma, alpha, beta, tc0)
                    a[y0 - reach:y0 + reach, x] = px[4 - reach:4 + reach]
    return a.astype(np.uint8)


def thresholds(qp, alpha_div2, beta_div2):
    ia, ib = clip3(0, 51, qp + 2 * alpha_div2), clip3(0, 51, qp + 2 * beta_div2)
    return ALPHA[ia], BETA[ib], TC0_BS1[ia]


def chroma_qp(qp, offset):
    qpi = clip3(0, 51, qp + offset)
    return qpi if qpi < 30 else QPC_TAIL[qpi - 30]


def mosaic(prev, k, w, h):
    """The picture before the filter: macroblock (x, y) comes from the previous picture shifted by the picture's vector,
    or by that vector + (9, -6) where x + y is odd."""
    mva = global_mv_of(k, w, h)
    mvb = (mva[0] + 9, mva[1] - 6)
    out = []
    for plane, mb, shift in zip(planes(prev, w, h), (16, 8, 8), (shift_luma, shift_chroma, shift_chroma)):
        pa, pb = shift(plane, *mva), shift(plane, *mvb)
        ph, pw = plane.shape
        odd = (np.add.outer(np.arange(ph) // mb, np.arange(pw) // mb) & 1).astype(bool)
        out.append(np.where(odd, pb, pa))
    return out


def expected(prev, k, w, h, qp, alpha_div2, beta_div2, cqo, slice_mbs=0):
    y, u, v = mosaic(prev, k, w, h)
    y = deblock_plane(y, 16, False, *thresholds(qp, alpha_div2, beta_div2), slice_mbs=slice_mbs)
    # chroma edges: the thresholds come from the chroma QP of the two macroblocks (here the same) in place of QP_Y;
    # Cb with chroma_qp_index_offset, Cr with second_chroma_qp_index_offset - which the corpus encoder's PPS sets one
    # lower than the first when that is positive (corpus/h264_encoder.c: write_pps; the clips here carry the extension)
    u = deblock_plane(u, 8, True, *thresholds(chroma_qp(qp, cqo), alpha_div2, beta_div2), slice_mbs=slice_mbs)
    v = deblock_plane(v, 8, True, *thresholds(chroma_qp(qp, cqo - 1 if cqo > 0 else cqo), alpha_div2, beta_div2), slice_mbs=slice_mbs)
    return np.concatenate([y.ravel(), u.ravel(), v.ravel()])


# (QP, FilterOffsetA / 2, FilterOffsetB / 2, chroma_qp_index_offset)
SETTINGS = [(26, 0, 0, 0), (36, 0, 0, 0), (44, 0, 0, 0), (30, 2, -1, 0), (38, -2, 3, 0), (34, 0, 0, -2), (41, 1, 1, 4), (18, 0, 0, 0)]


def clip_args(qp, a, b, cqo):
    return dict(frames=FRAMES, gop=FRAMES, bframes=0, refs=1, global_mv=2, qp=qp, alpha=a, beta=b, chroma_qp_offset=cqo)


@pytest.mark.parametrize("qp,a,b,cqo", SETTINGS)
@pytest.mark.parametrize("cabac", [1, 0])
def test_oracle_filters_known_edges_as_the_clauses_say(hw, oracle, tmp_path, qp, a, b, cqo, cabac):
    path = str(tmp_path / "dbk.mp4")
    gold = make_clip(path, width=W, height=H, cabac=cabac, **clip_args(qp, a, b, cqo))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == FRAMES
    touched = untouched = 0
    for k in range(1, FRAMES):
        want = expected(frames[k - 1][0], k, W, H, qp, a, b, cqo)
        assert np.array_equal(frames[k][0], want), "picture %d" % k
        assert np.array_equal(frames[k][0], gold[k])
        before = mosaic(frames[k - 1][0], k, W, H)[0]
        after = want[:W * H].reshape(H, W)
        for x0 in range(16, W, 16):   # samples next to the vertical luma edges: did the filter move them?
            moved = (before[:, x0 - 1] != after[:, x0 - 1]) | (before[:, x0] != after[:, x0])
            touched += int(moved.sum())
            untouched += int((~moved).sum())
    # (both outcomes of the alpha / beta tests must occur, or the comparison says little; QP 18: alpha' = 5, most lines fail)
    assert touched > 0 and untouched > 0


@pytest.mark.parametrize("slice_mbs", [11, 17, 40])
def test_oracle_leaves_slice_edges_alone_when_told_to(hw, oracle, tmp_path, slice_mbs):
    """disable_deblocking_filter_idc = 2 with slices that end in the middle of rows: edges between macroblocks of different
    slices are not filtered, all others are."""
    qp, a, b, cqo = 36, 1, 0, 0
    path = str(tmp_path / "dbk_sl.mp4")
    make_clip(path, width=W, height=H, yuv=False, **dict(clip_args(qp, a, b, cqo), deblock=2, slice_mbs=slice_mbs))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > FRAMES
    differs = 0
    for k in range(1, FRAMES):
        want = expected(frames[k - 1][0], k, W, H, qp, a, b, cqo, slice_mbs)
        assert np.array_equal(frames[k][0], want), "picture %d" % k
        differs += int((want != expected(frames[k - 1][0], k, W, H, qp, a, b, cqo)).sum())
    assert differs > 0   # (leaving those edges alone shows)


@pytest.mark.gpu
@pytest.mark.parametrize("slice_mbs", [11, 17, 40])
def test_hip_path_leaves_slice_edges_alone_when_told_to(hw, tmp_path, slice_mbs):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    qp, a, b, cqo = 36, 1, 0, 0
    path = str(tmp_path / "dbk_sl.mp4")
    make_clip(path, width=W, height=H, yuv=False, **dict(clip_args(qp, a, b, cqo), deblock=2, slice_mbs=slice_mbs))
    got = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(FRAMES)))
    for k in range(1, FRAMES):
        assert np.array_equal(got[k], expected(got[k - 1], k, W, H, qp, a, b, cqo, slice_mbs)), "picture %d" % k


def test_every_row_of_the_threshold_tables(hw, oracle, tmp_path):
    """QP 15 .. 51 without offsets: indexA = indexB = QP for luma (every non-zero row of Tables 8-16 and 8-17's bS = 1
    column) and the chroma QP of Table 8-15 for the chroma planes."""
    for qp in range(15, 52):
        path = str(tmp_path / ("dbk_qp%d.mp4" % qp))
        args = dict(clip_args(qp, 0, 0, 0), frames=4, gop=4)
        make_clip(path, width=W, height=H, yuv=False, **args)
        frames, _ = oracle.decode_file(path, hw.index_video(path))
        for k in range(1, 4):
            assert np.array_equal(frames[k][0], expected(frames[k - 1][0], k, W, H, qp, 0, 0, 0)), "QP %d, picture %d" % (qp, k)


# ---- boundary strength 4: a checkerboard of inter and I_PCM macroblocks (--global-mv 3) ------------------------------
def pcm_delta(k, pl, x, y):
    """corpus/h264_encoder.c: global_pcm_delta."""
    e = (x * 7 + y * 13 + k * 5 + pl * 3) % 13 - 6
    return e + 24 if ((x >> 2) + (y >> 2) + k) % 5 == 0 else e


def mosaic_pcm(prev, k, w, h):
    """Before the filter: every macroblock is the previous picture shifted by the picture's vector; where x + y is odd it
    is an I_PCM macroblock carrying those samples plus pcm_delta."""
    mv = global_mv_of(k, w, h)
    out = []
    for pl, (plane, mb, shift) in enumerate(zip(planes(prev, w, h), (16, 8, 8), (shift_luma, shift_chroma, shift_chroma))):
        pa = shift(plane, *mv).astype(np.int64)
        ph, pw = plane.shape
        ys, xs = np.arange(ph), np.arange(pw)
        delta = np.array([[pcm_delta(k, pl, int(x), int(y)) for x in xs] for y in ys])
        odd = (np.add.outer(ys // mb, xs // mb) & 1).astype(bool)
        out.append(np.where(odd, np.clip(pa + delta, 0, 255), pa).astype(np.uint8))
    return out


def expected_pcm(prev, k, w, h, qp, alpha_div2, beta_div2, cqo):
    """Every macroblock edge inside the picture lies between an I_PCM macroblock (qP = 0 for luma, QPc of 0 for chroma:
    8.7.2.2) and an inter macroblock with the slice's QP: bS = 4, thresholds from the average of the two."""
    y, u, v = mosaic_pcm(prev, k, w, h)
    qpav = (qp + 0 + 1) >> 1
    a, b, _ = thresholds(qpav, alpha_div2, beta_div2)
    y = deblock_plane(y, 16, False, a, b, None)
    out = [y.ravel()]
    for plane, off in ((u, cqo), (v, cqo - 1 if cqo > 0 else cqo)):
        qpav = (chroma_qp(qp, off) + chroma_qp(0, off) + 1) >> 1
        a, b, _ = thresholds(qpav, alpha_div2, beta_div2)
        out.append(deblock_plane(plane, 8, True, a, b, None).ravel())
    return np.concatenate(out)


# the inter side's QP is the slice's, the I_PCM side's is 0: the average reaches 26 at most; FilterOffsetA / B up to +12
PCM_SETTINGS = [(51, 0, 0, 0), (51, 6, 6, 0), (44, 3, 6, 0), (38, 6, 2, 0), (51, 4, 5, 6), (46, 6, 6, -4), (33, 6, 6, 0)]


def pcm_clip_args(qp, a, b, cqo):
    return dict(clip_args(qp, a, b, cqo), global_mv=3)


@pytest.mark.parametrize("qp,a,b,cqo", PCM_SETTINGS)
@pytest.mark.parametrize("cabac", [1, 0])
def test_oracle_filters_intra_edges_as_the_clauses_say(hw, oracle, tmp_path, qp, a, b, cqo, cabac):
    path = str(tmp_path / "dbk4.mp4")
    gold = make_clip(path, width=W, height=H, cabac=cabac, **pcm_clip_args(qp, a, b, cqo))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == FRAMES
    touched = untouched = 0
    for k in range(1, FRAMES):
        want = expected_pcm(frames[k - 1][0], k, W, H, qp, a, b, cqo)
        assert np.array_equal(frames[k][0], want), "picture %d" % k
        assert np.array_equal(frames[k][0], gold[k])
        before = mosaic_pcm(frames[k - 1][0], k, W, H)[0]
        after = want[:W * H].reshape(H, W)
        for x0 in range(16, W, 16):
            moved = (before[:, x0 - 1] != after[:, x0 - 1]) | (before[:, x0] != after[:, x0])
            touched += int(moved.sum())
            untouched += int((~moved).sum())
    assert touched > 0 and untouched > 0


@pytest.mark.gpu
@pytest.mark.parametrize("qp,a,b,cqo", PCM_SETTINGS)
def test_hip_path_filters_intra_edges_as_the_clauses_say(hw, tmp_path, qp, a, b, cqo):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = str(tmp_path / "dbk4.mp4")
    make_clip(path, width=W, height=H, yuv=False, **pcm_clip_args(qp, a, b, cqo))
    got = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(FRAMES)))
    for k in range(1, FRAMES):
        assert np.array_equal(got[k], expected_pcm(got[k - 1], k, W, H, qp, a, b, cqo)), "picture %d" % k


@pytest.mark.gpu
@pytest.mark.parametrize("qp,a,b,cqo", SETTINGS)
def test_hip_path_filters_known_edges_as_the_clauses_say(hw, tmp_path, qp, a, b, cqo):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = str(tmp_path / "dbk.mp4")
    make_clip(path, width=W, height=H, yuv=False, **clip_args(qp, a, b, cqo))
    got = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(FRAMES)))
    for k in range(1, FRAMES):
        assert np.array_equal(got[k], expected(got[k - 1], k, W, H, qp, a, b, cqo)), "picture %d" % k

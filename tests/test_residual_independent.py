"""Dequantisation and the 4x4 inverse transforms checked against a restatement that shares nothing with the oracle or the
kernels - the normAdjust table and the chroma QP table included: they are typed a second time here.

`--global-mv 4` (corpus/h264_encoder.c) writes P pictures whose macroblocks all predict with one vector from reference 0
(tests/test_inter_independent.py) and carry a residual whose coefficient LEVELS follow from the macroblock's address, the
block, the position and the picture's number (global_level) - sparse, small, now and then large - with the slice's QP in
every macroblock and the in-loop filter switched off (further down: switched on; `--t8x8 1`: the luma levels are those
of 8x8 blocks, 8.5.13).  So a picture must decode to

    Clip1(prediction + inverse transform(scaled levels))

with 8.5.12 (luma 4x4 blocks), 8.5.11 (chroma DC 2x2 and chroma AC) and the flat scaling of 8.5.9 (LevelScale4x4 =
16 * normAdjust4x4).  That is computed here with numpy for QPs on both sides of the scaling rule's case split (QP < 24 /
>= 24, chroma QPs up to 39) and compared with the oracle's decode (CPU) and the HIP path's (GPU: parser -> k_residual ->
k_inter's residual add)."""
import numpy as np
import pytest

from tests.corpus_util import make_clip
from tests.test_deblock_independent import ALPHA, BETA, chroma_qp, clip3, filter_line
from tests.test_inter_independent import global_mv_of, planes, shift_chroma, shift_luma

W, H = 64, 48
FRAMES = 8

# 8.5.9, Table of normAdjust4x4(m, i, j): v[m] = (positions with i, j both even; both odd; the others)
NORM_ADJUST = [(10, 16, 13), (11, 18, 14), (13, 20, 16), (14, 23, 18), (16, 25, 20), (18, 29, 23)]


# Scaling matrices (--scaling 4, corpus/h264_encoder.c: write_sps): entry j, in zig-zag order, of list i is
# 8 + (i * 7 + j * 3) % 25.  Lists 0-2: Intra Y / Cb / Cr 4x4, 3-5: Inter Y / Cb / Cr 4x4, 6 / 7: Intra / Inter Y 8x8.
ZIGZAG4 = [0, 1, 4, 8, 5, 2, 3, 6, 9, 12, 13, 10, 7, 11, 14, 15]   # raster position of zig-zag index (8.5.6, frames)
ZIGZAG8 = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49,
           56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]
SCALING = {"on": False}   # set by the tests that encode with --scaling 4


def weight_scale(lst, n):
    if not SCALING["on"]:
        return np.full((n, n), 16, dtype=np.int64)   # Flat_4x4_16 / Flat_8x8_16
    w = np.zeros(n * n, dtype=np.int64)
    for j, raster in enumerate(ZIGZAG4 if n == 4 else ZIGZAG8):
        w[raster] = 8 + (lst * 7 + j * 3) % 25
    return w.reshape(n, n)


def level_scale(m, lst=3):
    ls = np.zeros((4, 4), dtype=np.int64)
    ws = weight_scale(lst, 4)
    for i in range(4):
        for j in range(4):
            cls = 0 if (i % 2 == 0 and j % 2 == 0) else (1 if (i % 2 == 1 and j % 2 == 1) else 2)
            ls[i, j] = ws[i, j] * NORM_ADJUST[m][cls]
    return ls


def global_level(k, addr, comp, blk, pos):
    """corpus/h264_encoder.c: global_level."""
    t = addr * 7 + blk * 5 + pos * 3 + k * 11 + comp * 13
    if (comp <= 2 or comp == 5) and (addr + blk + k) % 3 == 0:
        return 0   # a third of the blocks carry nothing (their edges: bS 0)
    if t % 5 != 0:
        return 0
    v = (addr + blk * 3 + pos + comp) % 9 - 4
    return v * 9 if t % 35 == 0 else v


def scale_ac(c, qp, lst=3):
    """8.5.12.1: d_ij for a 4x4 block of levels c (all positions; the caller overrides d_00 where a DC path owns it)."""
    ls = level_scale(qp % 6, lst)
    if qp >= 24:
        return (c * ls) << (qp // 6 - 4)
    return (c * ls + (1 << (3 - qp // 6))) >> (4 - qp // 6)


def inverse4x4(d):
    """8.5.12.2: rows, then columns, then (x + 32) >> 6."""
    def butterfly(v):   # v: four values along the transformed axis (arrays over the other axis)
        e0, e1 = v[0] + v[2], v[0] - v[2]
        e2, e3 = (v[1] >> 1) - v[3], v[1] + (v[3] >> 1)
        return [e0 + e3, e1 + e2, e1 - e2, e0 - e3]
    f = np.stack(butterfly([d[:, 0], d[:, 1], d[:, 2], d[:, 3]]), axis=1)        # each row
    h = np.stack(butterfly([f[0, :], f[1, :], f[2, :], f[3, :]]), axis=0)        # each column
    return (h + 32) >> 6


def residual_luma(k, addr, qp):
    r = np.zeros((16, 16), dtype=np.int64)
    for blk in range(16):
        c = np.array([[global_level(k, addr, 0, blk, i * 4 + j) for j in range(4)] for i in range(4)], dtype=np.int64)
        by, bx = (blk // 4) * 4, (blk % 4) * 4
        r[by:by + 4, bx:bx + 4] = inverse4x4(scale_ac(c, qp, 3))   # (an inter macroblock: list 3)
    return r


# 8.5.9, normAdjust8x8(m, i, j): six classes of positions
NORM_ADJUST8 = [(20, 18, 32, 19, 25, 24), (22, 19, 35, 21, 28, 26), (26, 23, 42, 24, 33, 31), (28, 25, 45, 26, 35, 33),
                (32, 28, 51, 30, 40, 38), (36, 32, 58, 34, 46, 43)]


def level_scale8(m, lst=7):
    ls = np.zeros((8, 8), dtype=np.int64)
    ws = weight_scale(lst, 8)
    for i in range(8):
        for j in range(8):
            if i % 4 == 0 and j % 4 == 0:
                cls = 0
            elif i % 2 == 1 and j % 2 == 1:
                cls = 1
            elif i % 4 == 2 and j % 4 == 2:
                cls = 2
            elif (i % 4 == 0 and j % 2 == 1) or (i % 2 == 1 and j % 4 == 0):
                cls = 3
            elif (i % 4 == 0 and j % 4 == 2) or (i % 4 == 2 and j % 4 == 0):
                cls = 4
            else:
                cls = 5
            ls[i, j] = ws[i, j] * NORM_ADJUST8[m][cls]
    return ls


def inverse8x8(c, qp, lst=7):
    """8.5.13: scaling, the one-dimensional transform over every row, then over every column, (x + 32) >> 6."""
    ls = level_scale8(qp % 6, lst)
    d = (c * ls) << (qp // 6 - 6) if qp >= 36 else (c * ls + (1 << (5 - qp // 6))) >> (6 - qp // 6)

    def one_d(v):
        a0, a4 = v[0] + v[4], v[0] - v[4]
        a2, a6 = (v[2] >> 1) - v[6], v[2] + (v[6] >> 1)
        a1 = -v[3] + v[5] - v[7] - (v[7] >> 1)
        a3 = v[1] + v[7] - v[3] - (v[3] >> 1)
        a5 = -v[1] + v[7] + v[5] + (v[5] >> 1)
        a7 = v[3] + v[5] + v[1] + (v[1] >> 1)
        b0, b2, b4, b6 = a0 + a6, a4 + a2, a4 - a2, a0 - a6
        b1, b3, b5, b7 = a1 + (a7 >> 2), a3 + (a5 >> 2), (a3 >> 2) - a5, a7 - (a1 >> 2)
        return [b0 + b7, b2 + b5, b4 + b3, b6 + b1, b6 - b1, b4 - b3, b2 - b5, b0 - b7]
    f = np.stack(one_d([d[:, n] for n in range(8)]), axis=1)
    g = np.stack(one_d([f[n, :] for n in range(8)]), axis=0)
    return (g + 32) >> 6


def residual_luma8(k, addr, qp):
    r = np.zeros((16, 16), dtype=np.int64)
    for b8 in range(4):
        c = np.array([[global_level(k, addr, 5, b8, i * 8 + j) for j in range(8)] for i in range(8)], dtype=np.int64)
        by, bx = (b8 // 2) * 8, (b8 % 2) * 8
        r[by:by + 8, bx:bx + 8] = inverse8x8(c, qp)
    return r


def residual_chroma(k, addr, pl, qpc, intra=False):
    """8.5.11: the four DC levels through the 2x2 transform and their own scaling, then each block's AC with that DC."""
    lst = (1 if intra else 4) + pl
    c = np.array([[global_level(k, addr, 3 + pl, 0, 0), global_level(k, addr, 3 + pl, 0, 1)],
                  [global_level(k, addr, 3 + pl, 0, 2), global_level(k, addr, 3 + pl, 0, 3)]], dtype=np.int64)
    a = np.array([[1, 1], [1, -1]], dtype=np.int64)
    f = a @ c @ a
    dc = ((f * level_scale(qpc % 6, lst)[0, 0]) << (qpc // 6)) >> 5
    r = np.zeros((8, 8), dtype=np.int64)
    for blk in range(4):
        lv = np.array([[0 if (i == 0 and j == 0) else global_level(k, addr, 1 + pl, blk, i * 4 + j) for j in range(4)] for i in range(4)],
                      dtype=np.int64)
        d = scale_ac(lv, qpc, lst)
        d[0, 0] = dc[blk // 2, blk % 2]
        by, bx = (blk // 2) * 4, (blk % 2) * 4
        r[by:by + 4, bx:bx + 4] = inverse4x4(d)
    return r


def expected(prev, k, w, h, qp, cqo, t8x8=False):
    mv = global_mv_of(k, w, h)
    y, u, v = planes(prev, w, h)
    py, pu, pv = (shift_luma(y, *mv).astype(np.int64), shift_chroma(u, *mv).astype(np.int64), shift_chroma(v, *mv).astype(np.int64))
    mbw = w // 16
    # Cr: without the PPS extension (--t8x8 0) it follows Cb; with it the corpus encoder writes one less when positive
    qpcs = (chroma_qp(qp, cqo), chroma_qp(qp, cqo - 1 if (t8x8 and cqo > 0) else cqo))
    for addr in range(mbw * (h // 16)):
        mx, my = addr % mbw, addr // mbw
        py[my * 16:my * 16 + 16, mx * 16:mx * 16 + 16] += residual_luma8(k, addr, qp) if t8x8 else residual_luma(k, addr, qp)
        pu[my * 8:my * 8 + 8, mx * 8:mx * 8 + 8] += residual_chroma(k, addr, 0, qpcs[0])
        pv[my * 8:my * 8 + 8, mx * 8:mx * 8 + 8] += residual_chroma(k, addr, 1, qpcs[1])
    return np.concatenate([np.clip(p, 0, 255).astype(np.uint8).ravel() for p in (py, pu, pv)])


# (QP, chroma_qp_index_offset): every QP % 6, both sides of QP 24, chroma QPs in the table's tail
SETTINGS = [(12, 0), (19, 0), (23, 0), (24, 0), (26, 0), (27, 0), (28, 0), (29, 0), (31, 0), (36, 0), (45, 0), (51, 0), (30, 6), (40, -6), (17, 5)]


def clip_args(qp, cqo):
    return dict(frames=FRAMES, gop=FRAMES, bframes=0, refs=1, global_mv=4, qp=qp, chroma_qp_offset=cqo, deblock=1, t8x8=0)


@pytest.mark.parametrize("qp,cqo", SETTINGS)
@pytest.mark.parametrize("cabac", [1, 0])
def test_oracle_adds_known_residuals_as_the_clauses_say(hw, oracle, tmp_path, qp, cqo, cabac):
    path = str(tmp_path / "res.mp4")
    gold = make_clip(path, width=W, height=H, cabac=cabac, **clip_args(qp, cqo))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == FRAMES
    moved = 0
    for k in range(1, FRAMES):
        want = expected(frames[k - 1][0], k, W, H, qp, cqo)
        assert np.array_equal(frames[k][0], want), "picture %d" % k
        assert np.array_equal(frames[k][0], gold[k])
        mv = global_mv_of(k, W, H)
        moved += int((want[:W * H].reshape(H, W) != shift_luma(planes(frames[k - 1][0], W, H)[0], *mv)).sum())
    assert moved > 0   # (the residual does something)


T8_SETTINGS = [(12, 0), (19, 0), (26, 0), (27, 0), (28, 0), (29, 0), (35, 0), (36, 0), (37, 0), (44, 3), (51, 0)]   # both sides of QP 36


@pytest.mark.parametrize("qp,cqo", T8_SETTINGS)
@pytest.mark.parametrize("cabac", [1, 0])
def test_oracle_adds_known_8x8_residuals_as_the_clauses_say(hw, oracle, tmp_path, qp, cqo, cabac):
    path = str(tmp_path / "res8.mp4")
    gold = make_clip(path, width=W, height=H, cabac=cabac, **dict(clip_args(qp, cqo), t8x8=1))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == FRAMES
    for k in range(1, FRAMES):
        assert np.array_equal(frames[k][0], expected(frames[k - 1][0], k, W, H, qp, cqo, True)), "picture %d" % k
        assert np.array_equal(frames[k][0], gold[k])


@pytest.mark.gpu
@pytest.mark.parametrize("qp,cqo", T8_SETTINGS)
def test_hip_path_adds_known_8x8_residuals_as_the_clauses_say(hw, tmp_path, qp, cqo):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = str(tmp_path / "res8.mp4")
    make_clip(path, width=W, height=H, yuv=False, **dict(clip_args(qp, cqo), t8x8=1))
    got = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(FRAMES)))
    for k in range(1, FRAMES):
        assert np.array_equal(got[k], expected(got[k - 1], k, W, H, qp, cqo, True)), "picture %d" % k


@pytest.mark.gpu
@pytest.mark.parametrize("qp,cqo", SETTINGS)
def test_hip_path_adds_known_residuals_as_the_clauses_say(hw, tmp_path, qp, cqo):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = str(tmp_path / "res.mp4")
    make_clip(path, width=W, height=H, yuv=False, **clip_args(qp, cqo))
    got = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(FRAMES)))
    for k in range(1, FRAMES):
        assert np.array_equal(got[k], expected(got[k - 1], k, W, H, qp, cqo)), "picture %d" % k


# ---- the same pictures with the in-loop filter on: boundary strength 2 ---------------------------------------------------
# A 4x4 luma block with a non-zero level gives every edge it touches bS = 2 (8.7.2.1); all vectors and references are the
# same, so every other edge has bS = 0.  Chroma edges take the strengths of the luma edge they lie on.
TC0_BS2 = [0] * 21 + [1] * 10 + [2, 2, 2, 2, 3, 3, 3, 4, 4, 5, 5, 6, 7, 8, 8, 10, 11, 12, 13, 15, 17]   # Table 8-17, bS = 2
assert len(TC0_BS2) == 52


def coded(k, addr, blk):
    return any(global_level(k, addr, 0, blk, pos) for pos in range(16))


def filter_picture_bs2(y, u, v, k, w, h, qp, qpcs, alpha_div2, beta_div2):
    mbw, mbh = w // 16, h // 16
    nz = np.zeros((mbh * 4, mbw * 4), dtype=bool)   # per 4x4 luma block of the picture
    for addr in range(mbw * mbh):
        mx, my = addr % mbw, addr // mbw
        for blk in range(16):
            nz[my * 4 + blk // 4, mx * 4 + blk % 4] = coded(k, addr, blk)

    def thr(q):
        ia, ib = clip3(0, 51, q + 2 * alpha_div2), clip3(0, 51, q + 2 * beta_div2)
        return ALPHA[ia], BETA[ib], TC0_BS2[ia]
    pl = [p.astype(np.int64) for p in (y, u, v)]
    th = [thr(qp), thr(qpcs[0]), thr(qpcs[1])]
    for my in range(mbh):
        for mx in range(mbw):
            for vertical in (True, False):   # the macroblock's vertical edges left to right, then its horizontal ones
                for e in range(4):
                    if e == 0 and (mx == 0 if vertical else my == 0):
                        continue   # the picture's border
                    for seg in range(4):   # four lines of luma each
                        if vertical:
                            by, bx = my * 4 + seg, mx * 4 + e
                            strong = nz[by, bx] or nz[by, bx - 1]
                        else:
                            by, bx = my * 4 + e, mx * 4 + seg
                            strong = nz[by, bx] or nz[by - 1, bx]
                        if not strong:
                            continue
                        for p_i, (a, chroma) in enumerate(((pl[0], False), (pl[1], True), (pl[2], True))):
                            if chroma and e % 2:
                                continue   # chroma has edges where luma has its edges 0 and 2
                            alpha, beta, tc0 = th[p_i]
                            n, reach = (2, 2) if chroma else (4, 4)
                            for i in range(n):
                                if vertical:
                                    x0 = (mx * 16 + e * 4) // (2 if chroma else 1)
                                    yy = (my * 16 + seg * 4) // (2 if chroma else 1) + i
                                    px = [0] * 8
                                    px[4 - reach:4 + reach] = [int(t) for t in a[yy, x0 - reach:x0 + reach]]
                                    filter_line(px, chroma, alpha, beta, tc0)
                                    a[yy, x0 - reach:x0 + reach] = px[4 - reach:4 + reach]
                                else:
                                    y0 = (my * 16 + e * 4) // (2 if chroma else 1)
                                    xx = (mx * 16 + seg * 4) // (2 if chroma else 1) + i
                                    px = [0] * 8
                                    px[4 - reach:4 + reach] = [int(t) for t in a[y0 - reach:y0 + reach, xx]]
                                    filter_line(px, chroma, alpha, beta, tc0)
                                    a[y0 - reach:y0 + reach, xx] = px[4 - reach:4 + reach]
    return np.concatenate([np.clip(p, 0, 255).astype(np.uint8).ravel() for p in pl])


def expected_filtered(prev, k, w, h, qp, cqo, alpha_div2, beta_div2):
    unfiltered = expected(prev, k, w, h, qp, cqo)
    y, u, v = planes(unfiltered, w, h)
    return filter_picture_bs2(y, u, v, k, w, h, qp, (chroma_qp(qp, cqo), chroma_qp(qp, cqo)), alpha_div2, beta_div2)


# (QP, chroma_qp_index_offset, FilterOffsetA / 2, FilterOffsetB / 2)
BS2_SETTINGS = [(26, 0, 0, 0), (32, 0, 0, 0), (38, 0, 0, 0), (45, 0, 0, 0), (51, 0, 0, 0), (28, 0, 3, -2), (36, 4, -1, 2), (41, -3, 2, 2), (22, 0, 6, 6)]


def bs2_clip_args(qp, cqo, a, b):
    return dict(clip_args(qp, cqo), deblock=0, alpha=a, beta=b)


@pytest.mark.parametrize("qp,cqo,a,b", BS2_SETTINGS)
@pytest.mark.parametrize("cabac", [1, 0])
def test_oracle_filters_coded_blocks_edges_as_the_clauses_say(hw, oracle, tmp_path, qp, cqo, a, b, cabac):
    path = str(tmp_path / "res2.mp4")
    gold = make_clip(path, width=W, height=H, cabac=cabac, **bs2_clip_args(qp, cqo, a, b))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == FRAMES
    moved = 0
    for k in range(1, FRAMES):
        want = expected_filtered(frames[k - 1][0], k, W, H, qp, cqo, a, b)
        assert np.array_equal(frames[k][0], want), "picture %d" % k
        assert np.array_equal(frames[k][0], gold[k])
        moved += int((want != expected(frames[k - 1][0], k, W, H, qp, cqo)).sum())
    assert moved > 0   # (the filter does something)


def test_every_row_of_the_bs2_column(hw, oracle, tmp_path):
    """QP 17 .. 51 without offsets: indexA = QP for luma: every non-zero row of the bS = 2 column of Table 8-17."""
    for qp in range(17, 52):
        path = str(tmp_path / ("res2_qp%d.mp4" % qp))
        make_clip(path, width=W, height=H, yuv=False, **dict(bs2_clip_args(qp, 0, 0, 0), frames=3, gop=3))
        frames, _ = oracle.decode_file(path, hw.index_video(path))
        for k in range(1, 3):
            assert np.array_equal(frames[k][0], expected_filtered(frames[k - 1][0], k, W, H, qp, 0, 0, 0)), "QP %d, picture %d" % (qp, k)


@pytest.mark.gpu
@pytest.mark.parametrize("qp,cqo,a,b", BS2_SETTINGS)
def test_hip_path_filters_coded_blocks_edges_as_the_clauses_say(hw, tmp_path, qp, cqo, a, b):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = str(tmp_path / "res2.mp4")
    make_clip(path, width=W, height=H, yuv=False, **bs2_clip_args(qp, cqo, a, b))
    got = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(FRAMES)))
    for k in range(1, FRAMES):
        assert np.array_equal(got[k], expected_filtered(got[k - 1], k, W, H, qp, cqo, a, b)), "picture %d" % k


# ---- Intra16x16 macroblocks with known levels among the inter ones (--global-mv 5) ----------------------------------------
# Odd macroblocks of the checkerboard: Intra16x16 with prediction modes that follow from address and picture number, the
# luma DC levels of comp 6 (8.5.10: 4x4 Hadamard + its own scaling), AC levels of comp 0, chroma as before.  Their
# neighbours are inter macroblocks (and, diagonally, intra ones) of the same picture: the picture is rebuilt macroblock
# by macroblock in decoding order.  Filter off.
def pred16(img, x0, y0, mode):
    """8.3.3: Intra16x16 prediction at (x0, y0) of the luma plane under reconstruction."""
    top = img[y0 - 1, x0:x0 + 16] if y0 > 0 else None
    left = img[y0:y0 + 16, x0 - 1] if x0 > 0 else None
    if mode == 0:
        return np.tile(top, (16, 1))
    if mode == 1:
        return np.tile(left.reshape(16, 1), (1, 16))
    if mode == 2:
        if top is not None and left is not None:
            dc = (int(top.sum()) + int(left.sum()) + 16) >> 5
        elif top is not None or left is not None:
            dc = (int((top if top is not None else left).sum()) + 8) >> 4
        else:
            dc = 128
        return np.full((16, 16), dc, dtype=np.int64)
    p = lambda x, y: int(img[y0 + y, x0 + x])   # noqa: E731  (x or y may be -1)
    hh = sum((i + 1) * (p(8 + i, -1) - p(6 - i, -1)) for i in range(8))
    vv = sum((i + 1) * (p(-1, 8 + i) - p(-1, 6 - i)) for i in range(8))
    a, b, c = 16 * (p(-1, 15) + p(15, -1)), (5 * hh + 32) >> 6, (5 * vv + 32) >> 6
    return np.array([[clip3(0, 255, (a + b * (x - 7) + c * (y - 7) + 16) >> 5) for x in range(16)] for y in range(16)], dtype=np.int64)


def pred_chroma(img, x0, y0, mode):
    """8.3.4: chroma prediction (4:2:0) at (x0, y0) of a chroma plane under reconstruction."""
    top = img[y0 - 1, x0:x0 + 8] if y0 > 0 else None
    left = img[y0:y0 + 8, x0 - 1] if x0 > 0 else None
    if mode == 2:
        return np.tile(top, (8, 1))
    if mode == 1:
        return np.tile(left.reshape(8, 1), (1, 8))
    if mode == 0:
        out = np.zeros((8, 8), dtype=np.int64)
        for by in range(2):
            for bx in range(2):
                t = int(top[bx * 4:bx * 4 + 4].sum()) if top is not None else None
                lf = int(left[by * 4:by * 4 + 4].sum()) if left is not None else None
                if (bx, by) in ((0, 0), (1, 1)):
                    dc = (t + lf + 4) >> 3 if (t is not None and lf is not None) else ((t + 2) >> 2 if t is not None else ((lf + 2) >> 2 if lf is not None else 128))
                elif (bx, by) == (1, 0):   # prefers the samples above
                    dc = (t + 2) >> 2 if t is not None else ((lf + 2) >> 2 if lf is not None else 128)
                else:                      # (0, 1): prefers the samples to the left
                    dc = (lf + 2) >> 2 if lf is not None else ((t + 2) >> 2 if t is not None else 128)
                out[by * 4:by * 4 + 4, bx * 4:bx * 4 + 4] = dc
        return out
    p = lambda x, y: int(img[y0 + y, x0 + x])   # noqa: E731
    hh = sum((i + 1) * (p(4 + i, -1) - p(2 - i, -1)) for i in range(4))
    vv = sum((i + 1) * (p(-1, 4 + i) - p(-1, 2 - i)) for i in range(4))
    a, b, c = 16 * (p(-1, 7) + p(7, -1)), (34 * hh + 32) >> 6, (34 * vv + 32) >> 6
    return np.array([[clip3(0, 255, (a + b * (x - 3) + c * (y - 3) + 16) >> 5) for x in range(8)] for y in range(8)], dtype=np.int64)


def residual_luma16(k, addr, qp):
    """8.5.10 + 8.5.12: the DC levels through the 4x4 Hadamard transform and their scaling, then every block's AC."""
    c = np.array([[global_level(k, addr, 6, 0, i * 4 + j) for j in range(4)] for i in range(4)], dtype=np.int64)
    hm = np.array([[1, 1, 1, 1], [1, 1, -1, -1], [1, -1, -1, 1], [1, -1, 1, -1]], dtype=np.int64)
    f = hm @ c @ hm
    ls = level_scale(qp % 6, 0)[0, 0]   # (Intra Y: list 0)
    dc = (f * ls) << (qp // 6 - 6) if qp >= 36 else (f * ls + (1 << (5 - qp // 6))) >> (6 - qp // 6)
    r = np.zeros((16, 16), dtype=np.int64)
    for blk in range(16):
        lv = np.array([[0 if (i == 0 and j == 0) else global_level(k, addr, 0, blk, i * 4 + j) for j in range(4)] for i in range(4)],
                      dtype=np.int64)
        d = scale_ac(lv, qp, 0)
        d[0, 0] = dc[blk // 4, blk % 4]
        by, bx = (blk // 4) * 4, (blk % 4) * 4
        r[by:by + 4, bx:bx + 4] = inverse4x4(d)
    return r


def expected_intra16(prev, k, w, h, qp):
    mv = global_mv_of(k, w, h)
    y, u, v = planes(prev, w, h)
    sy, su, sv = shift_luma(y, *mv).astype(np.int64), shift_chroma(u, *mv).astype(np.int64), shift_chroma(v, *mv).astype(np.int64)
    oy, ou, ov = np.zeros_like(sy), np.zeros_like(su), np.zeros_like(sv)
    mbw, qpc = w // 16, chroma_qp(qp, 0)
    for addr in range(mbw * (h // 16)):
        mx, my = addr % mbw, addr // mbw
        ly, lx, cy, cx = my * 16, mx * 16, my * 8, mx * 8
        if (mx + my) & 1:
            left, top = mx > 0, my > 0
            mode, cmode = (addr + k) % 4, (addr * 3 + k) % 4
            if (mode == 0 and not top) or (mode == 1 and not left) or (mode == 3 and not (left and top)):
                mode = 2
            if (cmode == 1 and not left) or (cmode == 2 and not top) or (cmode == 3 and not (left and top)):
                cmode = 0
            py = pred16(oy, lx, ly, mode) + residual_luma16(k, addr, qp)
            pu = pred_chroma(ou, cx, cy, cmode) + residual_chroma(k, addr, 0, qpc, True)
            pv = pred_chroma(ov, cx, cy, cmode) + residual_chroma(k, addr, 1, qpc, True)
        else:
            py = sy[ly:ly + 16, lx:lx + 16] + residual_luma(k, addr, qp)
            pu = su[cy:cy + 8, cx:cx + 8] + residual_chroma(k, addr, 0, qpc)
            pv = sv[cy:cy + 8, cx:cx + 8] + residual_chroma(k, addr, 1, qpc)
        oy[ly:ly + 16, lx:lx + 16] = np.clip(py, 0, 255)
        ou[cy:cy + 8, cx:cx + 8] = np.clip(pu, 0, 255)
        ov[cy:cy + 8, cx:cx + 8] = np.clip(pv, 0, 255)
    return np.concatenate([p.astype(np.uint8).ravel() for p in (oy, ou, ov)])


I16_QPS = [5, 10, 14, 22, 26, 29, 33, 35, 36, 37, 42, 51]   # both sides of QP 36 (the DC scaling case split); below 12 its rounding term counts
I16_SIZE = (112, 80)                                  # 7 x 5 macroblocks: every mode meets every kind of neighbourhood


@pytest.mark.parametrize("qp", I16_QPS)
@pytest.mark.parametrize("cabac", [1, 0])
def test_oracle_rebuilds_known_intra16_macroblocks_as_the_clauses_say(hw, oracle, tmp_path, qp, cabac):
    w, h = I16_SIZE
    path = str(tmp_path / "i16.mp4")
    gold = make_clip(path, width=w, height=h, cabac=cabac, **dict(clip_args(qp, 0), global_mv=5))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == FRAMES
    for k in range(1, FRAMES):
        assert np.array_equal(frames[k][0], expected_intra16(frames[k - 1][0], k, w, h, qp)), "picture %d" % k
        assert np.array_equal(frames[k][0], gold[k])


def test_every_intra16_and_chroma_mode_occurs():
    w, h = I16_SIZE
    mbw, seen16, seenc = w // 16, set(), set()
    for k in range(1, FRAMES):
        for addr in range(mbw * (h // 16)):
            mx, my = addr % mbw, addr // mbw
            if (mx + my) & 1 and mx > 0 and my > 0:
                seen16.add((addr + k) % 4)
                seenc.add((addr * 3 + k) % 4)
    assert seen16 == {0, 1, 2, 3} and seenc == {0, 1, 2, 3}


@pytest.mark.gpu
@pytest.mark.parametrize("qp", I16_QPS)
def test_hip_path_rebuilds_known_intra16_macroblocks_as_the_clauses_say(hw, tmp_path, qp):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    w, h = I16_SIZE
    path = str(tmp_path / "i16.mp4")
    make_clip(path, width=w, height=h, yuv=False, **dict(clip_args(qp, 0), global_mv=5))
    got = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(FRAMES)))
    for k in range(1, FRAMES):
        assert np.array_equal(got[k], expected_intra16(got[k - 1], k, w, h, qp)), "picture %d" % k


# ---- the same mixed pictures with the in-loop filter on: all boundary strengths in one picture ----------------------------
# Macroblock edges next to an Intra16x16 macroblock: bS 4; that macroblock's inner edges: bS 3; an inter macroblock's
# edges: 2 where a 4x4 block on either side carries a level, else 0.  Every macroblock has the slice's QP.
TC0_BS3 = [0] * 17 + [1] * 10 + [2] * 4 + [3, 3, 3, 4, 4, 4, 5, 6, 6, 7, 8, 9, 10, 11, 13, 14, 16, 18, 20, 23, 25]   # Table 8-17, bS = 3
assert len(TC0_BS3) == 52


def filter_picture_mixed(y, u, v, k, w, h, qp, qpc, alpha_div2, beta_div2):
    mbw, mbh = w // 16, h // 16
    nz = np.zeros((mbh * 4, mbw * 4), dtype=bool)
    intra = np.zeros((mbh, mbw), dtype=bool)
    for addr in range(mbw * mbh):
        mx, my = addr % mbw, addr // mbw
        intra[my, mx] = (mx + my) & 1
        for blk in range(16):
            nz[my * 4 + blk // 4, mx * 4 + blk % 4] = coded(k, addr, blk)

    def thr(q, bs):
        ia, ib = clip3(0, 51, q + 2 * alpha_div2), clip3(0, 51, q + 2 * beta_div2)
        return ALPHA[ia], BETA[ib], (None if bs == 4 else (TC0_BS2 if bs == 2 else TC0_BS3)[ia])
    pl = [p.astype(np.int64) for p in (y, u, v)]
    for my in range(mbh):
        for mx in range(mbw):
            for vertical in (True, False):
                for e in range(4):
                    if e == 0 and (mx == 0 if vertical else my == 0):
                        continue
                    for seg in range(4):
                        if vertical:
                            by, bx = my * 4 + seg, mx * 4 + e
                            pby, pbx, pmy, pmx = by, bx - 1, my, (mx - 1 if e == 0 else mx)
                        else:
                            by, bx = my * 4 + e, mx * 4 + seg
                            pby, pbx, pmy, pmx = by - 1, bx, (my - 1 if e == 0 else my), mx
                        if intra[my, mx] or intra[pmy, pmx]:
                            bs = 4 if e == 0 else 3
                        else:
                            bs = 2 if (nz[by, bx] or nz[pby, pbx]) else 0
                        if bs == 0:
                            continue
                        for p_i, (a, chroma) in enumerate(((pl[0], False), (pl[1], True), (pl[2], True))):
                            if chroma and e % 2:
                                continue
                            alpha, beta, tc0 = thr(qpc if chroma else qp, bs)
                            n, reach = (2, 2) if chroma else (4, 4)
                            for i in range(n):
                                if vertical:
                                    x0 = (mx * 16 + e * 4) // (2 if chroma else 1)
                                    yy = (my * 16 + seg * 4) // (2 if chroma else 1) + i
                                    px = [0] * 8
                                    px[4 - reach:4 + reach] = [int(t) for t in a[yy, x0 - reach:x0 + reach]]
                                    filter_line(px, chroma, alpha, beta, tc0)
                                    a[yy, x0 - reach:x0 + reach] = px[4 - reach:4 + reach]
                                else:
                                    y0 = (my * 16 + e * 4) // (2 if chroma else 1)
                                    xx = (mx * 16 + seg * 4) // (2 if chroma else 1) + i
                                    px = [0] * 8
                                    px[4 - reach:4 + reach] = [int(t) for t in a[y0 - reach:y0 + reach, xx]]
                                    filter_line(px, chroma, alpha, beta, tc0)
                                    a[y0 - reach:y0 + reach, xx] = px[4 - reach:4 + reach]
    return np.concatenate([np.clip(p, 0, 255).astype(np.uint8).ravel() for p in pl])


def expected_mixed_filtered(prev, k, w, h, qp, alpha_div2, beta_div2):
    y, u, v = planes(expected_intra16(prev, k, w, h, qp), w, h)
    return filter_picture_mixed(y, u, v, k, w, h, qp, chroma_qp(qp, 0), alpha_div2, beta_div2)


# (QP, FilterOffsetA / 2, FilterOffsetB / 2)
MIXED_SETTINGS = [(22, 0, 0), (28, 0, 0), (34, 0, 0), (40, 0, 0), (46, 0, 0), (51, 0, 0), (30, 3, -2), (37, -2, 3), (25, 6, 6)]


def mixed_clip_args(qp, a, b):
    return dict(clip_args(qp, 0), global_mv=5, deblock=0, alpha=a, beta=b)


@pytest.mark.parametrize("qp,a,b", MIXED_SETTINGS)
@pytest.mark.parametrize("cabac", [1, 0])
def test_oracle_filters_mixed_pictures_as_the_clauses_say(hw, oracle, tmp_path, qp, a, b, cabac):
    w, h = I16_SIZE
    path = str(tmp_path / "mix.mp4")
    gold = make_clip(path, width=w, height=h, cabac=cabac, **mixed_clip_args(qp, a, b))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == FRAMES
    for k in range(1, FRAMES):
        assert np.array_equal(frames[k][0], expected_mixed_filtered(frames[k - 1][0], k, w, h, qp, a, b)), "picture %d" % k
        assert np.array_equal(frames[k][0], gold[k])


def test_every_row_of_the_bs3_column(hw, oracle, tmp_path):
    """QP 17 .. 51 without offsets: every non-zero row of the bS = 3 column of Table 8-17 (an Intra16x16 macroblock's inner edges)."""
    w, h = I16_SIZE
    for qp in range(17, 52):
        path = str(tmp_path / ("mix_qp%d.mp4" % qp))
        make_clip(path, width=w, height=h, yuv=False, **dict(mixed_clip_args(qp, 0, 0), frames=3, gop=3))
        frames, _ = oracle.decode_file(path, hw.index_video(path))
        for k in range(1, 3):
            assert np.array_equal(frames[k][0], expected_mixed_filtered(frames[k - 1][0], k, w, h, qp, 0, 0)), "QP %d, picture %d" % (qp, k)


@pytest.mark.gpu
@pytest.mark.parametrize("qp,a,b", MIXED_SETTINGS)
def test_hip_path_filters_mixed_pictures_as_the_clauses_say(hw, tmp_path, qp, a, b):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    w, h = I16_SIZE
    path = str(tmp_path / "mix.mp4")
    make_clip(path, width=w, height=h, yuv=False, **mixed_clip_args(qp, a, b))
    got = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(FRAMES)))
    for k in range(1, FRAMES):
        assert np.array_equal(got[k], expected_mixed_filtered(got[k - 1], k, w, h, qp, a, b)), "picture %d" % k


# ---- Intra4x4 macroblocks with known modes and levels (--global-mv 6) -----------------------------------------------------
def pred4(img, x0, y0, mode, top_right_ok, w):
    """8.3.1.2: the nine Intra4x4 prediction modes for the block at (x0, y0) of the luma plane under reconstruction.
    Inside the picture (x0, y0 > 0) left, top and top-left samples exist; the top-right ones (E..H) are replaced by D
    when that block has not been decoded yet or lies outside the picture."""
    p = {}
    for i in range(4):
        p[(i, -1)] = int(img[y0 - 1, x0 + i]) if y0 > 0 else None
        p[(-1, i)] = int(img[y0 + i, x0 - 1]) if x0 > 0 else None
    for i in range(4):
        p[(4 + i, -1)] = (int(img[y0 - 1, x0 + 4 + i]) if top_right_ok else p[(3, -1)]) if y0 > 0 else None
    p[(-1, -1)] = int(img[y0 - 1, x0 - 1]) if (x0 > 0 and y0 > 0) else None
    out = np.zeros((4, 4), dtype=np.int64)
    if mode == 2:
        t = [p[(i, -1)] for i in range(4)] if y0 > 0 else None
        lf = [p[(-1, i)] for i in range(4)] if x0 > 0 else None
        if t and lf:
            out[:] = (sum(t) + sum(lf) + 4) >> 3
        elif t or lf:
            out[:] = (sum(t or lf) + 2) >> 2
        else:
            out[:] = 128
        return out
    for y in range(4):
        for x in range(4):
            if mode == 0:
                v = p[(x, -1)]
            elif mode == 1:
                v = p[(-1, y)]
            elif mode == 3:   # diagonal down-left
                v = (p[(6, -1)] + 3 * p[(7, -1)] + 2) >> 2 if (x == 3 and y == 3) else (p[(x + y, -1)] + 2 * p[(x + y + 1, -1)] + p[(x + y + 2, -1)] + 2) >> 2
            elif mode == 4:   # diagonal down-right
                if x > y:
                    v = (p[(x - y - 2, -1)] + 2 * p[(x - y - 1, -1)] + p[(x - y, -1)] + 2) >> 2
                elif x < y:
                    v = (p[(-1, y - x - 2)] + 2 * p[(-1, y - x - 1)] + p[(-1, y - x)] + 2) >> 2
                else:
                    v = (p[(0, -1)] + 2 * p[(-1, -1)] + p[(-1, 0)] + 2) >> 2
            elif mode == 5:   # vertical-right
                z, s = 2 * x - y, x - (y >> 1)
                if z >= 0 and z % 2 == 0:
                    v = (p[(s - 1, -1)] + p[(s, -1)] + 1) >> 1
                elif z >= 0:
                    v = (p[(s - 2, -1)] + 2 * p[(s - 1, -1)] + p[(s, -1)] + 2) >> 2
                elif z == -1:
                    v = (p[(-1, 0)] + 2 * p[(-1, -1)] + p[(0, -1)] + 2) >> 2
                else:
                    v = (p[(-1, y - 1)] + 2 * p[(-1, y - 2)] + p[(-1, y - 3)] + 2) >> 2
            elif mode == 6:   # horizontal-down
                z, s = 2 * y - x, y - (x >> 1)
                if z >= 0 and z % 2 == 0:
                    v = (p[(-1, s - 1)] + p[(-1, s)] + 1) >> 1
                elif z >= 0:
                    v = (p[(-1, s - 2)] + 2 * p[(-1, s - 1)] + p[(-1, s)] + 2) >> 2
                elif z == -1:
                    v = (p[(-1, 0)] + 2 * p[(-1, -1)] + p[(0, -1)] + 2) >> 2
                else:
                    v = (p[(x - 1, -1)] + 2 * p[(x - 2, -1)] + p[(x - 3, -1)] + 2) >> 2
            elif mode == 7:   # vertical-left
                s = x + (y >> 1)
                v = (p[(s, -1)] + p[(s + 1, -1)] + 1) >> 1 if y % 2 == 0 else (p[(s, -1)] + 2 * p[(s + 1, -1)] + p[(s + 2, -1)] + 2) >> 2
            else:             # 8: horizontal-up
                z, s = x + 2 * y, y + (x >> 1)
                if z > 5:
                    v = p[(-1, 3)]
                elif z == 5:
                    v = (p[(-1, 2)] + 3 * p[(-1, 3)] + 2) >> 2
                elif z % 2 == 0:
                    v = (p[(-1, s)] + p[(-1, s + 1)] + 1) >> 1
                else:
                    v = (p[(-1, s)] + 2 * p[(-1, s + 1)] + p[(-1, s + 2)] + 2) >> 2
            out[y, x] = v
    # (the named sample p[-1, -1] and the four-sample runs use x < 0 / y < 0 only where the mode allows it)
    return out


def expected_intra4(prev, k, w, h, qp):
    mv = global_mv_of(k, w, h)
    y, u, v = planes(prev, w, h)
    sy, su, sv = shift_luma(y, *mv).astype(np.int64), shift_chroma(u, *mv).astype(np.int64), shift_chroma(v, *mv).astype(np.int64)
    oy, ou, ov = np.zeros_like(sy), np.zeros_like(su), np.zeros_like(sv)
    mbw, mbh, qpc = w // 16, h // 16, chroma_qp(qp, 0)
    for addr in range(mbw * mbh):
        mx, my = addr % mbw, addr // mbw
        ly, lx, cy, cx = my * 16, mx * 16, my * 8, mx * 8
        if (mx + my) & 1:
            inside = mx > 0 and my > 0
            cmode = (addr * 3 + k) % 4
            if (cmode == 1 and mx == 0) or (cmode == 2 and my == 0) or (cmode == 3 and not inside):
                cmode = 0
            for idx in range(16):   # the 4x4 blocks in decoding order (6.4.3)
                bx, by = (idx & 1) + ((idx >> 2) & 1) * 2, ((idx >> 1) & 1) + ((idx >> 3) & 1) * 2
                blk = by * 4 + bx
                mode = (addr * 5 + blk * 7 + k) % 9 if inside else 2
                if by == 0:   # the block above-right lies in the macroblock above, or (bx = 3) in the one above-right
                    tr_ok = my > 0 and (bx < 3 or mx + 1 < mbw)
                else:         # inside this macroblock: decoded already only for these positions; bx = 3: to the right, never
                    tr_ok = (bx, by) in ((0, 1), (0, 2), (0, 3), (2, 1), (2, 2), (2, 3), (1, 2))
                c = np.array([[global_level(k, addr, 0, blk, i * 4 + j) for j in range(4)] for i in range(4)], dtype=np.int64)
                x0, y0 = lx + bx * 4, ly + by * 4
                oy[y0:y0 + 4, x0:x0 + 4] = np.clip(pred4(oy, x0, y0, mode, tr_ok, w) + inverse4x4(scale_ac(c, qp, 0)), 0, 255)
            pu = pred_chroma(ou, cx, cy, cmode) + residual_chroma(k, addr, 0, qpc, True)
            pv = pred_chroma(ov, cx, cy, cmode) + residual_chroma(k, addr, 1, qpc, True)
        else:
            oy[ly:ly + 16, lx:lx + 16] = np.clip(sy[ly:ly + 16, lx:lx + 16] + residual_luma(k, addr, qp), 0, 255)
            pu = su[cy:cy + 8, cx:cx + 8] + residual_chroma(k, addr, 0, qpc)
            pv = sv[cy:cy + 8, cx:cx + 8] + residual_chroma(k, addr, 1, qpc)
        ou[cy:cy + 8, cx:cx + 8] = np.clip(pu, 0, 255)
        ov[cy:cy + 8, cx:cx + 8] = np.clip(pv, 0, 255)
    return np.concatenate([p.astype(np.uint8).ravel() for p in (oy, ou, ov)])


I4_QPS = [20, 28, 37, 46]


@pytest.mark.parametrize("qp", I4_QPS)
@pytest.mark.parametrize("cabac", [1, 0])
def test_oracle_rebuilds_known_intra4_macroblocks_as_the_clauses_say(hw, oracle, tmp_path, qp, cabac):
    w, h = I16_SIZE
    path = str(tmp_path / "i4.mp4")
    gold = make_clip(path, width=w, height=h, cabac=cabac, **dict(clip_args(qp, 0), global_mv=6, frames=12, gop=12))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == 12
    for k in range(1, 12):
        assert np.array_equal(frames[k][0], expected_intra4(frames[k - 1][0], k, w, h, qp)), "picture %d" % k
        assert np.array_equal(frames[k][0], gold[k])


def test_every_intra4_mode_occurs_at_every_block():
    w, h = I16_SIZE
    mbw, seen = w // 16, set()
    for k in range(1, 12):
        for addr in range(mbw * (h // 16)):
            mx, my = addr % mbw, addr // mbw
            if (mx + my) & 1 and mx > 0 and my > 0:
                for blk in range(16):
                    seen.add((blk, (addr * 5 + blk * 7 + k) % 9))
    assert len(seen) == 16 * 9


@pytest.mark.gpu
@pytest.mark.parametrize("qp", I4_QPS)
def test_hip_path_rebuilds_known_intra4_macroblocks_as_the_clauses_say(hw, tmp_path, qp):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    w, h = I16_SIZE
    path = str(tmp_path / "i4.mp4")
    make_clip(path, width=w, height=h, yuv=False, **dict(clip_args(qp, 0), global_mv=6, frames=12, gop=12))
    got = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(12)))
    for k in range(1, 12):
        assert np.array_equal(got[k], expected_intra4(got[k - 1], k, w, h, qp)), "picture %d" % k


# ---- Intra8x8 macroblocks with known modes and levels (--global-mv 8, --t8x8 1) -------------------------------------------
def pred8(img, x0, y0, mode, top_right_ok):
    """8.3.2.2: the reference samples of the 8x8 block at (x0, y0) are filtered first (8.3.2.2.1), then one of nine
    modes.  Inside the picture left, top and top-left samples exist; the sixteen samples above are p[0..15, -1], of
    which 8..15 repeat p[7, -1] when the block above-right is not there (yet)."""
    top_ok, left_ok = y0 > 0, x0 > 0
    tl_ok = top_ok and left_ok
    t = [int(img[y0 - 1, x0 + i]) for i in range(8)] if top_ok else None
    if top_ok:
        t += [int(img[y0 - 1, x0 + 8 + i]) for i in range(8)] if top_right_ok else [t[7]] * 8
    lf = [int(img[y0 + i, x0 - 1]) for i in range(8)] if left_ok else None
    tl = int(img[y0 - 1, x0 - 1]) if tl_ok else None
    ft = fl = ftl = None
    if top_ok:
        ft = [0] * 16
        ft[0] = (tl + 2 * t[0] + t[1] + 2) >> 2 if tl_ok else (3 * t[0] + t[1] + 2) >> 2
        for i in range(1, 15):
            ft[i] = (t[i - 1] + 2 * t[i] + t[i + 1] + 2) >> 2
        ft[15] = (t[14] + 3 * t[15] + 2) >> 2
    if left_ok:
        fl = [0] * 8
        fl[0] = (tl + 2 * lf[0] + lf[1] + 2) >> 2 if tl_ok else (3 * lf[0] + lf[1] + 2) >> 2
        for i in range(1, 7):
            fl[i] = (lf[i - 1] + 2 * lf[i] + lf[i + 1] + 2) >> 2
        fl[7] = (lf[6] + 3 * lf[7] + 2) >> 2
    if tl_ok:
        ftl = (t[0] + 2 * tl + lf[0] + 2) >> 2
    elif top_ok and tl is not None:
        ftl = (3 * tl + t[0] + 2) >> 2

    def p(x, y):   # the filtered samples by the clause's coordinates
        if x == -1 and y == -1:
            return ftl
        return ft[x] if y == -1 else fl[y]
    out = np.zeros((8, 8), dtype=np.int64)
    if mode == 2:
        if top_ok and left_ok:
            out[:] = (sum(ft[:8]) + sum(fl) + 8) >> 4
        elif top_ok or left_ok:
            out[:] = (sum(ft[:8] if top_ok else fl) + 4) >> 3
        else:
            out[:] = 128
        return out
    for y in range(8):
        for x in range(8):
            if mode == 0:
                v = p(x, -1)
            elif mode == 1:
                v = p(-1, y)
            elif mode == 3:
                v = (p(14, -1) + 3 * p(15, -1) + 2) >> 2 if (x == 7 and y == 7) else (p(x + y, -1) + 2 * p(x + y + 1, -1) + p(x + y + 2, -1) + 2) >> 2
            elif mode == 4:
                if x > y:
                    v = (p(x - y - 2, -1) + 2 * p(x - y - 1, -1) + p(x - y, -1) + 2) >> 2
                elif x < y:
                    v = (p(-1, y - x - 2) + 2 * p(-1, y - x - 1) + p(-1, y - x) + 2) >> 2
                else:
                    v = (p(0, -1) + 2 * p(-1, -1) + p(-1, 0) + 2) >> 2
            elif mode == 5:
                z, s = 2 * x - y, x - (y >> 1)
                if z >= 0 and z % 2 == 0:
                    v = (p(s - 1, -1) + p(s, -1) + 1) >> 1
                elif z >= 0:
                    v = (p(s - 2, -1) + 2 * p(s - 1, -1) + p(s, -1) + 2) >> 2
                elif z == -1:
                    v = (p(-1, 0) + 2 * p(-1, -1) + p(0, -1) + 2) >> 2
                else:
                    v = (p(-1, y - 2 * x - 1) + 2 * p(-1, y - 2 * x - 2) + p(-1, y - 2 * x - 3) + 2) >> 2
            elif mode == 6:
                z, s = 2 * y - x, y - (x >> 1)
                if z >= 0 and z % 2 == 0:
                    v = (p(-1, s - 1) + p(-1, s) + 1) >> 1
                elif z >= 0:
                    v = (p(-1, s - 2) + 2 * p(-1, s - 1) + p(-1, s) + 2) >> 2
                elif z == -1:
                    v = (p(-1, 0) + 2 * p(-1, -1) + p(0, -1) + 2) >> 2
                else:
                    v = (p(x - 2 * y - 1, -1) + 2 * p(x - 2 * y - 2, -1) + p(x - 2 * y - 3, -1) + 2) >> 2
            elif mode == 7:
                s = x + (y >> 1)
                v = (p(s, -1) + p(s + 1, -1) + 1) >> 1 if y % 2 == 0 else (p(s, -1) + 2 * p(s + 1, -1) + p(s + 2, -1) + 2) >> 2
            else:
                z, s = x + 2 * y, y + (x >> 1)
                if z > 13:
                    v = p(-1, 7)
                elif z == 13:
                    v = (p(-1, 6) + 3 * p(-1, 7) + 2) >> 2
                elif z % 2 == 0:
                    v = (p(-1, s) + p(-1, s + 1) + 1) >> 1
                else:
                    v = (p(-1, s) + 2 * p(-1, s + 1) + p(-1, s + 2) + 2) >> 2
            out[y, x] = v
    return out


def expected_intra8(prev, k, w, h, qp):
    mv = global_mv_of(k, w, h)
    y, u, v = planes(prev, w, h)
    sy, su, sv = shift_luma(y, *mv).astype(np.int64), shift_chroma(u, *mv).astype(np.int64), shift_chroma(v, *mv).astype(np.int64)
    oy, ou, ov = np.zeros_like(sy), np.zeros_like(su), np.zeros_like(sv)
    mbw, mbh, qpc = w // 16, h // 16, chroma_qp(qp, 0)
    for addr in range(mbw * mbh):
        mx, my = addr % mbw, addr // mbw
        ly, lx, cy, cx = my * 16, mx * 16, my * 8, mx * 8
        if (mx + my) & 1:
            inside = mx > 0 and my > 0
            cmode = (addr * 3 + k) % 4
            if (cmode == 1 and mx == 0) or (cmode == 2 and my == 0) or (cmode == 3 and not inside):
                cmode = 0
            for b8 in range(4):
                bx, by = b8 & 1, b8 >> 1
                mode = (addr * 5 + b8 * 7 + k) % 9 if inside else 2
                # above-right: block 0 - the macroblock above; 1 - the macroblock above-right; 2 - block 1; 3 - never
                tr_ok = (my > 0) if b8 == 0 else ((my > 0 and mx + 1 < mbw) if b8 == 1 else b8 == 2)
                c = np.array([[global_level(k, addr, 5, b8, i * 8 + j) for j in range(8)] for i in range(8)], dtype=np.int64)
                x0, y0 = lx + bx * 8, ly + by * 8
                oy[y0:y0 + 8, x0:x0 + 8] = np.clip(pred8(oy, x0, y0, mode, tr_ok) + inverse8x8(c, qp, 6), 0, 255)
            pu = pred_chroma(ou, cx, cy, cmode) + residual_chroma(k, addr, 0, qpc, True)
            pv = pred_chroma(ov, cx, cy, cmode) + residual_chroma(k, addr, 1, qpc, True)
        else:
            oy[ly:ly + 16, lx:lx + 16] = np.clip(sy[ly:ly + 16, lx:lx + 16] + residual_luma8(k, addr, qp), 0, 255)
            pu = su[cy:cy + 8, cx:cx + 8] + residual_chroma(k, addr, 0, qpc)
            pv = sv[cy:cy + 8, cx:cx + 8] + residual_chroma(k, addr, 1, qpc)
        ou[cy:cy + 8, cx:cx + 8] = np.clip(pu, 0, 255)
        ov[cy:cy + 8, cx:cx + 8] = np.clip(pv, 0, 255)
    return np.concatenate([p.astype(np.uint8).ravel() for p in (oy, ou, ov)])


@pytest.mark.parametrize("qp", I4_QPS)
@pytest.mark.parametrize("cabac", [1, 0])
def test_oracle_rebuilds_known_intra8_macroblocks_as_the_clauses_say(hw, oracle, tmp_path, qp, cabac):
    w, h = I16_SIZE
    path = str(tmp_path / "i8.mp4")
    gold = make_clip(path, width=w, height=h, cabac=cabac, **dict(clip_args(qp, 0), global_mv=8, t8x8=1, frames=12, gop=12))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == 12
    for k in range(1, 12):
        assert np.array_equal(frames[k][0], expected_intra8(frames[k - 1][0], k, w, h, qp)), "picture %d" % k
        assert np.array_equal(frames[k][0], gold[k])


@pytest.mark.gpu
@pytest.mark.parametrize("qp", I4_QPS)
def test_hip_path_rebuilds_known_intra8_macroblocks_as_the_clauses_say(hw, tmp_path, qp):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    w, h = I16_SIZE
    path = str(tmp_path / "i8.mp4")
    make_clip(path, width=w, height=h, yuv=False, **dict(clip_args(qp, 0), global_mv=8, t8x8=1, frames=12, gop=12))
    got = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(12)))
    for k in range(1, 12):
        assert np.array_equal(got[k], expected_intra8(got[k - 1], k, w, h, qp)), "picture %d" % k


# ---- the same with scaling matrices (--scaling 4: eight explicit lists in the SPS) -----------------------------------------
# LevelScale = weightScale * normAdjust with the list that belongs to the block (8.5.9): Intra / Inter x Y / Cb / Cr for
# 4x4, Intra / Inter Y for 8x8; the DC paths take entry (0, 0) of theirs.  With unequal weights the rounding terms of the
# scaling rules count (with the flat 16 they never do for 4x4 blocks).
@pytest.fixture
def scaling_on():
    SCALING["on"] = True
    yield
    SCALING["on"] = False


SCALED = [("inter4", 4, 0, (64, 48), [10, 17, 23, 24, 31, 44]), ("inter8", 4, 1, (64, 48), [12, 29, 35, 36, 47]),
          ("intra16", 5, 0, I16_SIZE, [8, 21, 35, 36, 45]), ("intra4", 6, 0, I16_SIZE, [15, 26, 40]), ("intra8", 8, 1, I16_SIZE, [16, 33, 41])]
SCALED_CASES = [(name, mode, t8, size, qp) for name, mode, t8, size, qps in SCALED for qp in qps]


def scaled_expected(name, prev, k, w, h, qp):
    return {"inter4": lambda: expected(prev, k, w, h, qp, 0), "inter8": lambda: expected(prev, k, w, h, qp, 0, True),
            "intra16": lambda: expected_intra16(prev, k, w, h, qp), "intra4": lambda: expected_intra4(prev, k, w, h, qp),
            "intra8": lambda: expected_intra8(prev, k, w, h, qp)}[name]()


@pytest.mark.parametrize("name,mode,t8,size,qp", SCALED_CASES, ids=["%s_qp%d" % (c[0], c[4]) for c in SCALED_CASES])
def test_oracle_applies_scaling_matrices_as_the_clauses_say(hw, oracle, tmp_path, scaling_on, name, mode, t8, size, qp):
    w, h = size
    path = str(tmp_path / "sc.mp4")
    gold = make_clip(path, width=w, height=h, **dict(clip_args(qp, 0), global_mv=mode, t8x8=t8, scaling=4, frames=6, gop=6))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == 6
    for k in range(1, 6):
        assert np.array_equal(frames[k][0], scaled_expected(name, frames[k - 1][0], k, w, h, qp)), "picture %d" % k
        assert np.array_equal(frames[k][0], gold[k])


@pytest.mark.gpu
@pytest.mark.parametrize("name,mode,t8,size,qp", SCALED_CASES, ids=["%s_qp%d" % (c[0], c[4]) for c in SCALED_CASES])
def test_hip_path_applies_scaling_matrices_as_the_clauses_say(hw, tmp_path, scaling_on, name, mode, t8, size, qp):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    w, h = size
    path = str(tmp_path / "sc.mp4")
    make_clip(path, width=w, height=h, yuv=False, **dict(clip_args(qp, 0), global_mv=mode, t8x8=t8, scaling=4, frames=6, gop=6))
    got = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(6)))
    for k in range(1, 6):
        assert np.array_equal(got[k], scaled_expected(name, got[k - 1], k, w, h, qp)), "picture %d" % k


# ---- intra pictures rebuilt from nothing (--global-mv 11, --gop 1) ----------------------------------------------------------
# Every picture an IDR picture: a checkerboard of I_PCM macroblocks (their samples a function of position and picture
# number) and Intra16x16 / Intra4x4 / Intra8x8 macroblocks with known modes and levels.  Nothing of it depends on a
# previous picture or on the encoder's content: the whole picture follows from the clauses.  (On the GPU an intra-only
# picture takes k_intra's other path - neighbours through the LDS line buffers - and the heavy engine.)
def pcm_sample(k, pl, x, y):
    """corpus/h264_encoder.c: global_pcm_sample."""
    return (x * 7 + y * 13 + k * 29 + pl * 31 + ((x >> 2) ^ (y >> 3)) * 5) & 0xff


def expected_intra_picture(k, w, h, qp, t8x8):
    oy, ou, ov = np.zeros((h, w), np.int64), np.zeros((h // 2, w // 2), np.int64), np.zeros((h // 2, w // 2), np.int64)
    mbw, mbh, qpc = w // 16, h // 16, chroma_qp(qp, 0)
    for addr in range(mbw * mbh):
        mx, my = addr % mbw, addr // mbw
        ly, lx, cy, cx = my * 16, mx * 16, my * 8, mx * 8
        if (mx + my) & 1:
            oy[ly:ly + 16, lx:lx + 16] = [[pcm_sample(k, 0, lx + x, ly + y) for x in range(16)] for y in range(16)]
            ou[cy:cy + 8, cx:cx + 8] = [[pcm_sample(k, 1, cx + x, cy + y) for x in range(8)] for y in range(8)]
            ov[cy:cy + 8, cx:cx + 8] = [[pcm_sample(k, 2, cx + x, cy + y) for x in range(8)] for y in range(8)]
            continue
        inside = mx > 0 and my > 0
        family = (addr // 2 + k) % 3   # 0 Intra16x16, 1 Intra4x4, 2 Intra8x8
        if family == 2 and not t8x8:
            family = 1
        cmode = (addr * 3 + k) % 4
        if (cmode == 1 and mx == 0) or (cmode == 2 and my == 0) or (cmode == 3 and not inside):
            cmode = 0
        if family == 0:
            mode = (addr + k) % 4
            if (mode == 0 and my == 0) or (mode == 1 and mx == 0) or (mode == 3 and not inside):
                mode = 2
            oy[ly:ly + 16, lx:lx + 16] = np.clip(pred16(oy, lx, ly, mode) + residual_luma16(k, addr, qp), 0, 255)
        elif family == 1:
            for idx in range(16):
                bx, by = (idx & 1) + ((idx >> 2) & 1) * 2, ((idx >> 1) & 1) + ((idx >> 3) & 1) * 2
                blk = by * 4 + bx
                mode = (addr * 5 + blk * 7 + k) % 9 if inside else 2
                tr_ok = (my > 0 and (bx < 3 or mx + 1 < mbw)) if by == 0 else (bx, by) in ((0, 1), (0, 2), (0, 3), (2, 1), (2, 2), (2, 3), (1, 2))
                c = np.array([[global_level(k, addr, 0, blk, i * 4 + j) for j in range(4)] for i in range(4)], dtype=np.int64)
                x0, y0 = lx + bx * 4, ly + by * 4
                oy[y0:y0 + 4, x0:x0 + 4] = np.clip(pred4(oy, x0, y0, mode, tr_ok, w) + inverse4x4(scale_ac(c, qp, 0)), 0, 255)
        else:
            for b8 in range(4):
                bx, by = b8 & 1, b8 >> 1
                mode = (addr * 5 + b8 * 7 + k) % 9 if inside else 2
                tr_ok = (my > 0) if b8 == 0 else ((my > 0 and mx + 1 < mbw) if b8 == 1 else b8 == 2)
                c = np.array([[global_level(k, addr, 5, b8, i * 8 + j) for j in range(8)] for i in range(8)], dtype=np.int64)
                x0, y0 = lx + bx * 8, ly + by * 8
                oy[y0:y0 + 8, x0:x0 + 8] = np.clip(pred8(oy, x0, y0, mode, tr_ok) + inverse8x8(c, qp, 6), 0, 255)
        ou[cy:cy + 8, cx:cx + 8] = np.clip(pred_chroma(ou, cx, cy, cmode) + residual_chroma(k, addr, 0, qpc, True), 0, 255)
        ov[cy:cy + 8, cx:cx + 8] = np.clip(pred_chroma(ov, cx, cy, cmode) + residual_chroma(k, addr, 1, qpc, True), 0, 255)
    return np.concatenate([p.astype(np.uint8).ravel() for p in (oy, ou, ov)])


INTRA_PICTURES = [(24, 1), (31, 1), (38, 1), (45, 1), (27, 0), (40, 0)]   # (QP, --t8x8)


def intra_picture_args(qp, t8):
    return dict(frames=7, gop=1, bframes=0, refs=1, global_mv=11, qp=qp, t8x8=t8, deblock=1)


@pytest.mark.parametrize("qp,t8", INTRA_PICTURES)
@pytest.mark.parametrize("cabac", [1, 0])
def test_oracle_rebuilds_intra_pictures_from_nothing(hw, oracle, tmp_path, qp, t8, cabac):
    w, h = I16_SIZE
    path = str(tmp_path / "ip.mp4")
    gold = make_clip(path, width=w, height=h, cabac=cabac, **intra_picture_args(qp, t8))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == 7
    for k in range(7):
        assert np.array_equal(frames[k][0], expected_intra_picture(k, w, h, qp, t8)), "picture %d" % k
        assert np.array_equal(frames[k][0], gold[k])


@pytest.mark.gpu
@pytest.mark.parametrize("qp,t8", INTRA_PICTURES)
def test_hip_path_rebuilds_intra_pictures_from_nothing(hw, tmp_path, qp, t8):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    w, h = I16_SIZE
    path = str(tmp_path / "ip.mp4")
    make_clip(path, width=w, height=h, yuv=False, **intra_picture_args(qp, t8))
    got = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(7)))
    for k in range(7):
        assert np.array_equal(got[k], expected_intra_picture(k, w, h, qp, t8)), "picture %d" % k


def filter_intra_picture(pic, k, w, h, qp, t8x8, alpha_div2, beta_div2):
    """The in-loop filter on such a picture: every macroblock is intra - macroblock edges bS 4, inner edges bS 3; an
    Intra8x8 macroblock has no luma edges 1 and 3 (transform_size_8x8_flag); qP is 0 for an I_PCM macroblock, and the
    thresholds of an edge come from the average of the two sides (8.7.2.2), for chroma from the chroma QPs."""
    mbw, mbh = w // 16, h // 16

    def mb_kind(addr):
        mx, my = addr % mbw, addr // mbw
        if (mx + my) & 1:
            return "pcm"
        family = (addr // 2 + k) % 3
        return "i8" if (family == 2 and t8x8) else "i"
    qpy = lambda addr: 0 if mb_kind(addr) == "pcm" else qp   # noqa: E731
    pl = [p.astype(np.int64) for p in planes(pic, w, h)]
    for my in range(mbh):
        for mx in range(mbw):
            addr = my * mbw + mx
            for vertical in (True, False):
                for e in range(4):
                    if e == 0 and (mx == 0 if vertical else my == 0):
                        continue
                    other = addr if e else (addr - 1 if vertical else addr - mbw)
                    bs = 3 if e else 4
                    for p_i, (a, chroma) in enumerate(((pl[0], False), (pl[1], True), (pl[2], True))):
                        if chroma and e % 2:
                            continue
                        if not chroma and e % 2 and mb_kind(addr) == "i8":
                            continue
                        qa, qb = (qpy(addr), qpy(other)) if not chroma else (chroma_qp(qpy(addr), 0), chroma_qp(qpy(other), 0))
                        qav = (qa + qb + 1) >> 1
                        ia, ib = clip3(0, 51, qav + 2 * alpha_div2), clip3(0, 51, qav + 2 * beta_div2)
                        alpha, beta, tc0 = ALPHA[ia], BETA[ib], (None if bs == 4 else TC0_BS3[ia])
                        n, reach, s = (8, 2, 2) if chroma else (16, 4, 1)
                        for i in range(n):
                            px = [0] * 8
                            if vertical:
                                x0, yy = (mx * 16 + e * 4) // s, (my * 16) // s + i
                                px[4 - reach:4 + reach] = [int(t) for t in a[yy, x0 - reach:x0 + reach]]
                                filter_line(px, chroma, alpha, beta, tc0)
                                a[yy, x0 - reach:x0 + reach] = px[4 - reach:4 + reach]
                            else:
                                y0, xx = (my * 16 + e * 4) // s, (mx * 16) // s + i
                                px[4 - reach:4 + reach] = [int(t) for t in a[y0 - reach:y0 + reach, xx]]
                                filter_line(px, chroma, alpha, beta, tc0)
                                a[y0 - reach:y0 + reach, xx] = px[4 - reach:4 + reach]
    return np.concatenate([np.clip(p, 0, 255).astype(np.uint8).ravel() for p in pl])


FILTERED_INTRA = [(28, 1, 0, 0), (36, 1, 0, 0), (44, 1, 2, -1), (51, 1, 0, 0), (33, 0, -2, 3), (47, 0, 6, 6)]   # (QP, --t8x8, offsets / 2)


@pytest.mark.parametrize("qp,t8,a,b", FILTERED_INTRA)
@pytest.mark.parametrize("cabac", [1, 0])
def test_oracle_filters_intra_pictures_as_the_clauses_say(hw, oracle, tmp_path, qp, t8, a, b, cabac):
    w, h = I16_SIZE
    path = str(tmp_path / "ipf.mp4")
    make_clip(path, width=w, height=h, cabac=cabac, yuv=False, **dict(intra_picture_args(qp, t8), deblock=0, alpha=a, beta=b))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == 7
    moved = 0
    for k in range(7):
        plain = expected_intra_picture(k, w, h, qp, t8)
        want = filter_intra_picture(plain, k, w, h, qp, t8, a, b)
        assert np.array_equal(frames[k][0], want), "picture %d" % k
        moved += int((want != plain).sum())
    assert moved > 0


@pytest.mark.gpu
@pytest.mark.parametrize("qp,t8,a,b", FILTERED_INTRA)
def test_hip_path_filters_intra_pictures_as_the_clauses_say(hw, tmp_path, qp, t8, a, b):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    w, h = I16_SIZE
    path = str(tmp_path / "ipf.mp4")
    make_clip(path, width=w, height=h, yuv=False, **dict(intra_picture_args(qp, t8), deblock=0, alpha=a, beta=b))
    got = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(7)))
    for k in range(7):
        want = filter_intra_picture(expected_intra_picture(k, w, h, qp, t8), k, w, h, qp, t8, a, b)
        assert np.array_equal(got[k], want), "picture %d" % k


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", [(640, 368), (1920, 1088)])
def test_hip_path_filters_a_larger_intra_picture_as_the_clauses_say(hw, tmp_path, w, h):
    """40 x 23 and 120 x 68 macroblocks: six and seventeen bands of the row pipelines (k_intra, k_deblock) and their
    hand-offs, against pictures in which no sample comes from a decoder."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    qp, t8, a, b = 37, 1, 1, 0
    path = str(tmp_path / "ipl.mp4")
    make_clip(path, width=w, height=h, yuv=False, **dict(intra_picture_args(qp, t8), frames=2, deblock=0, alpha=a, beta=b))
    got = hw.Decoder(path, num_lanes=1).retrieve_yuv([0, 1])
    for k in range(2):
        want = filter_intra_picture(expected_intra_picture(k, w, h, qp, t8), k, w, h, qp, t8, a, b)
        assert np.array_equal(got[k], want), "picture %d" % k


def test_oracle_filters_a_larger_intra_picture_as_the_clauses_say(hw, oracle, tmp_path):
    w, h, qp, t8, a, b = 640, 368, 37, 1, 1, 0
    path = str(tmp_path / "ipl.mp4")
    make_clip(path, width=w, height=h, yuv=False, **dict(intra_picture_args(qp, t8), frames=2, deblock=0, alpha=a, beta=b))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0
    for k in range(2):
        assert np.array_equal(frames[k][0], filter_intra_picture(expected_intra_picture(k, w, h, qp, t8), k, w, h, qp, t8, a, b)), "picture %d" % k


@pytest.mark.gpu
def test_hip_path_filters_a_larger_mixed_picture_as_the_clauses_say(hw, tmp_path):
    """640 x 368: the mixed pictures (known Intra16x16 macroblocks among coded inter ones, every boundary strength) over
    six bands of k_mv / k_intra / k_deblock."""
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    w, h, qp = 640, 368, 34
    path = str(tmp_path / "mixl.mp4")
    make_clip(path, width=w, height=h, yuv=False, **dict(mixed_clip_args(qp, 0, 0), frames=4, gop=4))
    got = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(4)))
    for k in range(1, 4):
        assert np.array_equal(got[k], expected_mixed_filtered(got[k - 1], k, w, h, qp, 0, 0)), "picture %d" % k

"""Inter prediction checked against a restatement that shares nothing with the oracle or the kernels.

The corpus encoder's `--global-mv` mode writes P pictures in which every macroblock is one 16x16 partition (or P_Skip)
with reference 0 and ONE vector per picture, and codes no residual; no edge gets a boundary strength, so the in-loop
filter does nothing.  Such a picture must decode to the previous picture shifted by that vector: the sample
interpolation of 8.4.2.2.1 (luma, all sixteen quarter-sample positions) and 8.4.2.2.2 (chroma, eighths), with the
reference coordinates clamped to the picture (8.4.2.2), and - with explicit weights, which in this mode follow from the
picture's number - the weighted sample prediction of 8.4.2.3.2 on top of it.  Here that is computed for whole pictures with numpy, straight
from the clauses' equations, and compared with

  * the oracle's decode of the stream (CPU) - a third statement of the filters beside oracle/h264_recon.c and
    tests/fixtures/inter_filters_host.cpp, written against the decoder's OUTPUT rather than its code, and
  * the HIP path's decode (GPU): parser -> k_mv (the vectors travel as differences) -> k_inter, including vectors that
    point far outside the picture.

With --bframes 2 the mode also writes B pictures predicted from list 0, list 1 or both (one vector per list and picture):
8.4.2.3's default average, implicit weights (from the POC distances) and explicit weights of both lists are restated
the same way.
"""
import numpy as np
import pytest

from tests.corpus_util import make_clip

SIZES = [(64, 48), (176, 144)]
FRAMES = 36


def global_mv_of(k, width, height):
    """corpus/h264_encoder.c: global_mv_of (quarter samples)."""
    fx, fy = k & 3, (k >> 2) & 3
    ix, iy = ((k * 5) % 9) - 4, ((k * 7) % 5) - 2
    if k % 6 == 5:
        ix = (-1 if k % 12 == 5 else 1) * (width // 2 + 3 * k)
    if k % 7 == 6:
        iy = (-1 if k % 14 == 6 else 1) * (height // 2 + 2 * k)
    return ix * 4 + fx, iy * 4 + fy


def clip8(a):
    return np.clip(a, 0, 255)


def shift_luma(ref, mvx, mvy):
    """8.4.2.2.1 for a whole picture: predPartL[x, y] for every (x, y), one vector."""
    h, w = ref.shape
    ox, oy, fx, fy = mvx >> 2, mvy >> 2, mvx & 3, mvy & 3
    m = max(abs(ox), abs(oy)) + 8
    r = np.pad(ref.astype(np.int64), m, mode="edge")   # = Clip3 of the reference coordinates (8-230, 8-231)

    def G(dx, dy):   # integer sample at (x + ox + dx, y + oy + dy)
        return r[m + oy + dy:m + oy + dy + h, m + ox + dx:m + ox + dx + w]

    def tap(f, dx, dy, horizontal):   # the six-tap filter (1, -5, 20, 20, -5, 1) over f along one axis
        o = [(-2, 0), (-1, 0), (0, 0), (1, 0), (2, 0), (3, 0)] if horizontal else [(0, -2), (0, -1), (0, 0), (0, 1), (0, 2), (0, 3)]
        c = (1, -5, 20, 20, -5, 1)
        return sum(ck * f(dx + ax, dy + ay) for ck, (ax, ay) in zip(c, o))

    def b1(dx, dy):   # intermediate of the half sample to the right of G(dx, dy) (8-241)
        return tap(G, dx, dy, True)

    def h1(dx, dy):   # ... below G(dx, dy) (8-242)
        return tap(G, dx, dy, False)

    def b(dx, dy):
        return clip8((b1(dx, dy) + 16) >> 5)   # (8-243)

    def hh(dx, dy):
        return clip8((h1(dx, dy) + 16) >> 5)   # (8-244)

    j = clip8((tap(b1, 0, 0, False) + 512) >> 10)   # (8-245, 8-246): six-tap over the intermediates, vertically
    g00, bb, hv = G(0, 0), b(0, 0), hh(0, 0)
    s, mm = b(0, 1), hh(1, 0)   # the half samples one row below / one column to the right
    avg = lambda p, q: (p + q + 1) >> 1   # noqa: E731  (8-250 .. 8-261)
    table = {
        (0, 0): g00, (1, 0): avg(g00, bb), (2, 0): bb, (3, 0): avg(G(1, 0), bb),
        (0, 1): avg(g00, hv), (1, 1): avg(bb, hv), (2, 1): avg(bb, j), (3, 1): avg(bb, mm),
        (0, 2): hv, (1, 2): avg(hv, j), (2, 2): j, (3, 2): avg(j, mm),
        (0, 3): avg(G(0, 1), hv), (1, 3): avg(hv, s), (2, 3): avg(j, s), (3, 3): avg(mm, s),
    }
    return table[(fx, fy)].astype(np.uint8)


def shift_chroma(ref, mvx, mvy):
    """8.4.2.2.2 for a whole 4:2:0 chroma plane: the luma vector read in eighths of a chroma sample."""
    h, w = ref.shape
    ox, oy, fx, fy = mvx >> 3, mvy >> 3, mvx & 7, mvy & 7
    m = max(abs(ox), abs(oy)) + 2
    r = np.pad(ref.astype(np.int64), m, mode="edge")

    def C(dx, dy):
        return r[m + oy + dy:m + oy + dy + h, m + ox + dx:m + ox + dx + w]
    out = ((8 - fx) * (8 - fy) * C(0, 0) + fx * (8 - fy) * C(1, 0) + (8 - fx) * fy * C(0, 1) + fx * fy * C(1, 1) + 32) >> 6   # (8-270)
    return out.astype(np.uint8)


def planes(frame, w, h):
    y = frame[:w * h].reshape(h, w)
    u = frame[w * h:w * h + (w // 2) * (h // 2)].reshape(h // 2, w // 2)
    v = frame[w * h + (w // 2) * (h // 2):].reshape(h // 2, w // 2)
    return y, u, v


def weights_of(k, ld, lst=0, b_slice=False):
    """corpus/h264_encoder.c: write_pwt in --global-mv mode: (log2 denominator, weight, offset) for Y, Cb, Cr of the
    entries of list `lst` in picture k."""
    cd = (ld + 6) & 7
    kk = k + 7 * lst
    cap = lambda d: (63 if d == 7 else 64) if (b_slice and d >= 6) else 127   # noqa: E731  (B: w0 + w1 <= 128, 127 at denominator 7)
    wy = (1 << ld) + ((kk * 3) % 7 - 3 if ld >= 3 else (kk % 3 - 1 if ld >= 1 else 0))
    out = [(ld, min(cap(ld), wy), (kk * 5) % 11 - 5)]   # (a coded weight is an se(v) in -128 .. 127)
    for pl in range(2):
        wc = (1 << cd) + ((kk + pl) % 5 - 2 if cd >= 3 else ((kk + pl) % 3 - 1 if cd >= 1 else 0))
        out.append((cd, min(cap(cd), wc), (kk * 3 + pl * 2) % 9 - 4))
    return out


def weighted(pred, log_wd, w, o):
    """8.4.2.3.2, one list: (8-297)."""
    p = pred.astype(np.int64)
    v = ((p * w + (1 << (log_wd - 1))) >> log_wd) + o if log_wd >= 1 else p * w + o
    return clip8(v).astype(np.uint8)


def weighted_bi(p0, p1, log_wd, w0, w1, o0, o1):
    """8.4.2.3.2, both lists: (8-299)."""
    v = ((p0.astype(np.int64) * w0 + p1.astype(np.int64) * w1 + (1 << log_wd)) >> (log_wd + 1)) + ((o0 + o1 + 1) >> 1)
    return clip8(v).astype(np.uint8)


def implicit_weights(poc_cur, poc0, poc1):
    """8.4.2.3.1 (implicit mode, frames, short-term references): (w0, w1), logWD = 5, offsets 0."""
    clip3 = lambda lo, hi, x: max(lo, min(hi, x))   # noqa: E731
    tb, td = clip3(-128, 127, poc_cur - poc0), clip3(-128, 127, poc1 - poc0)
    if td == 0:
        return 32, 32
    tx = int((16384 + abs(td // 2 if td > 0 else -((-td) // 2))) / td)   # (16384 + Abs(td / 2)) / td, "/" truncating
    dsf = clip3(-1024, 1023, (tb * tx + 32) >> 6)
    if (dsf >> 2) < -64 or (dsf >> 2) > 128:
        return 32, 32
    return 64 - (dsf >> 2), dsf >> 2


def shifted(frame, mv, w, h):
    y, u, v = planes(frame, w, h)
    return [shift_luma(y, *mv), shift_chroma(u, *mv), shift_chroma(v, *mv)]


def expected_from(prev, k, w, h, wp_denom=None):
    """P picture k of a clip without B pictures."""
    out = shifted(prev, global_mv_of(k, w, h), w, h)
    if wp_denom is not None:
        out = [weighted(p, *wt) for p, wt in zip(out, weights_of(k, wp_denom))]
    return np.concatenate([p.ravel() for p in out])


def expected_b(dec, d, w, h, mode):
    """Picture d of an I B B P B B P ... clip (--bframes 2): anchors refer to the anchor before them, a B picture to the
    anchors around it (list 0: the one before, list 1: the one after); POC = 2 * display position.
    mode: None (default prediction), "implicit" (weighted_bipred_idc 2) or a luma denominator (explicit weights)."""
    explicit = mode is not None and mode != "implicit"
    if d % 3 == 0:   # P
        out = shifted(dec[d - 3], global_mv_of(d, w, h), w, h)
        if explicit:
            out = [weighted(p, *wt) for p, wt in zip(out, weights_of(d, mode))]
        return np.concatenate([p.ravel() for p in out])
    a, kind = d - d % 3, 1 + d % 3   # 1 L0, 2 L1, 3 Bi (corpus/h264_encoder.c: decide_b in --global-mv mode)
    p0 = shifted(dec[a], global_mv_of(d, w, h), w, h)
    p1 = shifted(dec[a + 3], global_mv_of(d + 17, w, h), w, h)
    if kind != 3:
        out, lst = (p0, 0) if kind == 1 else (p1, 1)
        if explicit:
            out = [weighted(p, *wt) for p, wt in zip(out, weights_of(d, mode, lst, True))]
    elif explicit:
        w0s, w1s = weights_of(d, mode, 0, True), weights_of(d, mode, 1, True)
        out = [weighted_bi(x, y, a0[0], a0[1], a1[1], a0[2], a1[2]) for x, y, a0, a1 in zip(p0, p1, w0s, w1s)]
    elif mode == "implicit":
        w0, w1 = implicit_weights(2 * d, 2 * a, 2 * (a + 3))
        out = [weighted_bi(x, y, 5, w0, w1, 0, 0) for x, y in zip(p0, p1)]
    else:
        out = [((x.astype(np.int64) + y + 1) >> 1).astype(np.uint8) for x, y in zip(p0, p1)]   # (8-273)
    return np.concatenate([p.ravel() for p in out])


def test_the_vectors_cover_what_they_should():
    for w, h in SIZES:
        mvs = [global_mv_of(k, w, h) for k in range(1, FRAMES)]
        assert {(x & 3, y & 3) for x, y in mvs} == {(a, b) for a in range(4) for b in range(4)}
        assert {x & 7 for x, _ in mvs} == set(range(8)) and {y & 7 for _, y in mvs} == set(range(8))
        assert any(x >> 2 <= -w // 2 for x, _ in mvs) and any(x >> 2 >= w // 2 for x, _ in mvs)
        assert any(y >> 2 <= -h // 2 for _, y in mvs) and any(y >> 2 >= h // 2 for _, y in mvs)


WP_DENOMS = [None, 5, 0, 7, 2]   # None: default prediction; else explicit weights with this luma denominator
                                 # (the chroma one is (d + 6) & 7: 3, 6, 5, 0 - denominators 0 and >= 1 take different rules)


def clip_args(wp_denom):
    return {} if wp_denom is None else dict(explicit_wp=1, wp_denom=wp_denom)


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("cabac", [1, 0])
@pytest.mark.parametrize("wp_denom", WP_DENOMS)
def test_oracle_decodes_shifted_pictures_as_the_clauses_say(hw, oracle, tmp_path, w, h, cabac, wp_denom):
    path = str(tmp_path / "gmv.mp4")
    gold = make_clip(path, width=w, height=h, frames=FRAMES, gop=FRAMES, bframes=0, refs=1, global_mv=1, cabac=cabac,
                     **clip_args(wp_denom))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == FRAMES
    for k in range(1, FRAMES):
        want = expected_from(frames[k - 1][0], k, w, h, wp_denom)
        assert np.array_equal(frames[k][0], want), "picture %d, vector %s" % (k, global_mv_of(k, w, h))
        assert np.array_equal(frames[k][0], gold[k])


B_FRAMES = 37   # I B B P ... P: the last picture is an anchor
B_MODES = [None, "implicit", 5, 0, 7, 2]


def b_clip_args(mode):
    return {} if mode is None else (dict(implicit_wp=1) if mode == "implicit" else dict(explicit_wp=1, wp_denom=mode))


def test_implicit_weights_follow_the_distances():
    assert implicit_weights(2, 0, 6) == (43, 21) and implicit_weights(4, 0, 6) == (22, 42)
    assert implicit_weights(6, 0, 12) == (32, 32)


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("cabac", [1, 0])
@pytest.mark.parametrize("mode", B_MODES)
def test_oracle_decodes_bipredicted_pictures_as_the_clauses_say(hw, oracle, tmp_path, w, h, cabac, mode):
    """B pictures from list 0, list 1 or both (8.4.2.3: default, implicit and explicit weighting)."""
    path = str(tmp_path / "gmvb.mp4")
    gold = make_clip(path, width=w, height=h, frames=B_FRAMES, gop=B_FRAMES, bframes=2, refs=1, global_mv=1, cabac=cabac,
                     **b_clip_args(mode))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == B_FRAMES
    dec = [f[0] for f in frames]
    for d in range(1, B_FRAMES):
        assert np.array_equal(dec[d], expected_b(dec, d, w, h, mode)), "picture %d (%s)" % (d, "P" if d % 3 == 0 else "B")
        assert np.array_equal(dec[d], gold[d])


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("mode", B_MODES)
def test_hip_path_decodes_bipredicted_pictures_as_the_clauses_say(hw, tmp_path, w, h, mode):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = str(tmp_path / "gmvb.mp4")
    make_clip(path, width=w, height=h, frames=B_FRAMES, gop=B_FRAMES, bframes=2, refs=1, global_mv=1, yuv=False, **b_clip_args(mode))
    dec = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(B_FRAMES)))
    for d in range(1, B_FRAMES):
        assert np.array_equal(dec[d], expected_b(dec, d, w, h, mode)), "picture %d (%s)" % (d, "P" if d % 3 == 0 else "B")


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("wp_denom", WP_DENOMS)
def test_hip_path_decodes_shifted_pictures_as_the_clauses_say(hw, tmp_path, w, h, wp_denom):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = str(tmp_path / "gmv.mp4")
    make_clip(path, width=w, height=h, frames=FRAMES, gop=FRAMES, bframes=0, refs=1, global_mv=1, yuv=False, **clip_args(wp_denom))
    got = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(FRAMES)))
    for k in range(1, FRAMES):
        want = expected_from(got[k - 1], k, w, h, wp_denom)
        assert np.array_equal(got[k], want), "picture %d, vector %s" % (k, global_mv_of(k, w, h))


# ---- temporal direct prediction (8.4.1.2.3) --------------------------------------------------------------------------------
# --global-mv 9 with --temporal-direct 1: the anchors are shifted copies of each other as before; every macroblock of a B
# picture is B_Skip.  Its vectors come from the co-located macroblock of the anchor behind it (list 1, entry 0), which
# carries that anchor's one vector mvCol and refers to the anchor in front (list 0, entry 0):
#     mvL0 = (DistScaleFactor * mvCol + 128) >> 8,   mvL1 = mvL0 - mvCol
def temporal_direct_vectors(d, w, h):
    a = d - d % 3
    mv_col = global_mv_of(a + 3, w, h)
    tb, td = 2 * (d - a), 2 * 3               # POC distances: current - list 0 reference, list 1 reference - list 0 reference
    tx = (16384 + abs(td // 2)) // td
    dsf = max(-1024, min(1023, (tb * tx + 32) >> 6))
    mv0 = tuple((dsf * c + 128) >> 8 for c in mv_col)
    return mv0, tuple(m0 - c for m0, c in zip(mv0, mv_col))


def expected_direct(dec, d, w, h, implicit):
    if d % 3 == 0:
        return np.concatenate([p.ravel() for p in shifted(dec[d - 3], global_mv_of(d, w, h), w, h)])
    a = d - d % 3
    mv0, mv1 = temporal_direct_vectors(d, w, h)
    p0, p1 = shifted(dec[a], mv0, w, h), shifted(dec[a + 3], mv1, w, h)
    if implicit:
        w0, w1 = implicit_weights(2 * d, 2 * a, 2 * (a + 3))
        out = [weighted_bi(x, y, 5, w0, w1, 0, 0) for x, y in zip(p0, p1)]
    else:
        out = [((x.astype(np.int64) + y + 1) >> 1).astype(np.uint8) for x, y in zip(p0, p1)]
    return np.concatenate([p.ravel() for p in out])


def direct_clip_args(implicit):
    return dict(frames=B_FRAMES, gop=B_FRAMES, bframes=2, refs=1, global_mv=9, temporal_direct=1, implicit_wp=1 if implicit else 0)


@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("cabac", [1, 0])
@pytest.mark.parametrize("implicit", [False, True])
def test_oracle_derives_temporal_direct_vectors_as_the_clause_says(hw, oracle, tmp_path, w, h, cabac, implicit):
    path = str(tmp_path / "td.mp4")
    gold = make_clip(path, width=w, height=h, cabac=cabac, **direct_clip_args(implicit))
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == B_FRAMES
    dec = [f[0] for f in frames]
    for d in range(1, B_FRAMES):
        assert np.array_equal(dec[d], expected_direct(dec, d, w, h, implicit)), "picture %d (%s)" % (d, "P" if d % 3 == 0 else "B")
        assert np.array_equal(dec[d], gold[d])


@pytest.mark.gpu
@pytest.mark.parametrize("w,h", SIZES)
@pytest.mark.parametrize("implicit", [False, True])
def test_hip_path_derives_temporal_direct_vectors_as_the_clause_says(hw, tmp_path, w, h, implicit):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    path = str(tmp_path / "td.mp4")
    make_clip(path, width=w, height=h, yuv=False, **direct_clip_args(implicit))
    dec = hw.Decoder(path, num_lanes=1).retrieve_yuv(list(range(B_FRAMES)))
    for d in range(1, B_FRAMES):
        assert np.array_equal(dec[d], expected_direct(dec, d, w, h, implicit)), "picture %d (%s)" % (d, "P" if d % 3 == 0 else "B")

"""A second statement of the HEVC reconstruction arithmetic, in numpy, written from the clauses of ITU-T H.265 (as
remembered - no copy of the standard is in the image) and NOT from oracle/hevc_recon.c: different structure (matrices
and whole-array operations where the oracle loops over samples), different derivations (the transform matrix from its
31 cosine magnitudes, the inverse angles from the angles, thresholds from their formulas).  tests/test_hevc_independent.py
compares it with the oracle and, on the GPU, with the HIP kernels: the defence against one reading of a clause being
mirrored in encoder, oracle and product alike.  Test infrastructure only."""
import numpy as np

# ---------------------------------------------------------------------------------------------------- transforms
# 8.6.4.2: the 32-point matrix holds, at row k and column n, +-M[j] where j is the angle k (2n + 1) pi / 64 folded
# into the first quadrant; the smaller transforms use every 2nd / 4th / 8th row
MAG = [64, 90, 90, 90, 89, 88, 87, 85, 83, 82, 80, 78, 75, 73, 70, 67, 64, 61, 57, 54, 50, 46, 43, 38, 36, 31, 25, 22, 18, 13, 9, 4, 0]


def trans_matrix(n=32):
    m = np.zeros((32, 32), dtype=np.int64)
    for k in range(32):
        for c in range(32):
            a = (k * (2 * c + 1)) % 128          # in units of pi / 64; the cosine has period 128
            if a > 64:
                a = 128 - a                      # cos(2 pi - x) = cos x
            m[k, c] = MAG[a] if a <= 32 else -MAG[64 - a]   # cos(pi - x) = - cos x
    step = 32 // n
    return m[::step, :n][:n]


DST4 = np.array([[29, 55, 74, 84], [74, 74, 0, -74], [84, -29, -74, 55], [55, -84, 74, -29]], dtype=np.int64)
LEVEL_SCALE = [40, 45, 51, 57, 64, 72]


def residual(levels, qp, intra, c_idx, transform_skip=False, m=16):
    """8.6.2 - 8.6.4 for one n x n block of levels (raster: rows are vertical frequencies), 8-bit video, flat lists."""
    lev = np.asarray(levels, dtype=np.int64)
    n = lev.shape[0]
    log2 = n.bit_length() - 1
    bd_shift = 8 + log2 - 5
    d = np.clip((lev * m * LEVEL_SCALE[qp % 6] * (1 << (qp // 6)) + (1 << (bd_shift - 1))) >> bd_shift, -32768, 32767)
    if transform_skip:
        r = d << 7
    else:
        t = DST4 if (intra and c_idx == 0 and n == 4) else trans_matrix(n)
        # first the columns: sample i of column x is sum_j T[j][i] d[j][x]  ->  T^T d ; clipped after (x + 64) >> 7
        e = np.clip((t.T @ d + 64) >> 7, -32768, 32767)
        # then the rows: sample i of row y is sum_j T[j][i] e[y][j]  ->  e T
        r = e @ t
    return (r + (1 << 11)) >> 12


# ------------------------------------------------------------------------------------------------- interpolation
LUMA_TAPS = {0: [0, 0, 0, 64, 0, 0, 0, 0], 1: [-1, 4, -10, 58, 17, -5, 1, 0], 2: [-1, 4, -11, 40, 40, -11, 4, -1]}
LUMA_TAPS[3] = LUMA_TAPS[1][::-1]
CHROMA_TAPS = {0: [0, 64, 0, 0], 1: [-2, 58, 10, -2], 2: [-4, 54, 16, -2], 3: [-6, 46, 28, -4], 4: [-4, 36, 36, -4]}
for _k in (5, 6, 7):
    CHROMA_TAPS[_k] = CHROMA_TAPS[8 - _k][::-1]


def pred_samples(ref, x, y, w, h, mvx, mvy, chroma):
    """predSamplesLX of 8.5.3.3.3 at 14 bits: the reference padded by replication, one separable filter pass per axis."""
    ref = np.asarray(ref, dtype=np.int64)
    taps = CHROMA_TAPS if chroma else LUMA_TAPS
    fb = 3 if chroma else 2
    nt = 4 if chroma else 8
    before = 1 if chroma else 3
    xf, yf = mvx & ((1 << fb) - 1), mvy & ((1 << fb) - 1)
    x0, y0 = x + (mvx >> fb) - before, y + (mvy >> fb) - before
    ys = np.clip(np.arange(y0, y0 + h + nt - 1), 0, ref.shape[0] - 1)
    xs = np.clip(np.arange(x0, x0 + w + nt - 1), 0, ref.shape[1] - 1)
    win = ref[np.ix_(ys, xs)]                       # (h + taps - 1) x (w + taps - 1)
    fx, fy = np.array(taps[xf]), np.array(taps[yf])
    if xf == 0 and yf == 0:
        return win[before:before + h, before:before + w] << 6
    if yf == 0:
        rows = win[before:before + h]
        return sum(fx[t] * rows[:, t:t + w] for t in range(nt))      # shift1 = 0 at 8 bits
    if xf == 0:
        cols = win[:, before:before + w]
        return sum(fy[t] * cols[t:t + h] for t in range(nt))
    hor = sum(fx[t] * win[:, t:t + w] for t in range(nt))
    return sum(fy[t] * hor[t:t + h] for t in range(nt)) >> 6


def inter_default(ref, x, y, w, h, mvx, mvy, chroma):
    return np.clip((pred_samples(ref, x, y, w, h, mvx, mvy, chroma) + 32) >> 6, 0, 255)


def inter_bi(p0, p1):
    return np.clip((p0 + p1 + 64) >> 7, 0, 255)


def inter_weighted_uni(p, w0, o0, log2_denom):
    lwd = log2_denom + 6
    return np.clip(((p * w0 + (1 << (lwd - 1))) >> lwd) + o0, 0, 255)


def inter_weighted_bi(p0, p1, w0, w1, o0, o1, log2_denom):
    lwd = log2_denom + 6
    return np.clip((p0 * w0 + p1 * w1 + ((o0 + o1 + 1) << lwd)) >> (lwd + 1), 0, 255)


# --------------------------------------------------------------------------------------------------------- intra
ANGLES = {2: 32, 3: 26, 4: 21, 5: 17, 6: 13, 7: 9, 8: 5, 9: 2, 10: 0}
for _m in range(11, 19):
    ANGLES[_m] = -ANGLES[20 - _m]                 # 11 .. 18 mirror 9 .. 2 with the sign turned
for _m in range(19, 35):
    ANGLES[_m] = ANGLES[36 - _m]                  # the vertical family mirrors the horizontal one


def inv_angle(a):
    return int(round(8192.0 / a))                 # Table 8-5 is 256 * 32 / intraPredAngle, rounded


def intra_filtered(left, top, n, c_idx, mode, strong):
    """8.4.4.2.3.  left / top: arrays of 2n + 1 with index 0 = the corner p[-1][-1]."""
    left, top = np.array(left, dtype=np.int64), np.array(top, dtype=np.int64)
    if c_idx != 0 or mode == 1 or n == 4:
        return left, top
    thres = {8: 7, 16: 1, 32: 0}[n]
    if min(abs(mode - 26), abs(mode - 10)) <= thres:
        return left, top
    if strong and n == 32 and abs(left[0] + top[64] - 2 * top[32]) < 8 and abs(left[0] + left[64] - 2 * left[32]) < 8:
        i = np.arange(1, 64)
        fl, ft = left.copy(), top.copy()
        fl[1:64] = ((64 - i) * left[0] + i * left[64] + 32) >> 6
        ft[1:64] = ((64 - i) * top[0] + i * top[64] + 32) >> 6
        return fl, ft
    line = np.concatenate([left[:0:-1], [left[0]], top[1:]])    # bottom-left ... corner ... top-right as one line
    out = line.copy()
    out[1:-1] = (line[:-2] + 2 * line[1:-1] + line[2:] + 2) >> 2
    k = 2 * n
    return np.concatenate([[out[k]], out[:k][::-1]]), np.concatenate([[out[k]], out[k + 1:]])


def intra_predict(left, top, n, c_idx, mode, strong=True):
    """8.4.4.2.4 - 8.4.4.2.6 from substituted neighbours; returns an n x n array [y][x]."""
    left, top = intra_filtered(left, top, n, c_idx, mode, strong)
    log2 = n.bit_length() - 1
    x = np.arange(n)[None, :]
    y = np.arange(n)[:, None]
    if mode == 0:
        return ((n - 1 - x) * left[1:n + 1][:, None] + (x + 1) * top[n + 1] + (n - 1 - y) * top[1:n + 1][None, :] +
                (y + 1) * left[n + 1] + n) >> (log2 + 1)
    if mode == 1:
        dc = (left[1:n + 1].sum() + top[1:n + 1].sum() + n) >> (log2 + 1)
        p = np.full((n, n), dc, dtype=np.int64)
        if c_idx == 0 and n < 32:
            p[0, :] = (top[1:n + 1] + 3 * dc + 2) >> 2
            p[:, 0] = (left[1:n + 1] + 3 * dc + 2) >> 2
            p[0, 0] = (left[1] + 2 * dc + top[1] + 2) >> 2
        return p
    ang = ANGLES[mode]
    vertical = mode >= 18
    main, side = (top, left) if vertical else (left, top)
    ref = {k: int(main[k]) for k in range(0, n + 1)}
    if ang < 0:
        last = (n * ang) >> 5
        if last < -1:
            inv = inv_angle(ang)
            for k in range(last, 0):
                ref[k] = int(side[(k * inv + 128) >> 8])
    else:
        for k in range(n + 1, 2 * n + 1):
            ref[k] = int(main[k])
    p = np.zeros((n, n), dtype=np.int64)
    for j in range(n):                             # distance from the main reference
        idx, fact = ((j + 1) * ang) >> 5, ((j + 1) * ang) & 31
        for i in range(n):
            v = ((32 - fact) * ref[i + idx + 1] + fact * ref[i + idx + 2] + 16) >> 5 if fact else ref[i + idx + 1]
            if vertical:
                p[j, i] = v
            else:
                p[i, j] = v
    if c_idx == 0 and n < 32:
        if mode == 26:
            p[:, 0] = np.clip(top[1] + ((left[1:n + 1] - left[0]) >> 1), 0, 255)
        if mode == 10:
            p[0, :] = np.clip(left[1] + ((top[1:n + 1] - top[0]) >> 1), 0, 255)
    return p


# ---------------------------------------------------------------------------------------------------- deblocking
def beta_prime(q):
    return 0 if q < 16 else q - 10 if q <= 28 else 2 * q - 38       # Table 8-11, as its three straight pieces


TC_STEPS = [(18, 1), (27, 2), (31, 3), (35, 4), (38, 5), (40, 6), (42, 7), (43, 8), (44, 9), (45, 10), (46, 11), (47, 13), (48, 14),
            (49, 16), (50, 18), (51, 20), (52, 22), (53, 24)]


def tc_prime(q):
    v = 0
    for start, val in TC_STEPS:
        if q >= start:
            v = val
    return v


def deblock_luma(blk, bs, qp_p, qp_q, beta_off_div2=0, tc_off_div2=0, no_p=False, no_q=False):
    """8.7.2.5.3 + 8.7.2.5.6 / .7 for one 4-line segment: blk[line][0..7] = p3 p2 p1 p0 q0 q1 q2 q3."""
    b = np.array(blk, dtype=np.int64)
    qpl = (qp_p + qp_q + 1) >> 1
    beta = beta_prime(min(max(qpl + 2 * beta_off_div2, 0), 51))
    tc = tc_prime(min(max(qpl + 2 * (bs - 1) + 2 * tc_off_div2, 0), 53))
    p = lambda k, l: b[l, 3 - k]
    q = lambda k, l: b[l, 4 + k]
    d2 = lambda f, l: abs(f(2, l) - 2 * f(1, l) + f(0, l))
    dp0, dp3, dq0, dq3 = d2(p, 0), d2(p, 3), d2(q, 0), d2(q, 3)
    if dp0 + dq0 + dp3 + dq3 >= beta:
        return b
    def strong_line(l, dpq):
        return 2 * dpq < (beta >> 2) and abs(p(3, l) - p(0, l)) + abs(q(0, l) - q(3, l)) < (beta >> 3) and abs(p(0, l) - q(0, l)) < ((5 * tc + 1) >> 1)
    strong = strong_line(0, dp0 + dq0) and strong_line(3, dp3 + dq3)
    side = (beta + (beta >> 1)) >> 3
    dep, deq = (dp0 + dp3) < side, (dq0 + dq3) < side
    out = b.copy()
    for l in range(4):
        p0, p1, p2, p3 = (int(p(k, l)) for k in range(4))
        q0, q1, q2, q3 = (int(q(k, l)) for k in range(4))
        if strong:
            np_ = [min(max((p2 + 2 * p1 + 2 * p0 + 2 * q0 + q1 + 4) >> 3, p0 - 2 * tc), p0 + 2 * tc),
                   min(max((p2 + p1 + p0 + q0 + 2) >> 2, p1 - 2 * tc), p1 + 2 * tc),
                   min(max((2 * p3 + 3 * p2 + p1 + p0 + q0 + 4) >> 3, p2 - 2 * tc), p2 + 2 * tc)]
            nq = [min(max((p1 + 2 * p0 + 2 * q0 + 2 * q1 + q2 + 4) >> 3, q0 - 2 * tc), q0 + 2 * tc),
                  min(max((p0 + q0 + q1 + q2 + 2) >> 2, q1 - 2 * tc), q1 + 2 * tc),
                  min(max((p0 + q0 + q1 + 3 * q2 + 2 * q3 + 4) >> 3, q2 - 2 * tc), q2 + 2 * tc)]
            if not no_p:
                out[l, 3], out[l, 2], out[l, 1] = np_
            if not no_q:
                out[l, 4], out[l, 5], out[l, 6] = nq
        else:
            delta = (9 * (q0 - p0) - 3 * (q1 - p1) + 8) >> 4
            if abs(delta) >= 10 * tc:
                continue
            delta = min(max(delta, -tc), tc)
            if not no_p:
                out[l, 3] = min(max(p0 + delta, 0), 255)
                if dep:
                    out[l, 2] = min(max(p1 + min(max((((p2 + p0 + 1) >> 1) - p1 + delta) >> 1, -(tc >> 1)), tc >> 1), 0), 255)
            if not no_q:
                out[l, 4] = min(max(q0 - delta, 0), 255)
                if deq:
                    out[l, 5] = min(max(q1 + min(max((((q2 + q0 + 1) >> 1) - q1 - delta) >> 1, -(tc >> 1)), tc >> 1), 0), 255)
    return out


def qpc_of(qpi):
    if qpi < 30:
        return qpi
    if qpi >= 43:
        return qpi - 6
    return [29, 30, 31, 32, 33, 33, 34, 34, 35, 35, 36, 36, 37][qpi - 30]


def deblock_chroma(row, qp_p, qp_q, c_qp_pic_offset=0, tc_off_div2=0):
    p1, p0, q0, q1 = (int(v) for v in row)
    qpc = qpc_of(((qp_p + qp_q + 1) >> 1) + c_qp_pic_offset)
    tc = tc_prime(min(max(qpc + 2 + 2 * tc_off_div2, 0), 53))
    delta = min(max((((q0 - p0) << 2) + p1 - q1 + 4) >> 3, -tc), tc)
    return [p1, min(max(p0 + delta, 0), 255), min(max(q0 - delta, 0), 255), q1]


# ----------------------------------------------------------------------------------------------------------- SAO
def sao(block, typ, band_pos, eo_class, offsets):
    """8.7.3 on a block that is the whole picture: samples with a neighbour outside keep their value."""
    a = np.asarray(block, dtype=np.int64)
    out = a.copy()
    off = np.array([0] + list(offsets), dtype=np.int64)
    if typ == 1:
        k = ((a >> 3) - band_pos) & 31
        idx = np.where(k < 4, k + 1, 0)
        return np.clip(a + off[idx], 0, 255)
    if typ == 2:
        dx, dy = {0: (1, 0), 1: (0, 1), 2: (1, 1), 3: (-1, 1)}[eo_class]   # the second neighbour; the first is its mirror image
        h, w = a.shape
        ys, xs = np.mgrid[0:h, 0:w]
        ok = (xs - dx >= 0) & (xs - dx < w) & (xs + dx >= 0) & (xs + dx < w) & (ys - dy >= 0) & (ys + dy < h)
        ya, xa = np.clip(ys - dy, 0, h - 1), np.clip(xs - dx, 0, w - 1)
        yb, xb = np.clip(ys + dy, 0, h - 1), np.clip(xs + dx, 0, w - 1)
        s = np.sign(a - a[ya, xa]) + np.sign(a - a[yb, xb])           # -2 valley, -1, 0, 1, 2 peak
        cat = np.select([s == -2, s == -1, s == 1, s == 2], [1, 2, 3, 4], 0)
        return np.where(ok, np.clip(a + off[cat], 0, 255), a)
    return out

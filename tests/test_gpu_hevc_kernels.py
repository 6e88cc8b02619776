"""-m gpu: the HEVC kernels (hwang_amd/csrc/amd/k_hevc.hip) against the numpy statements of tests/hevc_numpy.py, fed
with records built here - no stream, no parser and no oracle in between (hwang_hevc_run_test_picture, include/
hwang_amd.h).  Residual blocks of every size / QP / transform, all 35 intra modes with every availability pattern the
substitution process distinguishes, every fractional position of both interpolation filters with vectors far outside
the picture, default / explicit weights, the luma and chroma deblocking filters over random edge maps, SAO band and
all four edge classes."""
import ctypes as C

import numpy as np
import pytest

from tests import hevc_numpy as hn

pytestmark = pytest.mark.gpu

RES = np.dtype([("x", "<u2"), ("y", "<u2"), ("log2", "u1"), ("c_idx", "u1"), ("qp", "u1"), ("flags", "u1"), ("matrix", "u1"), ("pad", "u1", 3),
                ("coef_offset", "<u4"), ("ncoef", "<u4")])
PU = np.dtype([("x", "<u2"), ("y", "<u2"), ("w", "u1"), ("h", "u1"), ("pred_flag", "u1"), ("weighted", "u1"), ("ref_slot", "u1", 2),
               ("mv", "<i2", (2, 2)), ("wp", "<u2", 2), ("pad", "<u2")])
WT = np.dtype([("w", "<i2", 3), ("o", "<i2", 3), ("log2wd", "u1", 2), ("pad", "u1", 2)])
ITU = np.dtype([("x", "<u2"), ("y", "<u2"), ("log2", "u1"), ("mode", "u1"), ("flags", "u1"), ("pad", "u1"), ("avail_left", "<u4"), ("avail_top", "<u4"),
                ("wait_left", "<u4"), ("wait_top", "<u4")])
SAO = np.dtype([("type", "u1", 3), ("band_pos", "u1", 3), ("eo_class", "u1", 3), ("offset", "i1", (3, 4)), ("pad", "u1", 3)])
SLICE = np.dtype([("beta", "i1"), ("tc", "i1"), ("across", "u1"), ("sao_luma", "u1"), ("sao_chroma", "u1"), ("pad", "u1", 3), ("first_ctb", "<u4")])
EDGE = np.dtype([("edge", "u1"), ("qp", "i1"), ("flags", "u1"), ("slice", "u1")])
E_TU_LEFT, E_PU_LEFT, E_TU_TOP, E_PU_TOP, E_ON_LEFT, E_ON_TOP, E_LEVELS, E_INTRA = 1, 2, 4, 8, 16, 32, 64, 128   # hevc_frame_job.h: EDGE_*
assert (RES.itemsize, PU.itemsize, WT.itemsize, ITU.itemsize, SAO.itemsize, SLICE.itemsize, EDGE.itemsize) == (20, 24, 16, 24, 24, 12, 4)


class PictureArgs(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("log2_ctb", C.c_int32), ("cb_qp_offset", C.c_int32), ("cr_qp_offset", C.c_int32),
                ("sao_enabled", C.c_int32), ("deblock_enabled", C.c_int32), ("intra_rows", C.c_int32), ("reserved", C.c_int32), ("recon_in", C.c_void_p), ("refs", C.c_void_p), ("num_refs", C.c_int32),
                ("num_res", C.c_int32), ("num_coefs", C.c_int32), ("num_pus", C.c_int32), ("num_weights", C.c_int32), ("num_intra", C.c_int32 * 3),
                ("num_slices", C.c_int32), ("res", C.c_void_p), ("coefs", C.c_void_p), ("pus", C.c_void_p), ("weights", C.c_void_p),
                ("intra", C.c_void_p * 3), ("edges", C.c_void_p), ("sao", C.c_void_p), ("slices", C.c_void_p),
                ("scaling", C.c_void_p), ("out", C.c_void_p), ("resid_out", C.c_void_p)]


def run_picture(hw, w, h, log2_ctb=4, recon=None, refs=(), res=None, coefs=None, pus=None, weights=None, intra=(None, None, None),
                edges=None, sao=None, slices=None, cb_off=0, cr_off=0, deblock=False, want_resid=False, intra_rows=False):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; there is no fallback"
    from hwang_amd._capi import lib
    ctbs = ((w + (1 << log2_ctb) - 1) >> log2_ctb) * ((h + (1 << log2_ctb) - 1) >> log2_ctb)
    wctb = (w + (1 << log2_ctb) - 1) >> log2_ctb
    keep = []

    def arr(a, dtype, n=0):
        a = np.zeros(max(n, 1), dtype=dtype) if a is None else np.ascontiguousarray(a, dtype=dtype)
        keep.append(a)
        return a
    t = PictureArgs()
    t.width, t.height, t.log2_ctb, t.cb_qp_offset, t.cr_qp_offset = w, h, log2_ctb, cb_off, cr_off
    t.sao_enabled, t.deblock_enabled, t.intra_rows = int(sao is not None), int(deblock), int(intra_rows)
    fs = w * h * 3 // 2
    rin = arr(recon if recon is not None else np.full(fs, 128), np.uint8)
    t.recon_in = rin.ctypes.data
    rf = arr(np.concatenate([np.asarray(r, dtype=np.uint8).ravel() for r in refs]) if refs else None, np.uint8)
    t.refs, t.num_refs = rf.ctypes.data, len(refs)
    r_, c_, p_, w_ = arr(res, RES), arr(coefs, np.uint32), arr(pus, PU), arr(weights, WT)
    t.res, t.num_res = r_.ctypes.data, 0 if res is None else len(r_)
    t.coefs, t.num_coefs = c_.ctypes.data, 0 if coefs is None else len(c_)
    t.pus, t.num_pus = p_.ctypes.data, 0 if pus is None else len(p_)
    t.weights, t.num_weights = w_.ctypes.data, 0 if weights is None else len(w_)
    for c in range(3):
        it = arr(intra[c], ITU)
        n = 0 if intra[c] is None else len(it)
        if n:       # both intra kernels want the list grouped by coding tree block, in raster order
            sub = 1 if c else 0
            ctb_of = ((it["y"].astype(int) << sub) >> log2_ctb) * wctb + ((it["x"].astype(int) << sub) >> log2_ctb)
            it = arr(it[np.argsort(ctb_of, kind="stable")], ITU)
        t.intra[c], t.num_intra[c] = it.ctypes.data, n
    e_ = arr(edges, EDGE, (w // 4) * (h // 4))
    s_ = arr(sao, SAO, ctbs)
    sl = arr(slices if slices is not None else np.array([(0, 0, 1, 1, 1, [0, 0, 0], 0)], dtype=SLICE), SLICE)
    t.edges, t.sao, t.slices, t.num_slices = e_.ctypes.data, s_.ctypes.data, sl.ctypes.data, len(sl)
    out = np.zeros(fs, dtype=np.uint8)
    resid = np.zeros(fs, dtype=np.int16)
    t.out = out.ctypes.data
    t.resid_out = resid.ctypes.data if want_resid else None
    err = C.create_string_buffer(256)
    rc = lib.hwang_hevc_run_test_picture(C.byref(t), err, 256)
    assert rc == 0, err.value.decode()
    planes = (out[:w * h].reshape(h, w), out[w * h:w * h + w * h // 4].reshape(h // 2, w // 2), out[w * h + w * h // 4:].reshape(h // 2, w // 2))
    rplanes = (resid[:w * h].reshape(h, w), resid[w * h:w * h + w * h // 4].reshape(h // 2, w // 2), resid[w * h + w * h // 4:].reshape(h // 2, w // 2))
    return planes, rplanes


def test_residual_kernel(hw):
    rnd = np.random.default_rng(21)
    W, H = 128, 64
    for rep in range(6):
        res, coefs, want = [], [], [np.zeros((H, W), dtype=np.int64), np.zeros((H // 2, W // 2), dtype=np.int64), np.zeros((H // 2, W // 2), dtype=np.int64)]
        # a grid of 32x32 cells, one block of some size and plane in each
        for cy in range(0, H, 32):
            for cx in range(0, W, 32):
                log2 = int(rnd.integers(2, 6))
                c_idx = int(rnd.integers(0, 3))
                if c_idx and log2 == 5:
                    log2 = 4
                n = 1 << log2
                x, y = (cx >> (1 if c_idx else 0)), (cy >> (1 if c_idx else 0))
                qp, intra = int(rnd.integers(0, 52)), int(rnd.integers(0, 2))
                tskip = bool(log2 == 2 and rnd.integers(0, 3) == 0)
                bypass = bool(rnd.integers(0, 7) == 0)
                lev = np.zeros((n, n), dtype=np.int64)
                k = int(rnd.integers(1, min(n * n, 40) + 1))
                lev[rnd.integers(0, n, k), rnd.integers(0, n, k)] = rnd.integers(-32768 if rep == 0 else -60, 32768 if rep == 0 else 61, k)
                pos = np.flatnonzero(lev)
                flags = (1 if (intra and c_idx == 0 and log2 == 2) else 0) | (2 if tskip else 0) | (4 if bypass else 0)
                res.append((x, y, log2, c_idx, qp, flags, 0xff, [0, 0, 0], len(coefs), len(pos)))
                coefs += [int(p) << 16 | (int(lev.ravel()[p]) & 0xffff) for p in pos]
                r = lev if bypass else hn.residual(lev, qp, intra, c_idx, transform_skip=tskip)
                want[c_idx][y:y + n, x:x + n] = np.asarray(r).astype(np.int16)
        _, got = run_picture(hw, W, H, res=np.array(res, dtype=RES), coefs=np.array(coefs, dtype=np.uint32), want_resid=True)
        for c in range(3):
            assert np.array_equal(got[c].astype(np.int64), want[c]), (rep, c)


def substitute(avail, vals, n):
    """8.4.4.2.2 on one line of 4n + 1 samples: bottom-left ... corner ... top-right."""
    if not avail.any():
        return np.full(4 * n + 1, 128, dtype=np.int64)
    v = vals.astype(np.int64).copy()
    a = avail.copy()
    if not a[0]:
        v[0] = v[np.flatnonzero(a)[0]]
        a[0] = True
    for i in range(1, 4 * n + 1):
        if not a[i]:
            v[i] = v[i - 1]
    return v


@pytest.mark.parametrize("log2", [2, 3, 4, 5])
def test_intra_kernel(hw, log2):
    rnd = np.random.default_rng(40 + log2)
    n = 1 << log2
    cell = 8 * n            # every block sits inside a cell of its own (chroma: half of it), its neighbours are what recon_in holds
    cols, rows = 6, 6
    W, H = cols * cell, rows * cell
    for c_idx in ((0, 1) if log2 < 5 else (0,)):     # (4:2:0: the largest chroma transform block is 16x16)
        sub = 1 if c_idx else 0
        pw, ph = W >> sub, H >> sub
        smooth = rnd.integers(0, 2)
        planes = [rnd.integers(0, 256, (H, W)), rnd.integers(0, 256, (H // 2, W // 2)), rnd.integers(0, 256, (H // 2, W // 2))]
        if smooth:
            planes[c_idx] = np.clip(90 + (np.add.outer(np.arange(ph), np.arange(pw)) // 6) + rnd.integers(0, 2, (ph, pw)), 0, 255)
        recon = np.concatenate([p.astype(np.uint8).ravel() for p in planes])
        unit = 2 if c_idx else 4
        S = 64 >> sub
        for rows_kernel in (False, True):
            recs, expect = [], []
            mode = 0
            for ry in range(rows):
                for rx in range(cols):
                    x, y = (rx * cell >> sub) + n, (ry * cell >> sub) + n
                    pat = int(rnd.integers(0, 6))
                    nu = 2 * n // unit
                    al = int(rnd.integers(0, 1 << nu)) if pat > 1 else ((1 << nu) - 1 if pat == 0 else 0)
                    at = int(rnd.integers(0, 1 << nu)) if pat > 1 else ((1 << nu) - 1 if pat == 0 else 0)
                    corner = int(rnd.integers(0, 2)) if pat > 1 else (1 if pat == 0 else 0)
                    if rows_kernel:
                        # k_hevc_intra_rows keeps one coding tree block and its rim in LDS: what a stream can mark available
                        # (6.4.1: nothing below the block's row, nothing right of it except in the row above) is all it holds
                        cx, cy = x // S * S, y // S * S
                        for k in range(nu):
                            if y + k * unit >= cy + S:
                                al &= ~(1 << k)
                            if x + k * unit >= cx + (S if y > cy else 2 * S):
                                at &= ~(1 << k)
                    strong = int(rnd.integers(0, 2))
                    recs.append((x, y, log2, mode, corner | strong << 1, 0, al, at, 0, 0))     # nothing to wait for: the neighbours are recon_in's
                    p = planes[c_idx]
                    left_vals = np.array([p[y + k, x - 1] for k in range(2 * n)])
                    top_vals = np.array([p[y - 1, x + k] for k in range(2 * n)])
                    la = np.array([(al >> (k // unit)) & 1 for k in range(2 * n)], dtype=bool)
                    ta = np.array([(at >> (k // unit)) & 1 for k in range(2 * n)], dtype=bool)
                    line_v = np.concatenate([left_vals[::-1], [p[y - 1, x - 1]], top_vals])
                    line_a = np.concatenate([la[::-1], [bool(corner)], ta])
                    line = substitute(line_a, line_v, n)
                    left = np.concatenate([[line[2 * n]], line[:2 * n][::-1]])
                    top = line[2 * n:]
                    expect.append((x, y, hn.intra_predict(left, top, n, c_idx, mode, bool(strong))))
                    mode = (mode + 1) % 35
            intra = [None, None, None]
            intra[c_idx] = np.array(recs, dtype=ITU)
            got, _ = run_picture(hw, W, H, log2_ctb=6, recon=recon, intra=tuple(intra), intra_rows=rows_kernel)
            for (x, y, want) in expect:
                assert np.array_equal(got[c_idx][y:y + n, x:x + n].astype(np.int64), want), (c_idx, x, y, rows_kernel)


def test_inter_kernel(hw):
    rnd = np.random.default_rng(50)
    W, H = 256, 192
    refs = [np.concatenate([rnd.integers(0, 256, W * H), rnd.integers(0, 256, W * H // 2)]).astype(np.uint8) for _ in range(2)]
    refs[0][:W * 3] = 255
    refs[1][-W // 2 * 4:] = 0
    rp = [(r[:W * H].reshape(H, W), r[W * H:W * H + W * H // 4].reshape(H // 2, W // 2), r[W * H + W * H // 4:].reshape(H // 2, W // 2)) for r in refs]
    weights = np.array([([64, 70, 60], [3, -2, 1], [6, 6], [0, 0]), ([61, 64, 66], [-4, 2, 0], [6, 6], [0, 0]),
                        ([1, 1, 1], [0, 5, -5], [0, 0], [0, 0]), ([130, 100, 90], [10, -10, 7], [7, 5], [0, 0])], dtype=WT)
    pus, expect = [], []
    sizes = [(8, 8), (16, 8), (8, 16), (16, 16), (8, 4), (4, 8), (32, 32), (16, 4), (12, 16), (64, 32)]
    k = 0
    for y in range(0, H, 32):
        for x in range(0, W, 64):
            w, h = sizes[k % len(sizes)]
            pf = [1, 2, 3][k % 3] if w + h != 12 else [1, 2][k % 2]
            mv = [[int(rnd.integers(-40, 41)), int(rnd.integers(-40, 41))], [int(rnd.integers(-700, 700)), int(rnd.integers(-500, 500))]]
            if k % 4 == 0:
                mv[0] = [(k // 4 % 4) + 4 * int(rnd.integers(-3, 4)), (k // 16 % 4) + 4 * int(rnd.integers(-3, 4))]
            weighted = int(k % 5 in (1, 3))
            wp = [int(rnd.integers(0, 4)), int(rnd.integers(0, 4))]     # (the denominators are those of the first list in use, as in the kernel)
            pus.append((x, y, w, h, pf, weighted, [0, 1], mv, wp, 0))
            for c in range(3):
                sub = 1 if c else 0
                p = [hn.pred_samples(rp[l][c], x >> sub, y >> sub, w >> sub, h >> sub, mv[l][0], mv[l][1], chroma=bool(c)) if (pf >> l) & 1 else None for l in range(2)]
                if not weighted:
                    v = hn.inter_bi(p[0], p[1]) if pf == 3 else np.clip(((p[0] if pf == 1 else p[1]) + 32) >> 6, 0, 255)
                else:
                    w0, w1 = weights[wp[0]], weights[wp[1]]
                    ld = int((w0 if pf & 1 else w1)["log2wd"][1 if c else 0])
                    if pf == 3:
                        v = hn.inter_weighted_bi(p[0], p[1], int(w0["w"][c]), int(w1["w"][c]), int(w0["o"][c]), int(w1["o"][c]), ld)
                    elif pf == 1:
                        v = hn.inter_weighted_uni(p[0], int(w0["w"][c]), int(w0["o"][c]), ld)
                    else:
                        v = hn.inter_weighted_uni(p[1], int(w1["w"][c]), int(w1["o"][c]), ld)
                expect.append((c, x >> sub, y >> sub, v))
            k += 1
    got, _ = run_picture(hw, W, H, refs=refs, pus=np.array(pus, dtype=PU), weights=weights)
    for (c, x, y, v) in expect:
        assert np.array_equal(got[c][y:y + v.shape[0], x:x + v.shape[1]].astype(np.int64), v), (c, x, y)


def test_all_fractional_positions(hw):
    rnd = np.random.default_rng(51)
    W, H = 256, 128
    ref = np.concatenate([rnd.integers(0, 256, W * H), rnd.integers(0, 256, W * H // 2)]).astype(np.uint8)
    planes = (ref[:W * H].reshape(H, W), ref[W * H:W * H + W * H // 4].reshape(H // 2, W // 2), ref[W * H + W * H // 4:].reshape(H // 2, W // 2))
    pus, expect = [], []
    for yf in range(8):
        for xf in range(8):
            x, y = xf * 32, yf * 16
            mv = [xf + 8 * int(rnd.integers(-4, 5)), yf + 8 * int(rnd.integers(-4, 5))]    # luma fractions (mv & 3), chroma fractions (mv & 7): all 64
            pus.append((x, y, 16, 16, 1, 0, [0, 0], [mv, [0, 0]], [0, 0], 0))
            for c in range(3):
                sub = 1 if c else 0
                expect.append((c, x >> sub, y >> sub, hn.inter_default(planes[c], x >> sub, y >> sub, 16 >> sub, 16 >> sub, mv[0], mv[1], chroma=bool(c))))
    got, _ = run_picture(hw, W, H, refs=[ref], pus=np.array(pus, dtype=PU))
    for (c, x, y, v) in expect:
        assert np.array_equal(got[c][y:y + v.shape[0], x:x + v.shape[1]].astype(np.int64), v), (c, x, y)


def test_deblocking_kernel(hw):
    rnd = np.random.default_rng(60)
    W, H = 128, 64
    w4, h4 = W // 4, H // 4
    for rep in range(4):
        base = rnd.integers(40, 200, (H // 8, W // 8))
        y = np.clip(np.kron(base, np.ones((8, 8), dtype=int)) + rnd.integers(-3, 4, (H, W)), 0, 255)      # 8x8 blocks of different levels: edges to smooth
        u = np.clip(np.kron(rnd.integers(60, 190, (H // 16, W // 16)), np.ones((8, 8), dtype=int)) + rnd.integers(-2, 3, (H // 2, W // 2)), 0, 255)
        v = np.clip(np.kron(rnd.integers(60, 190, (H // 16, W // 16)), np.ones((8, 8), dtype=int)) + rnd.integers(-2, 3, (H // 2, W // 2)), 0, 255)
        edges = np.zeros((h4, w4), dtype=EDGE)
        edges["qp"] = np.kron(rnd.integers(15, 50, (h4 // 2, w4 // 2)), np.ones((2, 2), dtype=int))
        edges["flags"] = (rnd.integers(0, 12, (h4, w4)) == 0)
        # what the host says about each 4x4 block (per 8x8 block: intra or not, levels or not; per side: a transform block
        # edge or only a prediction block edge, filtered or not) - the kernel derives the strengths (8.7.2.4), and so
        # does this test: 2 beside an intra block, 1 at a transform block edge with levels on either side, else 0 (no
        # prediction block records here: both sides are "the same block")
        intra = np.kron(rnd.integers(0, 10, (h4 // 2, w4 // 2)) < 3, np.ones((2, 2), dtype=int)).astype(bool)
        levels = np.kron(rnd.integers(0, 2, (h4 // 2, w4 // 2)), np.ones((2, 2), dtype=int)).astype(bool)
        on_l = (rnd.integers(0, 10, (h4, w4)) < 8) & (np.arange(w4)[None, :] > 0)
        on_t = (rnd.integers(0, 10, (h4, w4)) < 8) & (np.arange(h4)[:, None] > 0)
        tu_l, tu_t = rnd.integers(0, 10, (h4, w4)) < 7, rnd.integers(0, 10, (h4, w4)) < 7
        edges["edge"] = (on_l * E_ON_LEFT | on_t * E_ON_TOP | tu_l * E_TU_LEFT | ~tu_l * E_PU_LEFT | tu_t * E_TU_TOP | ~tu_t * E_PU_TOP |
                         levels * E_LEVELS | intra * E_INTRA).astype(np.uint8)
        bsv, bsh = np.zeros((h4, w4), dtype=int), np.zeros((h4, w4), dtype=int)
        for by in range(h4):
            for bx in range(w4):
                if on_l[by, bx] and bx % 2 == 0:
                    bsv[by, bx] = 2 if intra[by, bx] or intra[by, bx - 1] else 1 if tu_l[by, bx] and (levels[by, bx] or levels[by, bx - 1]) else 0
                if on_t[by, bx] and by % 2 == 0:
                    bsh[by, bx] = 2 if intra[by, bx] or intra[by - 1, bx] else 1 if tu_t[by, bx] and (levels[by, bx] or levels[by - 1, bx]) else 0
        assert (bsv == 1).any() and (bsv == 2).any() and (bsh == 1).any() and (bsh == 2).any()
        slices = np.array([(2 * int(rnd.integers(-3, 4)), 2 * int(rnd.integers(-3, 4)), 1, 1, 1, [0, 0, 0], 0)], dtype=SLICE)
        cb_off, cr_off = int(rnd.integers(-6, 7)), int(rnd.integers(-6, 7))
        # numpy: all vertical edges, then all horizontal ones (8.7.2)
        wy, wu, wv = y.copy(), u.copy(), v.copy()
        for d in range(2):
            for by in range(h4):
                for bx in range(w4):
                    bs = int((bsh if d else bsv)[by, bx])
                    if not bs:
                        continue
                    p = edges[by - 1, bx] if d else edges[by, bx - 1]
                    q = edges[by, bx]
                    if d:
                        blk = wy[by * 4 - 4:by * 4 + 4, bx * 4:bx * 4 + 4].T
                    else:
                        blk = wy[by * 4:by * 4 + 4, bx * 4 - 4:bx * 4 + 4]
                    out = hn.deblock_luma(blk, bs, int(p["qp"]), int(q["qp"]), int(slices[0]["beta"]) // 2, int(slices[0]["tc"]) // 2, bool(p["flags"]), bool(q["flags"]))
                    if d:
                        wy[by * 4 - 4:by * 4 + 4, bx * 4:bx * 4 + 4] = out.T
                    else:
                        wy[by * 4:by * 4 + 4, bx * 4 - 4:bx * 4 + 4] = out
                    if bs == 2 and (by if d else bx) % 4 == 0:
                        for plane, off in ((wu, cb_off), (wv, cr_off)):
                            for l in range(2):
                                if d:
                                    row = plane[by * 2 - 2:by * 2 + 2, bx * 2 + l]
                                else:
                                    row = plane[by * 2 + l, bx * 2 - 2:bx * 2 + 2]
                                r = hn.deblock_chroma(row, int(p["qp"]), int(q["qp"]), off, int(slices[0]["tc"]) // 2)
                                if p["flags"]:
                                    r[1] = int(row[1])
                                if q["flags"]:
                                    r[2] = int(row[2])
                                if d:
                                    plane[by * 2 - 2:by * 2 + 2, bx * 2 + l] = r
                                else:
                                    plane[by * 2 + l, bx * 2 - 2:bx * 2 + 2] = r
        recon = np.concatenate([y.ravel(), u.ravel(), v.ravel()]).astype(np.uint8)
        got, _ = run_picture(hw, W, H, recon=recon, edges=edges.ravel(), slices=slices, cb_off=cb_off, cr_off=cr_off, deblock=True)
        assert np.array_equal(got[0].astype(np.int64), wy), rep
        assert np.array_equal(got[1].astype(np.int64), wu) and np.array_equal(got[2].astype(np.int64), wv), rep
        assert (wy != y).any() and (wu != u).any()


def test_deblocking_motion_rule(hw):
    """Prediction block edges with no levels and nothing intra beside them: k_hevc_deblock decides them from the two
    prediction blocks' motion (8.7.2.4's last rule).  A
    picture of 8x8 prediction blocks over two flat reference pictures (a block's prediction is a constant whatever its
    vector: 100, 140 or their mean 120), vectors / references / lists drawn from a few values so that every case of the
    rule occurs; the expected picture is numpy's filter over the strengths this test derives from the rule's text."""
    rnd = np.random.default_rng(61)
    W, H = 128, 64
    w4, h4 = W // 4, H // 4
    flat = [np.concatenate([np.full(W * H, 100), np.full(W * H // 2, 90)]).astype(np.uint8),
            np.concatenate([np.full(W * H, 140), np.full(W * H // 2, 160)]).astype(np.uint8)]
    level = {(0,): 100, (1,): 140, (0, 0): 100, (1, 1): 140, (0, 1): 120, (1, 0): 120}

    def strength(a, b):
        """a, b: lists of (reference picture, vector) of the lists in use.  1 unless the same pictures are referred to by
        the same number of vectors and every vector has a partner within a sample (either pairing when both refer to
        one picture)."""
        if len(a) != len(b) or sorted(r for r, _ in a) != sorted(r for r, _ in b):
            return 1
        near = lambda u, v: abs(u[0] - v[0]) < 4 and abs(u[1] - v[1]) < 4
        if len(a) == 1:
            return 0 if near(a[0][1], b[0][1]) else 1
        pairings = []
        if a[0][0] == b[0][0] and a[1][0] == b[1][0]:
            pairings.append(near(a[0][1], b[0][1]) and near(a[1][1], b[1][1]))
        if a[0][0] == b[1][0] and a[1][0] == b[0][0]:
            pairings.append(near(a[0][1], b[1][1]) and near(a[1][1], b[0][1]))
        return 0 if any(pairings) else 1

    for rep in range(3):
        pus, motion = [], {}
        for by in range(H // 8):
            for bx in range(W // 8):
                kind = int(rnd.integers(0, 4))        # list 0, list 1, both, both into one picture
                refs = [(int(rnd.integers(0, 2)),), (int(rnd.integers(0, 2)),), (0, 1) if rnd.integers(0, 2) else (1, 0), (int(rnd.integers(0, 2)),) * 2][kind]
                mvs = [(int(rnd.choice([0, 2, 4, 9])), int(rnd.choice([0, 3, -5]))) for _ in refs]
                pred_flag = [1, 2, 3, 3][kind]
                rs, mv = [0, 0], [[0, 0], [0, 0]]
                lists = [0] if kind == 0 else [1] if kind == 1 else [0, 1]
                for l, r, m in zip(lists, refs, mvs):
                    rs[l], mv[l] = r, list(m)
                pus.append((bx * 8, by * 8, 8, 8, pred_flag, 0, rs, mv, [0, 0], 0))
                motion[(bx, by)] = (list(zip(refs, mvs)), level[refs])
        y = np.zeros((H, W), dtype=np.int64)
        for (bx, by), (_, lv) in motion.items():
            y[by * 8:by * 8 + 8, bx * 8:bx * 8 + 8] = lv
        edges = np.zeros((h4, w4), dtype=EDGE)
        edges["qp"] = 38
        bsv = np.where((np.arange(w4)[None, :] % 2 == 0) & (np.arange(w4)[None, :] > 0), 1, 0) * np.ones((h4, 1), dtype=int)   # prediction block edges
        bsh = np.where((np.arange(h4)[:, None] % 2 == 0) & (np.arange(h4)[:, None] > 0), 1, 0) * np.ones((1, w4), dtype=int)
        edges["edge"] = (bsv * (E_PU_LEFT | E_ON_LEFT) | bsh * (E_PU_TOP | E_ON_TOP)).astype(np.uint8)   # no levels, nothing intra: the motion decides
        slices = np.array([(0, 0, 1, 1, 1, [0, 0, 0], 0)], dtype=SLICE)
        want = y.copy()
        filtered = 0
        for d in range(2):
            for by in range(h4):
                for bx in range(w4):
                    if not int((bsh if d else bsv)[by, bx]):
                        continue
                    q = motion[(bx // 2, by // 2)][0]
                    p = motion[(bx // 2, (by - 1) // 2)][0] if d else motion[((bx - 1) // 2, by // 2)][0]
                    if not strength(p, q):
                        continue
                    filtered += 1
                    if d:
                        blk = want[by * 4 - 4:by * 4 + 4, bx * 4:bx * 4 + 4].T
                    else:
                        blk = want[by * 4:by * 4 + 4, bx * 4 - 4:bx * 4 + 4]
                    out = hn.deblock_luma(blk, 1, 38, 38, 0, 0, False, False)
                    if d:
                        want[by * 4 - 4:by * 4 + 4, bx * 4:bx * 4 + 4] = out.T
                    else:
                        want[by * 4:by * 4 + 4, bx * 4 - 4:bx * 4 + 4] = out
        got, _ = run_picture(hw, W, H, refs=flat, pus=np.array(pus, dtype=PU), edges=edges.ravel(), slices=slices, deblock=True)
        assert np.array_equal(got[0].astype(np.int64), want), rep
        assert filtered > 50 and (want != y).any()


def test_sao_kernel(hw):
    rnd = np.random.default_rng(70)
    W, H = 64, 48
    for typ in (1, 2):
        for cls in range(4 if typ == 2 else 1):
            y = rnd.integers(0, 256, (H, W))
            u = np.clip(128 + rnd.integers(-4, 5, (H // 2, W // 2)), 0, 255)
            v = rnd.integers(0, 256, (H // 2, W // 2))
            sao = np.zeros(1, dtype=SAO)      # one 64x64 coding tree block covers the picture
            sao["type"] = typ
            sao["eo_class"] = cls
            sao["band_pos"] = [int(rnd.integers(0, 32)) for _ in range(3)]
            sao["offset"] = rnd.integers(-7, 8, (3, 4))
            recon = np.concatenate([y.ravel(), u.ravel(), v.ravel()]).astype(np.uint8)
            got, _ = run_picture(hw, W, H, log2_ctb=6, recon=recon, sao=sao)
            for c, plane in enumerate((y, u, v)):
                want = hn.sao(plane, typ, int(sao["band_pos"][0][c]), cls, sao["offset"][0][c])
                assert np.array_equal(got[c].astype(np.int64), want), (typ, cls, c)


def test_intra_dataflow_waits_for_its_neighbours(hw):
    """A picture of 8x8 luma blocks that all lean on each other (left, top-left, top, top-right of every block are intra
    blocks of the same list): the kernel's tickets and "done" marks must reproduce the sequential result, whatever order
    its wavefronts run in.  Raster order of the blocks is a valid decoding order for these dependences."""
    rnd = np.random.default_rng(90)
    W, H, n = 256, 128, 8
    bw, bh = W // n, H // n
    resid_levels, recs = [], []
    plane = np.full((H, W), 0, dtype=np.int64)
    mode = 2
    res, coefs = [], []
    for by in range(bh):
        for bx in range(bw):
            x, y = bx * n, by * n
            left_ok, top_ok, tr_ok = bx > 0, by > 0, by > 0 and bx + 1 < bw
            al = 0b0011 if left_ok else 0           # 2n / 4 = 4 units: the two beside the block exist, the two below do not yet
            at = (0b0011 if top_ok else 0) | (0b1100 if tr_ok else 0)
            corner = int(left_ok and top_ok)
            recs.append((x, y, 3, mode, corner | (4 if corner else 0), 0, al, at, al, at))
            # a DC level per block so that the samples differ from block to block
            dc = int(rnd.integers(-200, 200))
            res.append((x, y, 3, 0, 30, 0, 0xff, [0, 0, 0], len(coefs), 1))
            coefs.append((0 << 16) | (dc & 0xffff))
            r = hn.residual(np.array([[dc] + [0] * 7] + [[0] * 8] * 7), 30, 1, 0)
            line_v = np.zeros(4 * n + 1, dtype=np.int64)
            line_a = np.zeros(4 * n + 1, dtype=bool)
            for k in range(2 * n):
                if (al >> (k // 4)) & 1:
                    line_v[2 * n - 1 - k], line_a[2 * n - 1 - k] = plane[y + k, x - 1], True
                if (at >> (k // 4)) & 1:
                    line_v[2 * n + 1 + k], line_a[2 * n + 1 + k] = plane[y - 1, x + k], True
            if corner:
                line_v[2 * n], line_a[2 * n] = plane[y - 1, x - 1], True
            line = substitute(line_a, line_v, n)
            pred = hn.intra_predict(np.concatenate([[line[2 * n]], line[:2 * n][::-1]]), line[2 * n:], n, 0, mode, True)
            plane[y:y + n, x:x + n] = np.clip(pred + r, 0, 255)
            mode = 2 + (mode - 1) % 33 if (bx + by) % 3 else (bx % 2)     # angular modes mostly, planar / DC now and then
    intra = (np.array(recs, dtype=ITU), None, None)
    for rows_kernel in (False, True):   # (for the row kernel every 8x8 block is a "coding tree block" of its own: raster order is its order)
        got, _ = run_picture(hw, W, H, log2_ctb=3 if rows_kernel else 6, intra=intra, res=np.array(res, dtype=RES), coefs=np.array(coefs, dtype=np.uint32),
                             intra_rows=rows_kernel)
        assert np.array_equal(got[0].astype(np.int64), plane), rows_kernel

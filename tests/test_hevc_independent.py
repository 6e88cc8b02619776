"""The HEVC oracle's reconstruction arithmetic (oracle/hevc_recon.c, through its stand-alone entry points) against the
numpy statements of tests/hevc_numpy.py, on random and on adversarial inputs.  CPU only; tests/test_gpu_hevc.py runs
the same statements against the HIP kernels."""
import ctypes as C

import numpy as np
import pytest

from tests import hevc_numpy as hn


@pytest.fixture(scope="module")
def lib(built):
    from oracle import oracle_py
    return oracle_py.load_hevc()


@pytest.mark.parametrize("log2", [2, 3, 4, 5])
@pytest.mark.parametrize("intra", [0, 1])
def test_residual_blocks(lib, log2, intra):
    rnd = np.random.default_rng(100 + log2 * 2 + intra)
    n = 1 << log2
    for case in range(60):
        qp = int(rnd.integers(0, 52))
        c_idx = int(rnd.integers(0, 3))
        lev = np.zeros((n, n), dtype=np.int16)
        k = int(rnd.integers(1, 1 + min(n * n, 24)))
        ys, xs = rnd.integers(0, n, k), rnd.integers(0, n, k)
        big = case % 7 == 0
        lev[ys, xs] = rnd.integers(-32768 if big else -40, 32768 if big else 41, k)
        if case % 5 == 0:
            lev[:] = 0
            lev[0, 0] = rnd.integers(-600, 600)        # DC only
        tskip = int(log2 == 2 and case % 3 == 0)
        res = np.zeros((n, n), dtype=np.int16)
        lib.hevco_residual_block(lev.ctypes.data, log2, c_idx, qp, intra, tskip, res.ctypes.data)
        want = hn.residual(lev, qp, intra, c_idx, transform_skip=bool(tskip))
        assert np.array_equal(res.astype(np.int64), want.astype(np.int16).astype(np.int64)), (case, qp, c_idx, tskip)


@pytest.mark.parametrize("log2", [2, 3, 4, 5])
def test_intra_modes(lib, log2):
    rnd = np.random.default_rng(7 + log2)
    n = 1 << log2
    for case in range(12):
        kind = case % 4
        if kind == 0:
            line = rnd.integers(0, 256, 4 * n + 1)
        elif kind == 1:        # smooth: the strong filter's condition holds for 32x32
            line = np.clip(100 + np.arange(4 * n + 1) // 8 + rnd.integers(0, 2, 4 * n + 1), 0, 255)
        elif kind == 2:
            line = np.full(4 * n + 1, int(rnd.integers(0, 256)))
        else:
            line = np.clip(np.cumsum(rnd.integers(-9, 10, 4 * n + 1)) + 128, 0, 255)
        left = np.concatenate([[line[2 * n]], line[:2 * n][::-1]]).astype(np.uint8)    # corner, then downwards
        top = line[2 * n + 1:].astype(np.uint8)
        for c_idx in (0, 1):
            for mode in range(35):
                for strong in (0, 1):
                    pred = np.zeros((n, n), dtype=np.uint8)
                    lib.hevco_intra_block(left.ctypes.data, top.ctypes.data, log2, c_idx, mode, strong, pred.ctypes.data)
                    want = hn.intra_predict(left.astype(np.int64), np.concatenate([[left[0]], top]).astype(np.int64), n, c_idx, mode, bool(strong))
                    assert np.array_equal(pred.astype(np.int64), want), (case, c_idx, mode, strong)


def test_interpolation_all_positions(lib):
    rnd = np.random.default_rng(5)
    rw, rh = 48, 40
    ref = rnd.integers(0, 256, (rh, rw)).astype(np.uint8)
    ref[:, :6] = 255
    ref[:5, :] = 0          # hard edges where the padding is
    for c_idx, fr in ((0, 4), (1, 8)):
        for yf in range(fr):
            for xf in range(fr):
                for (x, y, w, h, ix, iy) in ((8, 8, 16, 8, 1, -2), (0, 0, 8, 8, -3, -3), (40, 32, 8, 8, 5, 4), (4, 36, 4, 4, -20, 30)):
                    mvx, mvy = ix * fr + xf, iy * fr + yf
                    pred = np.zeros((h, w), dtype=np.uint8)
                    lib.hevco_inter_block(ref.ctypes.data, rw, rh, c_idx, x, y, w, h, mvx, mvy, pred.ctypes.data)
                    want = hn.inter_default(ref, x, y, w, h, mvx, mvy, chroma=bool(c_idx))
                    assert np.array_equal(pred.astype(np.int64), want), (c_idx, xf, yf, x, y)


def test_deblocking_segments(lib):
    rnd = np.random.default_rng(11)
    seen = set()
    for case in range(4000):
        base = int(rnd.integers(20, 236))
        step = int(rnd.integers(0, 24))
        noise = int(rnd.integers(0, 6))
        blk = np.zeros((4, 8), dtype=np.int64)
        blk[:, :4] = base
        blk[:, 4:] = base + step * (1 if case & 1 else -1)
        blk += rnd.integers(-noise, noise + 1, (4, 8))
        if case % 9 == 0:
            blk += np.arange(8)[None, :] * int(rnd.integers(-3, 4))     # a ramp: the second-derivative tests stay small
        blk = np.clip(blk, 0, 255)
        bs = 1 + (case % 2)
        qp_p, qp_q = int(rnd.integers(0, 52)), int(rnd.integers(0, 52))
        bo, to = int(rnd.integers(-6, 7)), int(rnd.integers(-6, 7))
        no_p, no_q = case % 13 == 0, case % 17 == 0
        got = blk.astype(np.uint8).copy()
        lib.hevco_deblock_luma(got.ctypes.data, bs, qp_p, qp_q, bo, to, int(no_p), int(no_q))
        want = hn.deblock_luma(blk, bs, qp_p, qp_q, bo, to, no_p, no_q)
        assert np.array_equal(got.astype(np.int64), want), case
        changed = (want != blk).any(axis=0)
        seen.add((bool(changed[1] or changed[6]), bool(changed[2] or changed[5]), bool(changed[3] or changed[4])))
    assert (True, True, True) in seen and (False, True, True) in seen and (False, False, True) in seen and (False, False, False) in seen
    for case in range(600):
        row = rnd.integers(0, 256, 4)
        if case % 2:
            row = np.clip(int(rnd.integers(10, 240)) + rnd.integers(-8, 9, 4), 0, 255)
        qp_p, qp_q, off, to = int(rnd.integers(0, 52)), int(rnd.integers(0, 52)), int(rnd.integers(-12, 13)), int(rnd.integers(-6, 7))
        got = row.astype(np.uint8).copy()
        lib.hevco_deblock_chroma(got.ctypes.data, qp_p, qp_q, off, to)
        assert list(got) == hn.deblock_chroma(row, qp_p, qp_q, off, to), case


def test_sao_blocks(lib):
    rnd = np.random.default_rng(13)
    for case in range(200):
        w, h = int(rnd.integers(3, 20)), int(rnd.integers(3, 20))
        a = rnd.integers(0, 256, (h, w)).astype(np.uint8)
        if case % 3 == 0:
            a = np.clip(128 + rnd.integers(-3, 4, (h, w)), 0, 255).astype(np.uint8)    # many ties: the "none" category
        typ = 1 + case % 2
        band, cls = int(rnd.integers(0, 32)), int(rnd.integers(0, 4))
        off = rnd.integers(-7, 8, 4).astype(np.int8)
        out = np.zeros_like(a)
        lib.hevco_sao_block.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.hevco_sao_block(a.ctypes.data, w, h, typ, band, cls, off.ctypes.data, out.ctypes.data)
        assert np.array_equal(out.astype(np.int64), hn.sao(a, typ, band, cls, off)), (case, typ, band, cls)

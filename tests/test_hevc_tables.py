"""Structural checks of the HEVC constant tables (oracle/hevc_tables.h and, byte for byte, the product's copy): nothing
here needs a stream or a decoder.  Each check states a property the standard's table has by construction, or compares
the table with a second, differently organised recollection of it (tests/hevc_numpy.py)."""
import ctypes as C

import numpy as np
import pytest

from tests import hevc_numpy as hn


@pytest.fixture(scope="module")
def tab(built):
    from oracle import oracle_py
    lib = oracle_py.load_hevc()
    lib.hevco_table.restype = C.c_void_p
    lib.hevco_table.argtypes = [C.c_char_p, C.POINTER(C.c_int)]

    def get(name, dtype, shape=None):
        n = C.c_int()
        p = lib.hevco_table(name.encode(), C.byref(n))
        assert p and n.value, name
        a = np.frombuffer((C.c_uint8 * n.value).from_address(p), dtype=dtype).copy()
        return a.reshape(shape) if shape else a
    return get


def test_transform_matrix_is_the_cosine_construction(tab):
    m = tab("trans_matrix", np.int8, (32, 32)).astype(np.int64)
    assert np.array_equal(m, hn.trans_matrix(32))
    for n in (4, 8, 16, 32):
        t = hn.trans_matrix(n)
        g = t @ t.T
        diag = np.diag(g)
        assert np.all(np.abs(diag - 4096 * n) <= 0.012 * 4096 * n), (n, diag)      # rows of (nearly) equal norm 64 sqrt(n)
        off = g - np.diag(diag)
        assert np.abs(off).max() <= 0.011 * 4096 * n, (n, np.abs(off).max())       # ... and nearly orthogonal
        assert np.array_equal(t[::2, :], np.concatenate([hn.trans_matrix(n // 2), hn.trans_matrix(n // 2)[:, ::-1]], axis=1)) or n == 4
        # even rows are symmetric, odd rows antisymmetric
        assert np.array_equal(t[::2], t[::2, ::-1]) and np.array_equal(t[1::2], -t[1::2, ::-1])
    assert [int(v) for v in hn.trans_matrix(4).ravel()] == [64, 64, 64, 64, 83, 36, -36, -83, 64, -64, -64, 64, 36, -83, 83, -36]


def test_dst_matrix(tab):
    d = tab("dst4", np.int8, (4, 4)).astype(np.int64)
    assert np.array_equal(d, hn.DST4)
    g = d @ d.T
    assert np.all(np.abs(np.diag(g) - 16384) <= 60) and np.abs(g - np.diag(np.diag(g))).max() <= 60
    # basis function k at sample i is sin((2 k + 1)(i + 1) pi / 9), scaled by 128 * 2 / 3
    i, k = np.meshgrid(np.arange(4), np.arange(4))
    assert np.array_equal(d, np.round(128 * 2 / 3 * np.sin((2 * k + 1) * (i + 1) * np.pi / 9)).astype(np.int64))


def test_interpolation_filters(tab):
    lf = tab("luma_filter", np.int8, (4, 8)).astype(int)
    cf = tab("chroma_filter", np.int8, (8, 4)).astype(int)
    assert np.all(lf.sum(axis=1) == 64) and np.all(cf.sum(axis=1) == 64)
    assert np.array_equal(lf[3], lf[1][::-1]) and np.array_equal(lf[2], lf[2][::-1])
    for k in range(1, 8):
        assert np.array_equal(cf[k], cf[8 - k][::-1])
    assert np.array_equal(lf, np.array([hn.LUMA_TAPS[k] for k in range(4)]))
    assert np.array_equal(cf, np.array([hn.CHROMA_TAPS[k] for k in range(8)]))


def test_intra_angles(tab):
    ang = tab("intra_angle", np.int8).astype(int)
    inv = tab("inv_angle", np.int16).astype(int)
    assert len(ang) == 35 and len(inv) == 15
    for m in range(2, 35):
        assert ang[m] == hn.ANGLES[m] == ang[36 - m]
    for m in range(11, 26):
        assert inv[m - 11] == hn.inv_angle(ang[m])


def test_deblocking_tables(tab):
    beta, tc, qpc = tab("beta", np.uint8).astype(int), tab("tc", np.uint8).astype(int), tab("qpc", np.uint8).astype(int)
    assert len(beta) == 52 and len(tc) == 54 and len(qpc) == 13
    assert np.all(np.diff(beta) >= 0) and np.all(np.diff(tc) >= 0) and np.all(np.diff(qpc) >= 0)
    assert [hn.beta_prime(q) for q in range(52)] == list(beta)
    assert [hn.tc_prime(q) for q in range(54)] == list(tc)
    assert [hn.qpc_of(q) for q in range(30, 43)] == list(qpc) and hn.qpc_of(29) == 29 and hn.qpc_of(43) == 37   # continuous at both ends


def test_cabac_tables(tab):
    init = tab("cabac_init", np.uint8, (3, 154)).astype(int)
    assert np.all(init > 0)
    # contexts of inter syntax are "not used" (154) in intra slices and only there
    inter_only = list(range(6, 10)) + [11, 12, 13] + list(range(16, 27)) + [36, 37]
    assert all(init[0][i] == 154 for i in inter_only)
    rng = tab("range_lps", np.uint8, (64, 4)).astype(int)
    assert np.all(np.diff(rng[:63], axis=0) <= 0) and np.all(np.diff(rng, axis=1)[:63] >= 0) and np.all(rng[63] == 2)
    # the LPS range is round(qRange * 0.5 * alpha^state) within a unit, alpha = (0.01875 / 0.5)^(1/63), never below 6
    alpha = (0.01875 / 0.5) ** (1.0 / 63)
    for q in range(4):
        mid = 256 + 64 * q + 32
        model = np.maximum(6, np.floor(mid * 0.5 * alpha ** np.arange(63)))
        if q == 0:
            model = np.minimum(model, 128)     # (never more than half of the smallest range)
        assert np.abs(rng[:63, q] - model).max() <= 2, q
    lps = tab("trans_lps", np.uint8).astype(int)
    assert lps[0] == 0 and lps[63] == 63 and np.all(lps[1:63] < np.arange(1, 63)) and np.all(np.diff(lps[:63]) >= 0)


def test_scans(tab):
    for l in range(4):
        for s in range(3):
            sc = tab("scan%d%d" % (l, s), np.uint8).astype(int)
            n = 1 << l
            assert sorted((v >> 4) * n + (v & 15) for v in sc) == list(range(n * n))
    d4 = [(v & 15, v >> 4) for v in tab("scan20", np.uint8)]
    assert d4[:6] == [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)] and d4[-1] == (3, 3)


def test_default_scaling_lists(tab):
    # Table 7-6 as 8x8 matrices (a second recollection, in raster form) against the oracle's lists in diagonal order
    intra = [16, 16, 16, 16, 17, 18, 21, 24, 16, 16, 16, 16, 17, 19, 22, 25, 16, 16, 17, 18, 20, 22, 25, 29, 16, 16, 18, 21, 24, 27, 31, 36,
             17, 17, 20, 24, 30, 35, 41, 47, 18, 19, 22, 27, 35, 44, 54, 65, 21, 22, 25, 31, 41, 54, 70, 88, 24, 25, 29, 36, 47, 65, 88, 115]
    inter = [16, 16, 16, 16, 17, 18, 20, 24, 16, 16, 16, 17, 18, 20, 24, 25, 16, 16, 17, 18, 20, 24, 25, 28, 16, 17, 18, 20, 24, 25, 28, 33,
             17, 18, 20, 24, 25, 28, 33, 41, 18, 20, 24, 25, 28, 33, 41, 54, 20, 24, 25, 28, 33, 41, 54, 71, 24, 25, 28, 33, 41, 54, 71, 91]
    sc = tab("scan30", np.uint8).astype(int)
    for name, want in (("scaling_intra8", intra), ("scaling_inter8", inter)):
        lst = tab(name, np.uint8).astype(int)
        m = np.zeros((8, 8), dtype=int)
        for i, v in enumerate(sc):
            m[v >> 4, v & 15] = lst[i]
        assert np.array_equal(m, np.array(want).reshape(8, 8)) and np.array_equal(m, m.T)

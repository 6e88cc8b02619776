"""Which cells of the oracle's coverage histogram (oracle/h264_oracle.h, H264O_COV_*) a conforming progressive
4:2:0 8-bit stream can reach at all.  Written from the standard's structure, independently of the oracle's code:
tests/test_corpus_roundtrip.py asserts that the corpus reaches every cell listed here and nothing else."""
import itertools

from oracle.oracle_py import COV_SECTIONS

# zig-zag index of the 4x4 block at (bx, by) inside a macroblock (6.4.3)
_ZZ = [[0, 1, 4, 5], [2, 3, 6, 7], [8, 9, 12, 13], [10, 11, 14, 15]]


def _intra_patterns(n4):
    """Feasible (left, top, topleft, topright) availability patterns of an Intra4x4 (n4 = 1) or Intra8x8 (n4 = 2)
    block.  Neighbouring macroblocks A (left), B (top), C (top right), D (top left) may each be usable or not in any
    combination (slice edges, picture edges, constrained_intra_pred with inter neighbours); blocks inside the
    macroblock are usable when they precede the current one in decoding order (6.4.11.4)."""
    pats = set()
    for A, B, C, D in itertools.product((0, 1), repeat=4):
        for by in range(0, 4, n4):
            for bx in range(0, 4, n4):
                zz = _ZZ[by][bx]

                def avail(x, y):
                    if x < 0 and y < 0:
                        return D
                    if x < 0:
                        return A
                    if y < 0:
                        return C if x > 3 else B
                    if x > 3:
                        return 0
                    return 1 if _ZZ[y][x] < zz else 0
                pats.add(avail(bx - 1, by) | avail(bx, by - 1) << 1 | avail(bx - 1, by - 1) << 2 |
                         avail(bx + n4, by - 1) << 3)
    return pats


def _intra_nxn(n4):
    need = {0: 2, 1: 1, 2: 0, 3: 2, 4: 7, 5: 7, 6: 7, 7: 2, 8: 1}   # bits of the pattern a mode requires (8.3.1.2 / 8.3.2.2)
    return {mode * 16 + pat for pat in _intra_patterns(n4) for mode in range(9) if (pat & need[mode]) == need[mode]}


def _ctx_cells():
    """(table, ctxIdx) pairs a frame-coded 4:2:0 slice can touch (Table 9-34 ranges); both bin values of each."""
    residual = (list(range(60, 70)) + list(range(73, 105)) + list(range(105, 166)) + list(range(166, 227)) +
                list(range(227, 276)) + list(range(399, 436)))
    by_type = {
        "I": list(range(3, 11)) + residual,
        "P": list(range(11, 24)) + list(range(40, 60)) + residual,          # skip, mb_type (+ intra prefix), sub, mvd, ref_idx
        "B": list(range(24, 40)) + list(range(40, 60)) + residual,
    }
    cells = set()
    for table in range(7):
        t = "I" if table == 0 else ("P" if table < 4 else "B")
        for ctx in by_type[t]:
            cells.add((table * 460 + ctx) * 2)
            cells.add((table * 460 + ctx) * 2 + 1)
    return cells


def reachable():
    """section name -> set of reachable cell indices (section-relative)."""
    r = {}
    r["i4"] = _intra_nxn(1)
    r["i8"] = _intra_nxn(2)
    # Intra16x16: V needs top, H left, DC nothing, plane both (+ top left, implied); chroma: DC, H, V, plane
    r["i16"] = {m * 4 + p for m in range(4) for p in range(4) if (p & {0: 2, 1: 1, 2: 0, 3: 3}[m]) == {0: 2, 1: 1, 2: 0, 3: 3}[m]}
    r["ic"] = {m * 4 + p for m in range(4) for p in range(4) if (p & {0: 0, 1: 1, 2: 2, 3: 3}[m]) == {0: 0, 1: 1, 2: 2, 3: 3}[m]}
    r["luma_mc"] = set(range(48))
    r["chroma_mc"] = set(range(64))
    r["wp"] = set(range(9))
    # bS 4 only on macroblock edges, bS 3 only inside (frame macroblocks, 8.7.2.1)
    r["bs"] = {((d * 2 + e) * 3 + pl) * 5 + bs for d in range(2) for e in range(2) for pl in range(3) for bs in range(5)
               if not (e == 0 and bs == 4) and not (e == 1 and bs == 3)}
    r["filt"] = set(range(12))
    r["mbtype_i"] = set(range(26))
    r["mbtype_p"] = set(range(32))
    r["mbtype_b"] = set(range(50))
    r["sub_p"] = set(range(4))
    r["sub_b"] = set(range(13))
    tok = set()
    for table in range(5):
        for total in range(0, 5 if table == 4 else 17):
            for t1 in range(0, min(3, total) + 1):
                tok.add(table * 68 + total * 4 + t1)
    r["coeff_token"] = tok
    r["total_zeros"] = {(t - 1) * 16 + z for t in range(1, 16) for z in range(0, 17 - t)}
    r["total_zeros_dc"] = {(t - 1) * 4 + z for t in range(1, 4) for z in range(0, 5 - t)}
    r["run_before"] = {(c - 1) * 15 + run for c in range(1, 8) for run in range(0, (c if c < 7 else 14) + 1)}
    r["escape"] = set(range(5))
    r["dequant"] = set(range(7))
    r["direct"] = set(range(8))
    r["ctx"] = _ctx_cells()
    assert set(r) == set(COV_SECTIONS)
    return r

"""Structural self-checks of the oracle's entropy-coding tables: every VLC
table must be a complete prefix code (Kraft sum exactly 1, no codeword a
prefix of another).  This catches mistyped lengths or code values in tables
that were restated from memory (see oracle/h264_tables.h)."""
import os
import re
from fractions import Fraction

from tests.util import ROOT


def parse_tables():
    text = open(os.path.join(ROOT, "oracle", "h264_tables.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"static const uint8_t (\w+)((?:\[[^\]]*\])+)\s*=\s*(\{.*?\});", text, flags=re.S):
        name, body = m.group(1), m.group(3)
        body = body.replace("{", "[").replace("}", "]")
        out[name] = eval(body)
    return out


def check_prefix_code(lens, codes, what, allow_zero_hole=True):
    """Complete prefix code; with allow_zero_hole the only unused part of the
    code space may be one all-zeros word (H.264 leaves the all-zeros prefix of
    the long coeff_token tables unassigned)."""
    words = []
    for ln, c in zip(lens, codes):
        if ln == 0:
            continue
        assert c < (1 << ln), (what, ln, c)
        words.append(format(c, "0%db" % ln))
    assert len(set(words)) == len(words), what + ": duplicate codeword"
    kraft = sum(Fraction(1, 2 ** len(w)) for w in words)
    if kraft != 1:
        hole = 1 - kraft
        assert allow_zero_hole and hole.numerator == 1, what + ": Kraft sum %s" % kraft
        zeros = "0" * (hole.denominator.bit_length() - 1)
        assert not any(w.startswith(zeros) or zeros.startswith(w) for w in words), \
            what + ": code space hole is not the all-zeros word"
    ws = sorted(words)
    for a, b in zip(ws, ws[1:]):
        assert not b.startswith(a), what + ": %s is a prefix of %s" % (a, b)
    return len(words)


def test_coeff_token_tables():
    t = parse_tables()
    for k in range(3):
        # tables 0-2 are complete codes over the 62 legal (total, t1) pairs
        n = check_prefix_code(t["coeff_token_len"][k], t["coeff_token_bits"][k], "coeff_token[%d]" % k,
                              allow_zero_hole=True)
        assert n == 62
    # table 3 is a 6-bit FLC over 62 of 64 words
    lens, codes = t["coeff_token_len"][3], t["coeff_token_bits"][3]
    words = {c for ln, c in zip(lens, codes) if ln}
    assert all(ln in (0, 6) for ln in lens) and len(words) == 62
    assert check_prefix_code(t["chroma_dc_coeff_token_len"], t["chroma_dc_coeff_token_bits"],
                             "chroma_dc_coeff_token") == 14


def test_legal_token_positions():
    t = parse_tables()
    for k in range(4):
        for tc in range(17):
            for t1 in range(4):
                legal = t1 <= min(tc, 3)
                assert (t["coeff_token_len"][k][4 * tc + t1] != 0) == legal


def test_total_zeros_and_run_before():
    t = parse_tables()
    for i in range(15):
        n = t["total_zeros_count"][i]
        assert n == 16 - i
        check_prefix_code(t["total_zeros_len"][i][:n], t["total_zeros_bits"][i][:n], "total_zeros[%d]" % i)
    for i, n in enumerate((4, 3, 2)):
        check_prefix_code(t["chroma_dc_total_zeros_len"][i][:n], t["chroma_dc_total_zeros_bits"][i][:n],
                          "chroma_dc_total_zeros[%d]" % i)
    for i in range(7):
        n = t["run_before_count"][i]
        if i < 6:
            check_prefix_code(t["run_before_len"][i][:n], t["run_before_bits"][i][:n], "run_before[%d]" % i)
        else:
            # zerosLeft > 6: 000..1 unary tail is open-ended (runs up to 14)
            words = [format(c, "0%db" % l) for l, c in zip(t["run_before_len"][i][:n], t["run_before_bits"][i][:n])]
            assert len(set(words)) == n
            ws = sorted(words)
            assert all(not b.startswith(a) for a, b in zip(ws, ws[1:]))


def test_scans_are_permutations():
    t = parse_tables()
    assert sorted(t["zigzag4x4"]) == list(range(16))
    assert sorted(t["zigzag8x8"]) == list(range(64))
    assert sorted(t["cbp_intra_from_code"]) == list(range(48))
    assert sorted(t["cbp_inter_from_code"]) == list(range(48))
    assert len(t["cabac_range_lps"]) == 64 and len(t["cabac_trans_lps"]) == 64
    # rangeTabLPS decreases with the state index, grows with the range quantile
    r = t["cabac_range_lps"]
    for s in range(62):
        assert all(r[s][q] >= r[s + 1][q] for q in range(4))
        assert r[s][0] <= r[s][1] <= r[s][2] <= r[s][3]

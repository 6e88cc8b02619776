"""The product carries its own copies of the H.264 constant tables (it must
not link or include anything under oracle/).  This test keeps the copies
identical to the oracle's, which tests/test_oracle_tables.py checks against
structural properties (Kraft sums, monotonicity, known entries)."""
import os
import re

from tests.util import ROOT

PRODUCT = os.path.join(ROOT, "hwang_amd", "csrc", "h264", "h264_tables.h")
ORACLE = os.path.join(ROOT, "oracle", "h264_tables.h")


def tables(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    out = {}
    for m in re.finditer(r"const\s+u?int8_t\s+(\w+)\s*((?:\[[^\]]*\])+)\s*=\s*\{(.*?)\};", text, flags=re.S):
        nums = [int(x, 0) for x in re.findall(r"-?(?:0x[0-9a-fA-F]+|\d+)", m.group(3))]
        out[m.group(1)] = nums
    return out


def test_constant_tables_identical():
    p, o = tables(open(PRODUCT).read()), tables(open(ORACLE).read())
    common = sorted(set(p) & set(o))
    assert len(common) >= 30, common
    assert set(o) <= set(p) | {"cabac_init_mn"}, sorted(set(o) - set(p))
    for name in common:
        assert p[name] == o[name], name


def test_cabac_init_tables_identical():
    prod = tables(open(os.path.join(ROOT, "hwang_amd", "csrc", "h264", "h264_cabac_init.cpp")).read())
    assert len(prod["cabac_init_mn"]) == 4 * 460 * 2
    # the oracle generates its table at build time (oracle/gen_cabac_init.py)
    import subprocess
    import sys
    text = subprocess.check_output([sys.executable, os.path.join(ROOT, "oracle", "gen_cabac_init.py"), "cabac_init_mn"],
                                   text=True)
    ora = tables(text)["cabac_init_mn"]
    assert prod["cabac_init_mn"] == ora


# ---------------------------------------------------------------------------------------------------------- HEVC
def tables_any(text):
    """Every constant integer array of a C / HIP source: name -> flat list (HC_CNU stands for 154)."""
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = text.replace("HC_CNU", "154")
    out = {}
    for m in re.finditer(r"(?:const|__constant__)\s+u?int(?:8|16|32)?_?t?\s+(\w+)\s*((?:\[[^\]]*\])+)\s*=\s*\{(.*?)\};", text, flags=re.S):
        out[m.group(1)] = [int(x, 0) for x in re.findall(r"-?(?:0x[0-9a-fA-F]+|\d+)", m.group(3))]
    return out


def test_hevc_tables_identical_in_oracle_product_and_kernels():
    """hwang_amd/csrc/hevc/hevc_tables.h (host parser), oracle/hevc_tables.h and the __constant__ copies inside
    k_hevc.hip are the same constants; tests/test_hevc_tables.py checks their structure on the oracle's."""
    o = tables_any(open(os.path.join(ROOT, "oracle", "hevc_tables.h")).read())
    p = tables_any(open(os.path.join(ROOT, "hwang_amd", "csrc", "hevc", "hevc_tables.h")).read())
    assert len(o) >= 15 and sorted(o) == sorted(p)
    for name in o:
        assert o[name] == p[name] and len(o[name]) > 0, name
    assert len(o["hevc_cabac_init"]) == 3 * 154 and len(o["hevc_trans_matrix"]) == 1024
    k = tables_any(open(os.path.join(ROOT, "hwang_amd", "csrc", "amd", "k_hevc.hip")).read())
    pairs = {"kTrans": "hevc_trans_matrix", "kDst": "hevc_dst4", "kLumaTaps": "hevc_luma_filter", "kChromaTaps": "hevc_chroma_filter",
             "kAngle": "hevc_intra_angle", "kInvAngle": "hevc_inv_angle", "kBeta": "hevc_beta_table", "kTc": "hevc_tc_table", "kQpc": "hevc_qpc_table"}
    for kn, on in pairs.items():
        assert k[kn] == o[on], kn
    assert k["kLevelScale"] == [40, 45, 51, 57, 64, 72]

"""HEVC: the corpus encoder's streams decode, through the oracle, to the encoder's own reconstruction -
every slice ending on its stop bit, every picture's hash SEI matching - and to the digests frozen in
tests/golden/hevc_corpus_md5.json.  That pins the ORACLE against later edits, not against the standard: HEVC parity is
unpinned (oracle/hevc_oracle.h)."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests.corpus_util import HEVC_MATRIX, make_hevc_clip
from tests.util import ROOT

GOLDEN = os.path.join(ROOT, "tests", "golden", "hevc_corpus_md5.json")


def digest(frame):
    return hashlib.md5(np.ascontiguousarray(frame).tobytes()).hexdigest()


@pytest.mark.parametrize("name,kw", HEVC_MATRIX, ids=[m[0] for m in HEVC_MATRIX])
def test_oracle_decodes_the_encoders_reconstruction(hw, oracle, tmp_path, name, kw):
    kw = dict(kw)
    w, h = kw.pop("width", 128), kw.pop("height", 96)
    path = str(tmp_path / (name + ".mp4"))
    gold = make_hevc_clip(path, width=w, height=h, **kw)
    idx = hw.index_video(path)
    assert idx.format() in ("hev1", "hvc1") and idx.frames() == len(gold)
    frames, ctr = oracle.decode_file_hevc(path, idx)
    assert len(frames) == len(gold)
    for i, f in enumerate(frames):
        assert (f[1], f[2]) == (w, h)
        assert np.array_equal(f[0], gold[i]), "%s: frame %d" % (name, i)
    assert ctr["slices_clean_end"] == ctr["slices"] > 0
    assert ctr["md5_checked"] == ctr["pictures"] == len(gold) and ctr["md5_mismatch"] == 0
    if os.environ.get("HWANG_WRITE_GOLDEN"):
        all_ = json.load(open(GOLDEN)) if os.path.exists(GOLDEN) else {}
        all_[name] = [digest(f[0]) for f in frames]
        json.dump(all_, open(GOLDEN, "w"), indent=0, sort_keys=True)
    want = json.load(open(GOLDEN))[name]
    assert [digest(f[0]) for f in frames] == want


def test_tools_are_exercised(hw, oracle, tmp_path):
    """The counters say the clips are not trivially intra / skip."""
    path = str(tmp_path / "c.mp4")
    make_hevc_clip(path, frames=9, gop=9, bframes=2, refs=2)
    idx = hw.index_video(path)
    _, ctr = oracle.decode_file_hevc(path, idx)
    assert ctr["p_slices"] and ctr["b_slices"] and ctr["skip_cus"] and ctr["intra_cus"]
    assert ctr["coding_units"] > ctr["skip_cus"] + ctr["intra_cus"]          # coded inter units too
    assert ctr["transform_units"] > ctr["coding_units"] // 2


def test_a_damaged_stream_is_an_error_not_a_crash(hw, oracle, tmp_path):
    path = str(tmp_path / "c.mp4")
    make_hevc_clip(path, frames=3, gop=3)
    idx = hw.index_video(path)
    offs, sizes = idx.sample_offsets(), idx.sample_sizes()
    blob = bytearray(open(path, "rb").read())
    rnd = np.random.default_rng(3)
    for trial in range(30):
        b = bytearray(blob)
        for _ in range(4):
            s = int(rnd.integers(0, idx.frames()))
            b[offs[s] + int(rnd.integers(6, sizes[s]))] ^= 1 << int(rnd.integers(0, 8))
        dec = oracle.HevcOracleDecoder(idx.metadata_bytes())
        try:
            for s in range(idx.frames()):
                dec.decode(bytes(b[offs[s]:offs[s] + sizes[s]]))
            dec.flush()
        except RuntimeError:
            pass

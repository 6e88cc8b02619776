"""Corpus generator <-> oracle decoder: the stream the generator writes must
parse without desync and decode to exactly the generator's reconstruction
(necessary, not sufficient: both share the oracle's arithmetic; the real
x264 stream in test_oracle_realstream.py is the independent anchor)."""
import numpy as np
import pytest

from tests.corpus_util import MATRIX, make_clip


@pytest.mark.parametrize("name,kw", MATRIX, ids=[m[0] for m in MATRIX])
def test_roundtrip(hw, oracle, tmp_path, name, kw):
    kw = dict(kw)
    w, h = kw.pop("width", 322 if name == "qp40" else 320), kw.pop("height", 242 if name == "qp40" else 240)
    path = str(tmp_path / (name + ".mp4"))
    gold = make_clip(path, width=w, height=h, **kw)
    idx = hw.index_video(path)
    assert (idx.frame_width(), idx.frame_height()) == (w, h)
    assert idx.frames() == kw["frames"]
    assert idx.keyframe_indices() == list(range(0, kw["frames"], kw["gop"]))
    frames, c = oracle.decode_file(path, idx)
    assert c["slices_clean_end"] == c["slices"] > 0
    assert len(frames) == len(gold)
    for i, f in enumerate(frames):
        assert np.array_equal(f[0], gold[i]), "frame %d differs" % i
    if kw.get("bframes"):
        assert c["b_slices"] > 0
    if name == "pcm":
        assert c["intra_mbs"] > 0


def test_sparse_equals_sequential_on_oracle(hw, oracle, tmp_path):
    """The reference's only results test (decoder_automata_test.cpp:262-342)
    restated for the oracle: decoding only the GOP that holds a frame gives
    the same frame as decoding the whole file."""
    path = str(tmp_path / "s.mp4")
    gold = make_clip(path, frames=24, gop=8, bframes=2)
    idx = hw.index_video(path)
    frames, _ = oracle.decode_file(path, idx, first=16, count=8)
    assert len(frames) == 8
    for i in range(8):
        assert np.array_equal(frames[i][0], gold[16 + i])

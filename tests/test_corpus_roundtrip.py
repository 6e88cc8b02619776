"""Corpus generator <-> oracle decoder: the stream the generator writes must
parse without desync and decode to exactly the generator's reconstruction
(necessary, not sufficient: both share the oracle's arithmetic; the real
x264 stream in test_oracle_realstream.py is the independent anchor)."""
import hashlib
import json
import os

import numpy as np
import pytest

from tests.corpus_util import BAND_SIZES, MATRIX, golden_md5, make_clip, plane_md5s
from tests.util import ROOT

# coverage histogram of the oracle (oracle/h264_oracle.h) summed over the MATRIX cases decoded in this process
_COVERAGE = {}


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
    cov = np.zeros(oracle.COV_CELLS, dtype=np.int64)
    frames, c = oracle.decode_file(path, idx, coverage=cov)
    _COVERAGE[name] = cov
    assert c["slices_clean_end"] == c["slices"] > 0
    assert len(frames) == len(gold)
    for i, f in enumerate(frames):
        assert np.array_equal(f[0], gold[i]), "frame %d differs" % i
    # the committed digests (tests/golden/corpus_md5.json): the encoder still writes the same bytes and the
    # oracle still decodes them to the same samples as when the fixture was made
    g = golden_md5(name)
    assert hashlib.md5(open(path, "rb").read()).hexdigest() == g["mp4_md5"], "the corpus encoder's output changed"
    assert [plane_md5s(f[0], w, h) for f in frames] == g["frames"], "the oracle's output changed"
    if kw.get("bframes"):
        assert c["b_slices"] > 0
    if name == "pcm":
        assert c["intra_mbs"] > 0


@pytest.mark.parametrize("name,w,h,kw", BAND_SIZES, ids=[b[0] for b in BAND_SIZES])
def test_band_cases_match_committed_digests(hw, oracle, tmp_path, name, w, h, kw):
    """The picture shapes the GPU row kernels are tested on (tests/test_gpu_decode.py): the oracle's output equals the
    encoder's reconstruction and the committed digests the GPU test compares the HIP path with."""
    path = str(tmp_path / (name + ".mp4"))
    gold = make_clip(path, width=w, height=h, **kw)
    frames, c = oracle.decode_file(path, hw.index_video(path))
    assert c["slices_clean_end"] == c["slices"] > 0 and len(frames) == len(gold)
    for i, f in enumerate(frames):
        assert np.array_equal(f[0], gold[i]), "frame %d differs" % i
    g = golden_md5(name)
    assert hashlib.md5(open(path, "rb").read()).hexdigest() == g["mp4_md5"]
    assert [plane_md5s(f[0], w, h) for f in frames] == g["frames"]


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


def test_corpus_reaches_every_reachable_case(hw, oracle, tmp_path):
    """The coverage histogram of the oracle over the whole MATRIX against tests/coverage_model.py: every intra mode x
    availability pattern, every luma / chroma fractional position, weight mode, bS x plane x edge kind, deblocking
    filter branch, mb_type / sub_mb_type per slice type, coeff_token / total_zeros / run_before table row, escape
    path, dequantisation branch, direct mode, and both values of every CABAC context index per slice type and
    cabac_init_idc that a progressive 4:2:0 stream can reach IS reached - and nothing the model calls unreachable is.
    (Cases decoded by test_roundtrip in this process are reused; the rest are decoded here.)"""
    from tests.coverage_model import reachable
    total = np.zeros(oracle.COV_CELLS, dtype=np.int64)
    for name, kw in MATRIX:
        if name not in _COVERAGE:
            kw = dict(kw)
            w, h = kw.pop("width", 322 if name == "qp40" else 320), kw.pop("height", 242 if name == "qp40" else 240)
            path = str(tmp_path / (name + ".mp4"))
            make_clip(path, width=w, height=h, yuv=False, **kw)
            cov = np.zeros(oracle.COV_CELLS, dtype=np.int64)
            oracle.decode_file(path, hw.index_video(path), coverage=cov)
            os.unlink(path)
            _COVERAGE[name] = cov
        total += _COVERAGE[name]
    model = reachable()
    missing, unexpected = {}, {}
    for sec, (base, n) in oracle.COV_SECTIONS.items():
        hit = set(np.nonzero(total[base:base + n])[0].tolist())
        if model[sec] - hit:
            missing[sec] = sorted(model[sec] - hit)
        if hit - model[sec]:
            unexpected[sec] = sorted(hit - model[sec])
    assert not unexpected, "cells the model calls unreachable were reached: %r" % unexpected
    assert not missing, "reachable cells no corpus case reaches: %r" % missing
    # the committed summary (tests/golden/corpus_coverage.json) lists exactly the unreachable cells
    with open(os.path.join(ROOT, "tests", "golden", "corpus_coverage.json")) as f:
        committed = json.load(f)["sections"]
    for sec, (base, n) in oracle.COV_SECTIONS.items():
        assert set(committed[sec]["unreached"]) == set(range(n)) - model[sec], sec

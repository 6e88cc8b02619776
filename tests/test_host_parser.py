"""Host half of the MI355X backend on CPU (hwang_h264_probe: entropy decode,
MV derivation, DPB, output order, sparse skipping - no GPU involved).

* every slice of every corpus configuration ends exactly on
  rbsp_trailing_bits (a CABAC/CAVLC desync cannot hide), one output per sample;
* slice / picture counts agree with the oracle decoding the same file;
* the digest of all records the kernels would receive equals
  tests/golden/parser_digest.json, frozen from a build whose GPU output was
  bit-exact against the oracle (regenerate with
  `python -m tests.test_host_parser` only after `pytest -m gpu` is green);
* sparse hints skip exactly the unwanted non-reference pictures."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from tests.corpus_util import MATRIX, make_clip
from tests.util import REALSHORT, ROOT, have_file

GOLDEN = os.path.join(ROOT, "tests", "golden", "parser_digest.json")


def probe(hw, path, wanted=None):
    from hwang_amd._capi import lib
    idx = hw.index_video(path)
    data = np.fromfile(path, dtype=np.uint8)
    offs = np.ascontiguousarray(idx.sample_offsets_np().astype(np.uint64))
    sizes = np.ascontiguousarray(idx.sample_sizes_np().astype(np.uint64))
    avcc = bytes(idx.metadata_bytes())
    out = (C.c_int64 * 8)()
    err = C.create_string_buffer(256)
    w = np.asarray(wanted if wanted is not None else [], dtype=np.uint64)
    rc = lib.hwang_h264_probe(avcc, len(avcc), data.ctypes.data_as(C.POINTER(C.c_uint8)),
                              offs.ctypes.data_as(C.POINTER(C.c_uint64)), sizes.ctypes.data_as(C.POINTER(C.c_uint64)),
                              idx.frames(), w.ctypes.data_as(C.POINTER(C.c_uint64)) if len(w) else None, len(w), out,
                              err, 256)
    assert rc == 0, err.value
    keys = ("slices", "clean_end", "pictures", "skipped", "outputs", "outputs_skipped", "coefficients", "digest")
    return idx, dict(zip(keys, list(out)))


def probe_workers(hw, path, workers, wanted=None, keyframes=None):
    """The same through hwang_h264_probe_workers: slice data decoded by `workers` entropy threads (0: serial)."""
    from hwang_amd._capi import lib
    idx = hw.index_video(path)
    data = np.fromfile(path, dtype=np.uint8)
    offs = np.ascontiguousarray(idx.sample_offsets_np().astype(np.uint64))
    sizes = np.ascontiguousarray(idx.sample_sizes_np().astype(np.uint64))
    avcc = bytes(idx.metadata_bytes())
    out = (C.c_int64 * 9)()
    err = C.create_string_buffer(256)
    w = np.asarray(wanted if wanted is not None else [], dtype=np.uint64)
    kf = np.asarray(keyframes if keyframes is not None else [], dtype=np.uint64)
    rc = lib.hwang_h264_probe_workers(avcc, len(avcc), data.ctypes.data_as(C.POINTER(C.c_uint8)),
                                      offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                                      sizes.ctypes.data_as(C.POINTER(C.c_uint64)), idx.frames(),
                                      w.ctypes.data_as(C.POINTER(C.c_uint64)) if len(w) else None, len(w),
                                      kf.ctypes.data_as(C.POINTER(C.c_uint64)) if len(kf) else None, len(kf), workers,
                                      out, err, 256)
    assert rc == 0, err.value
    keys = ("slices", "clean_end", "pictures", "skipped", "outputs", "outputs_skipped", "coefficients", "digest",
            "digest_noslot")
    return dict(zip(keys, list(out)))


def golden():
    return json.load(open(GOLDEN)) if os.path.exists(GOLDEN) else {}


@pytest.mark.parametrize("name,kw", MATRIX, ids=[m[0] for m in MATRIX])
def test_corpus_clip(hw, oracle, tmp_path, name, kw):
    path = str(tmp_path / (name + ".mp4"))
    make_clip(path, yuv=False, **kw)
    idx, r = probe(hw, path)
    assert r["slices"] == r["clean_end"] > 0
    assert r["pictures"] == r["outputs"] == idx.frames() and r["skipped"] == 0
    _, c = oracle.decode_file(path, idx)
    assert c["slices"] == r["slices"] and c["pictures"] == r["pictures"]
    assert probe(hw, path)[1]["digest"] == r["digest"]  # deterministic
    g = golden()
    assert name in g, "no golden digest for %s (run python -m tests.test_host_parser)" % name
    assert g[name] == {k: r[k] for k in ("slices", "coefficients", "digest")}


@pytest.mark.parametrize("name,kw", MATRIX, ids=[m[0] for m in MATRIX])
def test_entropy_workers_produce_the_same_records(hw, tmp_path, name, kw):
    """num_devices > 1 (the reference's decoder thread count, software_video_decoder.cpp:148): consecutive pictures'
    slice data decoded by a pool of workers while the calling thread runs ahead through the headers.  Every record
    the GPU would receive, in hand-over order, is the same as in the serial decode - for 3 workers on every case, for 1
    and 8 on every fourth, with and without a sparse hint - and the serial digest is the committed golden one."""
    path = str(tmp_path / (name + ".mp4"))
    make_clip(path, yuv=False, **kw)
    serial = probe_workers(hw, path, 0)
    assert serial["digest"] == golden()[name]["digest"]
    n = serial["outputs"]
    sparse = sorted(set([0, n // 3, n // 2, n - 1]))
    serial_sparse = probe_workers(hw, path, 0, wanted=sparse)
    # three workers for every case; one and eight as well for every fourth (the suite's time)
    every = [m[0] for m in MATRIX].index(name) % 4 == 0
    for workers in ((1, 3, 8) if every else (3,)):
        r = probe_workers(hw, path, workers)
        for k in ("slices", "clean_end", "pictures", "skipped", "outputs", "outputs_skipped", "coefficients",
                  "digest_noslot"):
            assert r[k] == serial[k], (workers, k)
        rs = probe_workers(hw, path, workers, wanted=sparse)
        for k in ("slices", "pictures", "skipped", "outputs", "outputs_skipped", "coefficients", "digest_noslot"):
            assert rs[k] == serial_sparse[k], (workers, k, "sparse")


def test_sparse_hint_skips_only_unwanted_non_reference(hw, tmp_path):
    path = str(tmp_path / "ibp.mp4")
    make_clip(path, yuv=False, frames=13, gop=13, bframes=2, cabac=1, t8x8=1, refs=2)
    idx, full = probe(hw, path)
    # display order I B B P B B P ...: wanting only P/I positions skips all 8 B pictures
    idx, r = probe(hw, path, wanted=[0, 3, 6, 12])
    assert r["outputs"] == idx.frames()
    assert r["skipped"] == r["outputs_skipped"] == 8
    assert r["pictures"] == full["pictures"] - 8 and r["slices"] == r["clean_end"]
    # wanting a B picture decodes that one
    _, r2 = probe(hw, path, wanted=[1, 12])
    assert r2["skipped"] == 7


def test_sparse_hint_ends_the_interval_early(hw, tmp_path):
    """Nothing behind the last wanted position + the reorder depth is decoded: the reference's automata feeds every
    sample of a touched interval (decoder_automata.cpp:287-331), and each of them still yields one output position."""
    path = str(tmp_path / "tail.mp4")
    make_clip(path, yuv=False, frames=40, gop=40, bframes=2, cabac=1, refs=2)   # I B B P B B P ...; num_reorder_frames 1
    idx, full = probe(hw, path)
    assert full["pictures"] == 40
    # wanted: display position 4 (a B picture between P3 and P6).  Decoding order I0 P3 B1 B2 P6 B4 B5 P9 ...:
    # I0, P3, P6 and B4 are needed; decoding positions 0 .. 4 + 1 are looked at, everything later is void.
    _, r = probe(hw, path, wanted=[4])
    assert r["outputs"] == 40 and r["slices"] == r["clean_end"]
    assert r["pictures"] == 4 and r["skipped"] == 36
    for workers in (1, 4):
        rw = probe_workers(hw, path, workers, wanted=[4])
        assert (rw["pictures"], rw["skipped"], rw["outputs"]) == (4, 36, 40)
    # the first frame only: the IDR picture, and (reorder depth 1) the P picture behind it
    _, r0 = probe(hw, path, wanted=[0])
    assert r0["pictures"] == 2 and r0["outputs"] == 40
    # an interval of three GOPs (the planner merges byte-adjacent GOPs): with the sync samples known every GOP ends
    # behind its own last wanted position, and a GOP nobody wants anything of is not touched at all
    path3 = str(tmp_path / "tail3.mp4")
    make_clip(path3, yuv=False, frames=36, gop=12, bframes=2, cabac=1, refs=2)
    for workers in (0, 3):
        whole = probe_workers(hw, path3, workers, wanted=[4, 27])
        cut = probe_workers(hw, path3, workers, wanted=[4, 27], keyframes=[0, 12, 24])
        assert whole["outputs"] == cut["outputs"] == 36
        # without: every reference picture up to decoding position 27 + 1, and B4
        assert whole["pictures"] == 14
        # with: I0 P3 P6 B4 | nothing of the second GOP | I24 P27 and (decoding position 28 = 27 + the reorder depth) P30
        assert cut["pictures"] == 7


def test_sparse_hint_keeps_reference_b_pictures(hw, tmp_path):
    path = str(tmp_path / "pyr.mp4")
    make_clip(path, yuv=False, frames=13, gop=13, bframes=3, refs=4, bpyramid=1)
    idx, full = probe(hw, path)
    # display order I b B b P b B b P b B b P: wanting the anchors only skips the six non-reference b pictures,
    # never the three middle ones (references)
    _, r = probe(hw, path, wanted=[0, 4, 8, 12])
    assert r["skipped"] == 6 and r["pictures"] == full["pictures"] - 6 and r["slices"] == r["clean_end"]


@pytest.mark.skipif(not have_file(REALSHORT), reason="realshort.mp4 not in this image")
def test_real_x264_stream(hw):
    idx, r = probe(hw, REALSHORT)
    assert r["slices"] == r["clean_end"] == r["pictures"] == idx.frames() == 36


if __name__ == "__main__":  # regenerate the golden digests
    import tempfile

    import hwang_amd
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, kw in MATRIX:
            p = os.path.join(tmp, name + ".mp4")
            make_clip(p, yuv=False, **kw)
            r = probe(hwang_amd, p)[1]
            out[name] = {k: r[k] for k in ("slices", "coefficients", "digest")}
    json.dump(out, open(GOLDEN, "w"), indent=1, sort_keys=True)
    print("wrote", GOLDEN)

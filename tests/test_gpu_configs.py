"""-m gpu: BASELINE.json's configurations as parity cases on one GPU
(VERDICT r1 weak #2).  C1 is bench.py's workload (and
test_gpu_decode.py::test_sparse_equals_sequential_at_scale); here:

C2  3840x2160 High, several GOPs, rows = every 10th frame, 8 lanes, non-reference
    B pictures skipped - every returned frame against the encoder's
    reconstruction (== the oracle's decode, tests/test_corpus_roundtrip.py);
C3  the batch shape: many clips (different seeds, two sizes), a 1 %-style
    sample of each, ONE retrieve_batch call whose GOP items share the lanes -
    each frame against its clip's reconstruction (8 GPUs need a node; the
    pooling, the per-request output slices and the sharding are what one GPU
    can pin);
C4  one of the eight 4K CABAC high-bitrate streams (QP 18), every frame,
    2 GOPs against the reconstruction, then sparse == dense == serial over 10
    GOPs (the reference's GatherFramesComparison invariant)."""
import hashlib

import numpy as np
import pytest

from tests.corpus_util import make_clip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; there is no fallback"
    return torch


def md5s(frames):
    return [hashlib.md5(np.ascontiguousarray(f).tobytes()).hexdigest() for f in frames]


def test_c2_4k_stride10_multi_gop(hw, oracle, gpu, tmp_path):
    path = str(tmp_path / "c2.mp4")
    gold = make_clip(path, width=3840, height=2160, frames=60, gop=12, bframes=2, threads=5)
    idx = hw.index_video(path)
    assert idx.frames() == 60 and len(idx.keyframe_indices()) == 5
    rows = list(range(0, 60, 10))
    dec = hw.Decoder(path, video_index=idx, num_lanes=8)
    got = dec.retrieve_yuv(rows)
    st = dec.stats()
    assert st["frames_returned"] == len(rows) and st["intervals"] == 5
    assert st["pictures_skipped"] > 0, "unwanted non-reference B pictures must not be reconstructed"
    for k, r in enumerate(rows):
        assert np.array_equal(got[k], gold[r]), "4K frame %d differs from the reconstruction" % r
    rgb = dec.retrieve(rows[:2])
    for k in range(2):
        assert np.array_equal(rgb[k], oracle.i420_to_rgb(gold[rows[k]], 3840, 2160))
    # whatever the lane count, and through the reference-shaped serial loop
    assert md5s(hw.Decoder(path, video_index=idx, num_lanes=2).retrieve_yuv(rows)) == md5s(got)
    assert md5s(hw.Decoder(path, video_index=idx, num_lanes=1).retrieve_serial(rows[:3])) == md5s(rgb + dec.retrieve([rows[2]]))


def test_c3_batch_of_clips(hw, oracle, gpu, tmp_path):
    from hwang_amd.decoder import BatchRetriever
    rng = np.random.default_rng(20240601)
    clips = []
    for k in range(12):
        w, h = (320, 240) if k % 3 else (352, 288)
        path = str(tmp_path / ("c3_%d.mp4" % k))
        gold = make_clip(path, width=w, height=h, frames=90, gop=15, bframes=2, seed=100 + k, refs=2 + k % 2,
                         cabac=1 if k % 4 else 0)
        rows = sorted(rng.choice(90, 4, replace=False).tolist())
        clips.append((hw.index_video(path), path, rows, gold, w, h))
    br = BatchRetriever(device_ids=[0], num_lanes=6)
    got = br.retrieve([(c[0], c[1], c[2]) for c in clips])
    for k, (idx, path, rows, gold, w, h) in enumerate(clips):
        assert len(got[k]) == len(rows)
        for i, r in enumerate(rows):
            assert got[k][i].shape == (h, w, 3)
            assert np.array_equal(got[k][i], oracle.i420_to_rgb(gold[r], w, h)), "clip %d row %d" % (k, r)
    st = br.stats()
    assert st["frames_returned"] == 48 and st["intervals"] >= 12
    # a 3-way sharded job returns the same frames, each exactly once
    seen = [[0] * 4 for _ in clips]
    for rank in range(3):
        part = BatchRetriever(device_ids=[0], num_lanes=3, shard=(rank, 3)).retrieve([(c[0], c[1], c[2]) for c in clips])
        for k in range(len(clips)):
            for i in range(4):
                if part[k][i].any():
                    assert np.array_equal(part[k][i], got[k][i])
                    seen[k][i] += 1
    assert all(s == [1, 1, 1, 1] for s in seen), seen


def test_c4_4k_high_bitrate_dense(hw, oracle, gpu, tmp_path):
    path = str(tmp_path / "c4.mp4")
    gold = make_clip(path, width=3840, height=2160, frames=16, gop=8, bframes=2, qp=18, fps=60, threads=2)
    idx = hw.index_video(path)
    dec = hw.Decoder(path, video_index=idx, num_lanes=2)
    got = dec.retrieve_yuv(list(range(16)))
    assert dec.stats()["pictures_skipped"] == 0
    for r in range(16):
        assert np.array_equal(got[r], gold[r]), "4K QP18 frame %d" % r
    del gold, got
    # sparse == dense == serial at 10 GOPs
    path = str(tmp_path / "c4long.mp4")
    make_clip(path, width=3840, height=2160, frames=60, gop=6, bframes=2, qp=18, fps=60, threads=10, yuv=False)
    idx = hw.index_video(path)
    rows = sorted(np.random.default_rng(4).choice(60, 12, replace=False).tolist())
    dense = hw.Decoder(path, video_index=idx, num_lanes=8).retrieve_yuv(list(range(60)))
    dense = md5s([dense[r] for r in rows])
    sparse = md5s(hw.Decoder(path, video_index=idx, num_lanes=5).retrieve_yuv(rows))
    assert sparse == dense
    serial = hw.Decoder(path, video_index=idx, num_lanes=1).retrieve_serial(rows[:4])
    rgb = hw.Decoder(path, video_index=idx, num_lanes=3).retrieve(rows[:4])
    assert md5s(serial) == md5s(rgb)

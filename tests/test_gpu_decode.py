"""-m gpu: whole-picture parity of the MI355X backend against the CPU oracle,
through the C ABI (libhwang_amd.so: host entropy decode + HIP kernels).  The
bar is bit-exact I420 planes (H.264 reconstruction is integer arithmetic) and
bit-exact RGB against the oracle's statement of the colour conversion."""
import os

import numpy as np
import pytest

from tests.corpus_util import BAND_SIZES, MATRIX, golden_md5, make_clip, plane_md5s
from tests.util import REALSHORT, have_file

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; there is no fallback"
    return torch


def oracle_frames(oracle, hw, path):
    idx = hw.index_video(path)
    frames, counters = oracle.decode_file(path, idx)
    return idx, frames


@pytest.mark.parametrize("name,kw", MATRIX, ids=[m[0] for m in MATRIX])
def test_all_frames_bit_exact(hw, oracle, gpu, tmp_path, name, kw):
    kw = dict(kw)
    w, h = kw.pop("width", 322 if name == "qp40" else 320), kw.pop("height", 242 if name == "qp40" else 240)
    path = str(tmp_path / (name + ".mp4"))
    gold = make_clip(path, width=w, height=h, **kw)
    idx = hw.index_video(path)
    dec = hw.Decoder(path, video_index=idx, num_lanes=2)
    got = dec.retrieve_yuv(list(range(idx.frames())))
    assert len(got) == len(gold)
    for i in range(len(gold)):
        assert np.array_equal(got[i], gold[i]), "frame %d differs from the oracle/encoder reconstruction" % i
    # ... and from the digests committed in tests/golden/corpus_md5.json (frozen oracle output)
    assert [plane_md5s(f, w, h) for f in got] == golden_md5(name)["frames"]


@pytest.mark.parametrize("name,w,h,kw", BAND_SIZES, ids=[b[0] for b in BAND_SIZES])
def test_band_edge_sizes_bit_exact(hw, gpu, tmp_path, name, w, h, kw):
    path = str(tmp_path / (name + ".mp4"))
    gold = make_clip(path, width=w, height=h, **kw)
    idx = hw.index_video(path)
    dec = hw.Decoder(path, video_index=idx, num_lanes=2)
    got = dec.retrieve_yuv(list(range(idx.frames())))
    assert len(got) == len(gold)
    for i in range(len(gold)):
        assert np.array_equal(got[i], gold[i]), "%s: frame %d differs from the encoder's reconstruction" % (name, i)
    assert [plane_md5s(f, w, h) for f in got] == golden_md5(name)["frames"]


@pytest.mark.skipif(not have_file(REALSHORT), reason="realshort.mp4 not in this image")
def test_real_x264_stream_bit_exact(hw, oracle, gpu):
    idx, ref = oracle_frames(oracle, hw, REALSHORT)
    dec = hw.Decoder(REALSHORT, video_index=idx, num_lanes=1)
    got = dec.retrieve_yuv(list(range(36)))
    for i in range(36):
        assert np.array_equal(got[i], ref[i][0]), "frame %d" % i


def test_sparse_rows_and_rgb(hw, oracle, gpu, tmp_path):
    """Decoder.retrieve(rows): k-th array == frame rows[k] of a sequential
    decode (the reference's GatherFramesComparison invariant), RGB."""
    path = str(tmp_path / "sparse.mp4")
    gold = make_clip(path, frames=48, gop=12, bframes=2, slices=2)
    idx = hw.index_video(path)
    rows = [0, 1, 5, 11, 12, 13, 30, 35, 47]
    dec = hw.Decoder(path, video_index=idx, num_lanes=3)
    got = dec.retrieve(rows)
    st = dec.stats()
    assert st["frames_returned"] == len(rows)
    assert st["pictures_skipped"] > 0, "unwanted non-reference pictures should not be reconstructed"
    for k, r in enumerate(rows):
        want = oracle.i420_to_rgb(gold[r], 320, 240)
        assert got[k].shape == (240, 320, 3)
        assert np.array_equal(got[k], want), "row %d" % r
    # same through the reference-shaped serial path (DecoderAutomata)
    serial = dec.retrieve_serial(rows)
    for k, r in enumerate(rows):
        assert np.array_equal(serial[k], got[k]), "serial row %d" % r


def test_sparse_rows_with_reference_b_pictures(hw, oracle, gpu, tmp_path):
    """b-pyramid: the middle B picture of every group is a reference and must be reconstructed for its neighbours
    even when it is not wanted itself; only the non-reference ones may be skipped."""
    path = str(tmp_path / "pyr.mp4")
    gold = make_clip(path, frames=52, gop=13, bframes=3, refs=4, bpyramid=1, temporal_direct=1, implicit_wp=1)
    idx = hw.index_video(path)
    rows = [1, 3, 6, 13, 17, 27, 31, 45, 51]   # mostly non-reference B positions, whose prediction needs the middle one
    dec = hw.Decoder(path, video_index=idx, num_lanes=2)
    got = dec.retrieve_yuv(rows)
    assert dec.stats()["pictures_skipped"] > 0
    for k, r in enumerate(rows):
        assert np.array_equal(got[k], gold[r]), "row %d" % r
    serial = dec.retrieve_serial(rows)
    rgb = dec.retrieve(rows)
    for k, r in enumerate(rows):
        assert np.array_equal(serial[k], rgb[k]), "serial row %d" % r


def test_sync_samples_that_are_not_idr_pictures_do_not_end_a_gop(hw, oracle, gpu, tmp_path):
    """A sync-sample table may list random-access points that are not IDR pictures (x264's open-GOP mode lists
    recovery-point I pictures in `stss`).  What lies behind such a sample may refer across it, so neither the per-GOP
    cut behind the last wanted frame (automata path) nor the per-GOP work items (retriever path) may treat it as the
    start of something self-contained.  Here the table of a closed-GOP clip is given extra entries in the
    middle of its GOPs - P / B pictures, which certainly refer backwards: a planner interval that starts at a real
    IDR picture and runs across such an entry must still decode every wanted frame exactly."""
    path = str(tmp_path / "fakesync.mp4")
    gold = make_clip(path, frames=48, gop=24, bframes=2, refs=2)
    idx = hw.index_video(path)
    assert idx.keyframe_indices() == [0, 24]
    lying = hw.VideoIndex(timescale=idx.timescale(), duration=idx.duration(), width=idx.frame_width(),
                          height=idx.frame_height(), format=idx.format(), sample_offsets=idx.sample_offsets(),
                          sample_sizes=idx.sample_sizes(), keyframe_indices=[0, 12, 24, 37], metadata=idx.metadata_bytes())
    rows = [5, 20, 30, 44]   # 20 and 44 lie behind a fake sync sample; their intervals start at 0 and 24
    plan = hw.slice_into_video_intervals(lying, rows)
    assert [tuple(iv) for iv, _ in plan] == [(0, 48)]   # (byte-adjacent GOPs: one interval with four "sync samples")
    dec = hw.Decoder(path, video_index=lying, num_lanes=2)
    got = dec.retrieve_yuv(rows)
    for k, r in enumerate(rows):
        assert np.array_equal(got[k], gold[r]), "retriever path, row %d" % r
    rgb = dec.retrieve(rows)
    serial = dec.retrieve_serial(rows)
    for k, r in enumerate(rows):
        assert np.array_equal(serial[k], rgb[k]), "automata path, row %d" % r
        assert np.array_equal(rgb[k], oracle.i420_to_rgb(gold[r], 320, 240)), "row %d" % r


def test_device_resident_output(hw, oracle, gpu, tmp_path):
    torch = gpu
    path = str(tmp_path / "dev.mp4")
    gold = make_clip(path, frames=10, gop=10, bframes=1)
    idx = hw.index_video(path)
    rows = [2, 3, 9]
    out = torch.zeros((len(rows), 240, 320, 3), dtype=torch.uint8, device="cuda")
    dec = hw.Decoder(path, video_index=idx, num_lanes=1)
    dec.retrieve_into(rows, out.data_ptr(), out.numel())
    torch.cuda.synchronize()
    host = out.cpu().numpy()
    for k, r in enumerate(rows):
        assert np.array_equal(host[k], oracle.i420_to_rgb(gold[r], 320, 240))


def test_page_locked_host_output(hw, oracle, gpu, tmp_path):
    """A caller's buffer in page-locked host memory receives the D2H copy itself (no bounce buffer's host half, no
    memcpy): the same bytes as a pageable buffer gets - through one decoder and through the lanes of a batched call,
    and at an offset into the allocation (the pointer handed over is not the allocation's base)."""
    torch = gpu
    path = str(tmp_path / "pin.mp4")
    gold = make_clip(path, frames=24, gop=12, bframes=2)
    idx = hw.index_video(path)
    rows = [1, 2, 5, 11, 12, 17, 23]
    frame = 240 * 320 * 3
    pinned = torch.zeros((len(rows) + 1) * frame, dtype=torch.uint8, pin_memory=True)
    assert pinned.is_pinned()
    dec = hw.Decoder(path, video_index=idx, num_lanes=2)
    dec.retrieve_into(rows, pinned.data_ptr() + frame, len(rows) * frame)
    got = pinned.numpy()[frame:].reshape(len(rows), 240, 320, 3)
    assert not pinned.numpy()[:frame].any()
    for k, r in enumerate(rows):
        assert np.array_equal(got[k], oracle.i420_to_rgb(gold[r], 320, 240)), "row %d" % r
    pageable = np.zeros((len(rows), 240, 320, 3), dtype=np.uint8)
    br = hw.BatchRetriever(device_ids=[0], num_lanes=4)
    p2 = torch.zeros(len(rows) * frame, dtype=torch.uint8, pin_memory=True)
    with open(path, "rb") as f:
        br.retrieve_into([(idx, f, rows, pageable.ctypes.data, pageable.nbytes), (idx, f, rows, p2.data_ptr(), p2.numel())])
    assert np.array_equal(pageable, got)
    assert np.array_equal(p2.numpy().reshape(pageable.shape), got)


def test_1080p_cropping(hw, oracle, gpu, tmp_path):
    path = str(tmp_path / "hd.mp4")
    gold = make_clip(path, width=1920, height=1080, frames=5, gop=5, bframes=2, threads=4)
    idx = hw.index_video(path)
    dec = hw.Decoder(path, video_index=idx, num_lanes=1)
    got = dec.retrieve_yuv([0, 2, 4])
    for k, r in enumerate([0, 2, 4]):
        assert np.array_equal(got[k], gold[r])


def test_slice_count_limit(hw, gpu, tmp_path):
    """255 slices per picture (one macroblock each) decode bit-exactly; 300 are refused by name, loudly."""
    path = str(tmp_path / "s255.mp4")
    gold = make_clip(path, width=240, height=272, frames=5, gop=5, bframes=1, refs=2, slice_mbs=1, deblock=2)
    idx = hw.index_video(path)
    got = hw.Decoder(path, video_index=idx, num_lanes=1).retrieve_yuv(list(range(5)))
    for i in range(5):
        assert np.array_equal(got[i], gold[i]), "frame %d" % i
    many = str(tmp_path / "s300.mp4")
    make_clip(many, yuv=False, frames=3, gop=3, bframes=0, slice_mbs=1)
    with pytest.raises(RuntimeError, match="too many slices"):
        hw.Decoder(many, video_index=hw.index_video(many), num_lanes=1).retrieve_yuv([0, 1, 2])


def test_left_top_cropping_rgb(hw, oracle, gpu, tmp_path):
    """The cropping rectangle starts inside the coded picture: the conversion reads from the offset surface."""
    path = str(tmp_path / "crop.mp4")
    gold = make_clip(path, width=306, height=226, frames=7, gop=7, bframes=1, crop_left=5, crop_top=7)
    idx = hw.index_video(path)
    assert (idx.frame_width(), idx.frame_height()) == (306, 226)
    dec = hw.Decoder(path, video_index=idx, num_lanes=1)
    got = dec.retrieve([0, 3, 6])
    for k, r in enumerate([0, 3, 6]):
        assert got[k].shape == (226, 306, 3)
        assert np.array_equal(got[k], oracle.i420_to_rgb(gold[r], 306, 226)), "row %d" % r


def test_1080p_everything_at_once(hw, oracle, gpu, tmp_path):
    """Full-size pictures (17 bands per row kernel) with the syntax variety of the small corpus cases combined:
    reference B pictures, slices of mixed types cut in the middle of rows, two parameter sets that switch the
    entropy coder, rate control, explicit weights, list modification."""
    path = str(tmp_path / "hd_all.mp4")
    gold = make_clip(path, width=1920, height=1080, frames=18, gop=9, bframes=3, refs=4, bpyramid=1, mixed_slices=1,
                     slice_mbs=1777, multi_pps=2, rate_kbit=900, explicit_wp=1, wp_denom=5, rplm=1, threads=4)
    idx = hw.index_video(path)
    dec = hw.Decoder(path, video_index=idx, num_lanes=2)
    rows = list(range(idx.frames()))
    got = dec.retrieve_yuv(rows)
    for r in rows:
        assert np.array_equal(got[r], gold[r]), "frame %d" % r


def test_many_lanes_random_rows_1080p(hw, gpu, tmp_path):
    """The concurrent machinery end to end: 8 lanes x 2 decoders, intra-only and
    P/B engines side by side, LRU surfaces, per-GOP work items - random rows of a
    4-GOP 1080p clip, I420 and RGB, twice (state carried across retrieves),
    against the encoder's reconstruction (== oracle, tests/test_corpus_roundtrip)."""
    from oracle import oracle_py
    path = str(tmp_path / "hd4.mp4")
    gold = make_clip(path, width=1920, height=1080, frames=96, gop=24, bframes=2, threads=4)
    idx = hw.index_video(path)
    dec = hw.Decoder(path, video_index=idx, num_lanes=8)
    rng = np.random.default_rng(7)
    for rep in range(2):
        rows = sorted(rng.choice(96, 24, replace=False).tolist())
        got = dec.retrieve_yuv(rows)
        for k, r in enumerate(rows):
            assert np.array_equal(got[k], gold[r]), "pass %d frame %d" % (rep, r)
    rows = [0, 23, 24, 47, 95]
    rgb = dec.retrieve(rows)
    for k, r in enumerate(rows):
        assert np.array_equal(rgb[k], oracle_py.i420_to_rgb(gold[r], 1920, 1080))
    st = dec.stats()
    assert st["pictures_skipped"] > 0 and st["frames_returned"] == len(rows)


def test_sparse_equals_sequential_at_scale(hw, gpu, tmp_path):
    """The reference's only results test (GatherFramesComparison,
    decoder_automata_test.cpp:262-342) at 1080p over 10 GOPs: frames gathered
    sparsely == the same frames out of a dense decode, whatever the path
    (parallel retriever with sparse hints and B skipping, other lane count,
    the reference-shaped serial DecoderAutomata loop)."""
    import hashlib
    path = str(tmp_path / "long.mp4")
    make_clip(path, width=1920, height=1080, frames=300, gop=30, bframes=2, threads=16, yuv=False)
    idx = hw.index_video(path)
    rows = sorted(np.random.default_rng(3).choice(300, 24, replace=False).tolist())
    md5 = lambda frames: [hashlib.md5(f.tobytes()).hexdigest() for f in frames]
    sparse16 = md5(hw.Decoder(path, video_index=idx, num_lanes=16).retrieve(rows))
    sparse3 = md5(hw.Decoder(path, video_index=idx, num_lanes=3).retrieve(rows))
    serial = md5(hw.Decoder(path, video_index=idx, num_lanes=1).retrieve_serial(rows))
    dense = hw.Decoder(path, video_index=idx, num_lanes=8).retrieve(list(range(300)))
    dense = md5([dense[r] for r in rows])
    assert sparse16 == dense and sparse3 == dense and serial == dense


def test_wide_batches_are_deterministic(hw, gpu, tmp_path):
    """The row kernels hand rows over between workgroups through global memory.
    A hand-off that is merely *usually* in time passes every small parity case,
    so: one sparse request decoded with a single lane (one picture per launch,
    nothing concurrent) and then repeatedly with many lanes (wide batches, all
    engines busy) must return identical frames every time.  (This is the test
    that caught a counter published before the last macroblock's samples.)"""
    torch = gpu
    path = str(tmp_path / "soak.mp4")
    make_clip(path, width=1920, height=1080, frames=300, gop=60, bframes=2, threads=16, repeat=12, yuv=False)
    idx = hw.index_video(path)
    n = idx.frames()
    rows = sorted(np.random.default_rng(7).choice(n, 120, replace=False).tolist())
    out = torch.empty((len(rows), 1080, 1920, 3), dtype=torch.uint8, device="cuda")

    def sums():
        return (out.to(torch.int64).sum(dim=(1, 2, 3)).cpu().numpy(),
                (out[:, ::7, ::5, :].to(torch.int64) * 31).sum(dim=(1, 2, 3)).cpu().numpy())

    dec = hw.Decoder(path, video_index=idx, num_lanes=1)
    dec.retrieve_into(rows, out.data_ptr(), out.numel())
    ref = sums()
    del dec
    for lanes in (48, 24):
        dec = hw.Decoder(path, video_index=idx, num_lanes=lanes)
        for rnd in range(4):
            out.zero_()
            dec.retrieve_into(rows, out.data_ptr(), out.numel())
            got = sums()
            bad = np.nonzero((got[0] != ref[0]) | (got[1] != ref[1]))[0]
            assert bad.size == 0, "lanes %d round %d: frames %s differ from the single-lane decode" % (lanes, rnd, bad[:8].tolist())
        del dec
    # ... and the way long queues are served (round 5: a thread per CPU driving several lanes in turn, engines that
    # linger for company, keyframes predicted on the heavy engine and deblocked in the light batches): the retriever
    # takes it from eight work items per lane up - here from one (HWANG_LONG_QUEUE_FROM), and with two or five lanes
    # per thread
    for lanes, share in ((48, None), (24, "2"), (40, "5")):
        os.environ["HWANG_LONG_QUEUE_FROM"] = "1"
        if share:
            os.environ["HWANG_LANES_PER_THREAD"] = share
        try:
            dec = hw.Decoder(path, video_index=idx, num_lanes=lanes)
            for rnd in range(3):
                out.zero_()
                dec.retrieve_into(rows, out.data_ptr(), out.numel())
                got = sums()
                bad = np.nonzero((got[0] != ref[0]) | (got[1] != ref[1]))[0]
                assert bad.size == 0, "long queue, lanes %d share %s round %d: frames %s differ from the single-lane decode" % (
                    lanes, share, rnd, bad[:8].tolist())
            del dec
        finally:
            os.environ.pop("HWANG_LONG_QUEUE_FROM", None)
            os.environ.pop("HWANG_LANES_PER_THREAD", None)


def test_4k_wide_pictures(hw, oracle, gpu, tmp_path):
    """BASELINE config C2 geometry (3840x2160 = 240 x 135 macroblocks): wider
    than the intra kernel's LDS line buffer, so intra-only pictures take the
    surface-neighbour path; 3840-byte pitches; I, P and B pictures."""
    path = str(tmp_path / "uhd.mp4")
    gold = make_clip(path, width=3840, height=2160, frames=4, gop=4, bframes=1, threads=4)
    idx = hw.index_video(path)
    frames, _ = oracle.decode_file(path, idx)
    dec = hw.Decoder(path, video_index=idx, num_lanes=1)
    got = dec.retrieve_yuv([0, 1, 2, 3])
    for r in range(4):
        assert np.array_equal(frames[r][0], gold[r]), "oracle != encoder reconstruction at frame %d" % r
        assert np.array_equal(got[r], gold[r]), "GPU != oracle at frame %d" % r


def test_errors_are_loud(hw, gpu, tmp_path):
    path = str(tmp_path / "e.mp4")
    make_clip(path, frames=4, gop=4, bframes=0, yuv=False)
    idx = hw.index_video(path)
    data = bytearray(open(path, "rb").read())
    off = idx.sample_offsets()[0]
    data[off + 40:off + 400] = b"\xff" * 360   # corrupt the IDR's slice data
    bad = str(tmp_path / "bad.mp4")
    open(bad, "wb").write(data)
    dec = hw.Decoder(bad, video_index=idx, num_lanes=1)
    with pytest.raises(RuntimeError):
        dec.retrieve_yuv([0, 1])


def test_expired_band_poll_fails_the_request(hw, gpu, tmp_path):
    """A row kernel whose bounded poll of the band above expires must fail the
    request, not hand back wrong samples.  HWANG_INJECT_FAULT
    makes k_deblock's exporting rows tag their progress words with a wrong epoch,
    which to the band below is a band that never reports.  The error is sticky
    for the process's executor, hence the child process."""
    import subprocess
    import sys
    from tests.util import ROOT
    path = str(tmp_path / "fault.mp4")
    make_clip(path, width=320, height=240, frames=3, gop=3, bframes=0, yuv=False)   # 15 rows = 4 bands
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import hwang_amd as hw\n"
        "dec = hw.Decoder(%r, num_lanes=1)\n"
        "try:\n"
        "    dec.retrieve_yuv([0, 1, 2])\n"
        "except RuntimeError as e:\n"
        "    print('RAISED', e)\n"
        "else:\n"
        "    print('NO ERROR')\n" % (ROOT, path))
    import os
    env = dict(os.environ, HWANG_INJECT_FAULT="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert "RAISED" in r.stdout and "hand-off timed out" in r.stdout and "k_deblock" in r.stdout, r.stdout + r.stderr
    # and the same clip decodes fine without the hook
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "NO ERROR" in r.stdout, r.stdout + r.stderr


FUZZ_TRIALS = int(os.environ.get("HWANG_FUZZ_TRIALS", "120"))
FUZZ_CLIPS = [
    ("plain", dict(frames=12, gop=6, bframes=2, refs=3, slices=2, seed=5)),
    ("pyramid_mixed", dict(frames=13, gop=13, bframes=3, refs=4, bpyramid=1, mixed_slices=1, slice_mbs=23, multi_pps=2,
                           temporal_direct=1, implicit_wp=1, seed=6)),
    ("cavlc_wp_mmco", dict(frames=14, gop=7, bframes=1, refs=4, cabac=0, explicit_wp=1, wp_denom=5, mmco=1, rplm=1, scaling=3,
                           extra_nals=1, seed=7)),
]


@pytest.mark.parametrize("name,kw", FUZZ_CLIPS, ids=[c[0] for c in FUZZ_CLIPS])
def test_corrupt_streams_never_hang_the_gpu(hw, gpu, tmp_path, name, kw):
    """Bit flips in slice data reach the kernels as absurd vectors, levels and modes (or are refused by the
    parser).  Either is fine; a hung or faulted GPU is not: every request returns, and the executor is still
    healthy afterwards (the pristine clip decodes bit-exactly through the same process)."""
    path = str(tmp_path / "fz.mp4")
    gold = make_clip(path, width=176, height=144, **kw)
    idx = hw.index_video(path)
    nf = idx.frames()
    data0 = np.fromfile(path, dtype=np.uint8)
    offs, sizes = idx.sample_offsets(), idx.sample_sizes()
    rng = np.random.default_rng(11)
    errors = decoded = 0
    for trial in range(FUZZ_TRIALS):
        data = data0.copy()
        for _ in range(1 + trial % 5):
            s = int(rng.integers(0, len(offs)))
            pos = offs[s] + int(rng.integers(6, sizes[s]))
            data[pos] ^= 1 << int(rng.integers(0, 8))
        bad = str(tmp_path / ("fz_%d.mp4" % trial))
        data.tofile(bad)
        dec = hw.Decoder(bad, video_index=idx, num_lanes=2)
        try:
            out = dec.retrieve_yuv(list(range(nf)))
            assert len(out) == nf
            decoded += 1
        except RuntimeError:
            errors += 1
        del dec
    assert errors + decoded == FUZZ_TRIALS
    got = hw.Decoder(path, video_index=idx, num_lanes=2).retrieve_yuv(list(range(nf)))
    for i in range(nf):
        assert np.array_equal(got[i], gold[i])

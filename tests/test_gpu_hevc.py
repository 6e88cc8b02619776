"""-m gpu: HEVC through the C ABI - host parser + k_hevc.hip - against the corpus encoder's
reconstruction, which the CPU oracle reproduces (tests/test_hevc_corpus.py).  The bar is bit-exact I420 and bit-exact
RGB against the oracle's statement of the colour conversion.  HEVC parity with FFmpeg is unpinned (no HEVC decoder or
foreign stream exists on either machine): what these tests pin is the HIP path against the oracle."""
import os

import numpy as np
import pytest

from tests.corpus_util import HEVC_MATRIX, make_hevc_clip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; there is no fallback"
    return torch


@pytest.mark.parametrize("name,kw", HEVC_MATRIX, ids=[m[0] for m in HEVC_MATRIX])
def test_all_frames_bit_exact(hw, oracle, gpu, tmp_path, name, kw):
    kw = dict(kw)
    w, h = kw.pop("width", 128), kw.pop("height", 96)
    path = str(tmp_path / (name + ".mp4"))
    gold = make_hevc_clip(path, width=w, height=h, **kw)
    idx = hw.index_video(path)
    assert idx.format() in ("hev1", "hvc1")
    dec = hw.Decoder(path, video_index=idx, num_lanes=2)
    got = dec.retrieve_yuv(list(range(idx.frames())))
    assert len(got) == len(gold)
    for i in range(len(gold)):
        assert np.array_equal(got[i], gold[i]), "%s: frame %d differs from the oracle / encoder reconstruction" % (name, i)


def test_sparse_rows_rgb_and_the_reference_api(hw, oracle, gpu, tmp_path):
    """Decoder.retrieve(rows): k-th array == frame rows[k] of a sequential decode (GatherFramesComparison's invariant,
    decoder_automata_test.cpp:262-342 - the reference's only enabled test runs it on an HEVC clip, :160-161), RGB;
    unwanted non-reference pictures are not reconstructed; the same through one DecoderAutomata (retrieve_serial) and
    through the pybind11 module's hwang.Decoder."""
    path = str(tmp_path / "sparse.mp4")
    gold = make_hevc_clip(path, width=176, height=144, frames=48, gop=12, bframes=2, refs=2, sao=1, ctb=32)
    idx = hw.index_video(path)
    rows = [0, 1, 5, 11, 12, 13, 30, 35, 47]
    dec = hw.Decoder(path, video_index=idx, num_lanes=3)
    got = dec.retrieve(rows)
    st = dec.stats()
    assert st["frames_returned"] == len(rows) and st["pictures_skipped"] > 0
    for k, r in enumerate(rows):
        assert got[k].shape == (144, 176, 3)
        assert np.array_equal(got[k], oracle.i420_to_rgb(gold[r], 176, 144)), "row %d" % r
    serial = dec.retrieve_serial(rows)
    for k in range(len(rows)):
        assert np.array_equal(serial[k], got[k]), "automata path, row %d" % rows[k]
    import hwang
    ref = hwang.Decoder(path, device_type=hwang.DeviceType.GPU).retrieve(rows)
    for k in range(len(rows)):
        assert np.array_equal(ref[k], got[k]), "hwang.Decoder, row %d" % rows[k]


def test_1080p_and_cropping(hw, oracle, gpu, tmp_path):
    """1920x1080 (coded 1920x1080: a multiple of 8, the last row of 64x64 blocks is cut at 56 rows), CTB 64, B pictures."""
    path = str(tmp_path / "hd.mp4")
    gold = make_hevc_clip(path, width=1920, height=1080, frames=4, gop=4, bframes=1, refs=2, ctb=64, sao=1, sdh=1, amp=1, qp=32)
    idx = hw.index_video(path)
    dec = hw.Decoder(path, video_index=idx, num_lanes=1)
    got = dec.retrieve_yuv([0, 1, 2, 3])
    for i in range(4):
        assert np.array_equal(got[i], gold[i]), "frame %d" % i


def test_2160p(hw, oracle, gpu, tmp_path):
    """3840x2160 (coded 3840x2160; the last row of 64x64 blocks is cut at 48 rows): 4 096-byte surface pitch, 2 040 coding
    tree blocks, staging sets above the size that gets five of them (three then), the kernels' grids at four times the
    bench clip's area."""
    path = str(tmp_path / "uhd.mp4")
    gold = make_hevc_clip(path, width=3840, height=2160, frames=3, gop=3, bframes=1, refs=2, ctb=64, sao=1, amp=1, qp=36)
    idx = hw.index_video(path)
    dec = hw.Decoder(path, video_index=idx, num_lanes=1)
    got = dec.retrieve_yuv([0, 1, 2])
    for i in range(3):
        assert np.array_equal(got[i], gold[i]), "frame %d" % i


def test_h264_after_hevc_on_one_decoder_object(hw, oracle, gpu, tmp_path):
    """configure() switches halves: the same automata decodes an HEVC clip, then an H.264 clip, then HEVC again."""
    from tests.corpus_util import make_clip
    p5 = str(tmp_path / "a.mp4")
    p4 = str(tmp_path / "b.mp4")
    g5 = make_hevc_clip(p5, frames=6, gop=6, bframes=1)
    g4 = make_clip(p4, width=128, height=96, frames=6, gop=6, bframes=1)
    for path, gold in ((p5, g5), (p4, g4), (p5, g5)):
        idx = hw.index_video(path)
        dec = hw.Decoder(path, video_index=idx, num_lanes=1)
        got = dec.retrieve_yuv([0, 2, 5])
        for k, r in enumerate([0, 2, 5]):
            assert np.array_equal(got[k], gold[r])


def test_engine_batches_pictures_of_many_decoders_and_sizes(hw, oracle, gpu, tmp_path):
    """The submission engine (hevc_engine.cpp) puts pictures of every HEVC decoder on the GPU into the same launches:
    two clips of different sizes and coding tree block sizes, short GOPs (an intra picture every few frames: heavy and
    light batches alternate for every decoder), 6 lanes each = 24 decoders, driven from two threads at once, three
    rounds - every frame of every round bit-exact, I420 (copied out plane by plane) and RGB (converted, through the
    page-locked bounce buffers into pageable arrays)."""
    import threading
    clips = []
    for name, kw in (("a", dict(width=176, height=144, frames=60, gop=6, bframes=2, refs=2, ctb=16, sao=1)),
                     ("b", dict(width=320, height=192, frames=48, gop=8, bframes=1, refs=2, ctb=64, sao=1, amp=1, sdh=1))):
        path = str(tmp_path / (name + ".mp4"))
        w, h = kw.pop("width"), kw.pop("height")
        gold = make_hevc_clip(path, width=w, height=h, **kw)
        idx = hw.index_video(path)
        clips.append((path, idx, gold, w, h, hw.Decoder(path, video_index=idx, num_lanes=6)))
    errors = []

    def work(path, idx, gold, w, h, dec):
        try:
            for rnd in range(3):
                got = dec.retrieve_yuv(list(range(idx.frames())))
                for i in range(len(gold)):
                    if not np.array_equal(got[i], gold[i]):
                        raise AssertionError("%s round %d: I420 frame %d differs" % (path, rnd, i))
                rows = list(range(rnd, idx.frames(), 5))
                rgb = dec.retrieve(rows)
                for k, r in enumerate(rows):
                    if not np.array_equal(rgb[k], oracle.i420_to_rgb(gold[r], w, h)):
                        raise AssertionError("%s round %d: RGB row %d differs" % (path, rnd, r))
        except Exception as e:  # noqa: BLE001 - reported by the test's own thread
            errors.append(e)

    threads = [threading.Thread(target=work, args=c) for c in clips]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[0]


def test_every_closed_gop_with_a_wanted_row_is_a_work_item(hw, oracle, gpu, tmp_path):
    """Rows in byte-adjacent GOPs: the planner merges those GOPs into one interval (one seek for the reference's
    serial loop); the retriever cuts the interval at every sync sample that is an IDR picture again (util/avc_sample.h
    reads the NAL unit types through the hvcC box's length size) and feeds each GOP only up to its last wanted frame."""
    path = str(tmp_path / "g.mp4")
    gold = make_hevc_clip(path, width=176, height=144, frames=96, gop=12, bframes=0, refs=1, ctb=16)
    idx = hw.index_video(path)
    assert list(idx.keyframe_indices()) == list(range(0, 96, 12))
    rows = [1, 13, 26, 40, 41, 95]                     # GOPs 0, 1, 2, 3, 3, 7: 0 .. 3 are adjacent
    dec = hw.Decoder(path, video_index=idx, num_lanes=4)
    got = dec.retrieve_yuv(rows)
    for k, r in enumerate(rows):
        assert np.array_equal(got[k], gold[r]), "row %d differs" % r
    st = dec.stats()
    assert st["intervals"] == 5, st                     # not 2 (the planner's intervals)
    assert st["pictures_reconstructed"] == 2 + 2 + 3 + 6 + 12, st   # each GOP up to its last wanted frame, not to its end
    assert st["frames_returned"] == len(rows)


@pytest.mark.parametrize("name", ["open_gop_bb", "open_gop_tools"])
def test_open_gops_and_decoding_that_enters_at_a_cra_picture(hw, oracle, gpu, tmp_path, name):
    """x265's default GOP structure: CRA pictures with leading pictures that predict across them.  Sequentially every
    frame is the encoder's reconstruction.  A sparse request whose interval starts at a CRA sync sample enters the
    stream there through DecoderAutomata: the leading pictures' references were never decoded, 8.3.3 makes them up as
    surfaces of 128s (HEVCVideoDecoder::missing_surface), and what comes out is what the oracle gives from the same
    entry point - the leading pictures included.  The retriever starts one sync sample earlier when such a leading
    picture is wanted, and returns the sequential decode's frames."""
    kw = dict(dict(HEVC_MATRIX)[name])
    path = str(tmp_path / "o.mp4")
    gold = make_hevc_clip(path, **kw)
    idx = hw.index_video(path)
    n, gop = kw["frames"], kw["gop"]
    dec = hw.Decoder(path, video_index=idx, num_lanes=3)
    got = dec.retrieve_yuv(list(range(n)))
    for i in range(n):
        assert np.array_equal(got[i], gold[i]), "sequential, frame %d" % i
    # (the reference's planner - pinned by its golden values - starts the request's FIRST interval at the file's first
    # sync sample when the GOPs in between are byte-adjacent: rows of the second GOP are decoded from the IDR picture,
    # rows of the third enter at its CRA picture)
    enters = 2 * gop
    assert [iv[0] for iv in hw.slice_into_video_intervals(idx, [enters, enters + 1])] == [(enters, n)]
    want, _ = oracle.decode_file_hevc(path, idx, first=enters, count=n - enters)
    for rows in ([enters, enters + 1, enters + 2, enters + 7], [enters + 1], [enters + 3, n - 1], [gop, gop + 1, enters, enters + 1]):
        # the retriever: a wanted leading picture of the CRA picture an interval starts at moves the start one sync sample
        # back - every row is the sequential decode's frame
        got = dec.retrieve_yuv(rows)
        for k, r in enumerate(rows):
            assert np.array_equal(got[k], gold[r]), "rows %s: row %d" % (rows, r)
        # DecoderAutomata (the reference's API): decoding enters where the planner's interval starts
        (start, _end), _ = hw.slice_into_video_intervals(idx, rows)[0]
        serial = dec.retrieve_serial(rows)
        for k, r in enumerate(rows):
            frame = want[r - enters][0] if start == enters else gold[r]
            assert np.array_equal(serial[k], oracle.i420_to_rgb(frame, 128, 96)), "automata path, rows %s: row %d" % (rows, r)
            if start == enters:
                assert np.array_equal(frame, gold[r]) == (r >= enters + kw["bframes"])
    st_rows = [enters + 1]
    dec.retrieve_yuv(st_rows)
    assert dec.stats()["pictures_reconstructed"] > 2      # (not just the CRA picture and its leading picture)


def test_leading_pictures_of_a_closed_gop(hw, oracle, gpu, tmp_path):
    """IDR_W_RADL pictures with RADL pictures: a work item that starts at such a sync sample outputs the leading
    pictures first (negative picture order counts), so its first wanted rows are not its first sample's picture."""
    kw = dict(dict(HEVC_MATRIX)["radl_bb_tmvp"])
    path = str(tmp_path / "r.mp4")
    gold = make_hevc_clip(path, **kw)
    idx = hw.index_video(path)
    n, gop = kw["frames"], kw["gop"]
    dec = hw.Decoder(path, video_index=idx, num_lanes=3)
    got = dec.retrieve_yuv(list(range(n)))
    for i in range(n):
        assert np.array_equal(got[i], gold[i]), "sequential, frame %d" % i
    assert dec.stats()["intervals"] == n // gop                       # every GOP is closed: three work items
    for rows in ([gop, gop + 1, gop + 2], [gop + 1, 2 * gop, 2 * gop + 1, n - 1], [2 * gop + 1]):
        got = dec.retrieve_yuv(rows)
        for k, r in enumerate(rows):
            assert np.array_equal(got[k], gold[r]), "rows %s: row %d" % (rows, r)
        serial = dec.retrieve_serial(rows)
        for k in range(len(rows)):
            assert np.array_equal(serial[k], oracle.i420_to_rgb(got[k], 128, 96)), "automata path, rows %s: row %d" % (rows, rows[k])


def test_a_cra_gop_is_a_work_item_unless_one_of_its_leading_pictures_is_wanted(hw, oracle, gpu, tmp_path):
    """Open GOPs (x265's default): a GOP that starts with a CRA picture is a work item of its own - decoding enters at
    the CRA picture - when no wanted row is one of the CRA picture's leading pictures; a wanted leading picture keeps its
    GOP in the item of the GOP before, where it has its references.  Every frame is the sequential decode's either way.
    (Where the PLANNER starts an interval at a CRA picture - rows [5, 25]: the GOP between holds nothing - decoding enters
    there as the reference's would, leading picture or not: test_open_gops_and_decoding_that_enters_at_a_cra_picture.)"""
    kw = dict(dict(HEVC_MATRIX)["open_gop_bb"])
    path = str(tmp_path / "o.mp4")
    gold = make_hevc_clip(path, **kw)
    idx = hw.index_video(path)
    dec = hw.Decoder(path, video_index=idx, num_lanes=3)
    for rows, items in (([5, 27, 35], 2),        # [0, 12) and [24, 36): the second GOP holds nothing wanted
                        ([5, 14, 25], 2),        # 25 leads the CRA picture of sample 24: [0, 12) and [12, 36), entered at the CRA picture of sample 12
                        ([13, 26], 2),           # 13 leads the CRA picture of sample 12: [0, 24) and [24, 36)
                        ([5, 14, 26, 35], 3),    # nothing leading: three items
                        ([12, 13, 24, 25], 1)):  # leading pictures everywhere: one item
        got = dec.retrieve_yuv(rows)
        for k, r in enumerate(rows):
            assert np.array_equal(got[k], gold[r]), "rows %s: row %d" % (rows, r)
        assert dec.stats()["intervals"] == items, (rows, dec.stats())


def test_fetching_and_feeding_from_two_threads_many_times(hw, oracle, gpu, tmp_path):
    """DecoderAutomata feeds on one thread and fetches on another; frames fetched into pageable memory wait in a ring of
    page-locked bounce buffers that BOTH threads deliver from.  An entry armed by the fetching thread before its job had
    reached the engine looked "landed" to the feeding thread, which copied the entry's previous frame into the new
    destination: 1 - 3 % of such requests returned an earlier frame of the same request in a later row.  300 requests
    through the reference's API, every frame against the retriever's."""
    path = str(tmp_path / "t.mp4")
    make_hevc_clip(path, frames=36, gop=12, bframes=2, refs=2, open_gop=1, yuv=False)
    idx = hw.index_video(path)
    dec = hw.Decoder(path, video_index=idx, num_lanes=2)
    for rows in ([3, 11, 12, 30], [26, 27, 31, 35], [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]):   # (no leading picture of a CRA picture an interval starts at: the two paths enter at different samples for those)
        want = dec.retrieve(rows)
        for it in range(100):
            got = dec.retrieve_serial(rows)
            for k, r in enumerate(rows):
                assert np.array_equal(got[k], want[k]), "request %d, rows %s: row %d is not the retriever's frame" % (it, rows, r)


def test_a_thread_drives_several_hevc_decoders(hw, oracle, gpu, tmp_path):
    """Long queues are served by a thread per CPU that feeds several lanes in turn, a sample of each at a time, and
    passes over a decoder whose next picture would start by waiting for the GPU (BatchingHints::feed_would_wait:
    the staging set it needs is still being read).  Three and six lanes per thread, every frame of a clip of short
    GOPs (40 work items), three rounds each: bit-exact, and the same through the sparse path."""
    import os
    path = str(tmp_path / "t.mp4")
    gold = make_hevc_clip(path, width=320, height=192, frames=120, gop=3, bframes=0, refs=2, ctb=32, sao=1)
    idx = hw.index_video(path)
    for lanes, share in ((6, "3"), (12, "6"), (8, "1")):
        os.environ["HWANG_LONG_QUEUE_FROM"] = "1"
        os.environ["HWANG_LANES_PER_THREAD"] = share
        try:
            dec = hw.Decoder(path, video_index=idx, num_lanes=lanes)
            for rnd in range(3):
                got = dec.retrieve_yuv(list(range(idx.frames())))
                for i in range(len(gold)):
                    assert np.array_equal(got[i], gold[i]), "lanes %d, %s a thread, round %d: frame %d differs" % (lanes, share, rnd, i)
                rows = list(range(rnd, idx.frames(), 7))
                rgb = dec.retrieve(rows)
                for k, r in enumerate(rows):
                    assert np.array_equal(rgb[k], oracle.i420_to_rgb(gold[r], 320, 192)), "lanes %d round %d: RGB row %d differs" % (lanes, rnd, r)
            assert dec.stats()["lane_threads"] == (lanes + int(share) - 1) // int(share)
            del dec
        finally:
            os.environ.pop("HWANG_LONG_QUEUE_FROM", None)
            os.environ.pop("HWANG_LANES_PER_THREAD", None)


@pytest.mark.parametrize("name", ["ibp_amp_wp_ctb32", "everything", "i_slices_cip_nolf"])
def test_corrupt_streams_never_hang_the_gpu(hw, oracle, gpu, tmp_path, name):
    """Bit flips in HEVC samples reach the kernels as absurd vectors, levels and modes (or are refused by the parser -
    tests/test_hevc_fuzz.py checks on the CPU that no record points outside the picture).  Either is fine; a hung or
    faulted GPU is not: every request returns, and the engine is still healthy afterwards - the pristine clip decodes
    bit-exactly through the same process."""
    kw = dict(dict(HEVC_MATRIX)[name])
    w, h = kw.pop("width", 128), kw.pop("height", 96)
    path = str(tmp_path / "fz.mp4")
    gold = make_hevc_clip(path, width=w, height=h, **kw)
    idx = hw.index_video(path)
    nf = idx.frames()
    data0 = np.fromfile(path, dtype=np.uint8)
    offs, sizes = idx.sample_offsets(), idx.sample_sizes()
    rng = np.random.default_rng(23)
    errors = decoded = 0
    trials = 60
    for trial in range(trials):
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
        os.unlink(bad)
    assert errors + decoded == trials and errors > 0
    got = hw.Decoder(path, video_index=idx, num_lanes=2).retrieve_yuv(list(range(nf)))
    for i in range(nf):
        assert np.array_equal(got[i], gold[i])


def test_frames_into_hbm_and_into_page_locked_memory(hw, oracle, gpu, tmp_path):
    """The three kinds of destination get_frame may be handed (hevc_video_decoder.cpp: take): HBM - the conversion kernel
    writes it -, page-locked host memory - the D2H copy writes it, also at an offset into the allocation -, pageable
    memory - through a bounce buffer.  The same bytes in all three, RGB and I420."""
    torch = gpu
    path = str(tmp_path / "dst.mp4")
    w, h = 176, 144
    gold = make_hevc_clip(path, width=w, height=h, frames=24, gop=12, bframes=2, refs=2, sao=1, ctb=32)
    idx = hw.index_video(path)
    rows = [1, 2, 5, 11, 12, 17, 23]
    frame = w * h * 3
    dec = hw.Decoder(path, video_index=idx, num_lanes=2)
    want = np.stack([oracle.i420_to_rgb(gold[r], w, h) for r in rows])
    dev = torch.zeros((len(rows), h, w, 3), dtype=torch.uint8, device="cuda")
    dec.retrieve_into(rows, dev.data_ptr(), dev.numel())
    torch.cuda.synchronize()
    assert np.array_equal(dev.cpu().numpy(), want)
    pinned = torch.zeros((len(rows) + 1) * frame, dtype=torch.uint8, pin_memory=True)
    dec.retrieve_into(rows, pinned.data_ptr() + frame, len(rows) * frame)
    assert not pinned.numpy()[:frame].any()
    assert np.array_equal(pinned.numpy()[frame:].reshape(want.shape), want)
    assert np.array_equal(np.stack(dec.retrieve(rows)), want)
    # I420 into HBM (three strided device-to-device copies) and into page-locked memory
    yframe = w * h * 3 // 2
    dev_yuv = torch.zeros(len(rows) * yframe, dtype=torch.uint8, device="cuda")
    dec.retrieve_into(rows, dev_yuv.data_ptr(), dev_yuv.numel(), yuv=True)
    torch.cuda.synchronize()
    pin_yuv = torch.zeros(len(rows) * yframe, dtype=torch.uint8, pin_memory=True)
    dec.retrieve_into(rows, pin_yuv.data_ptr(), pin_yuv.numel(), yuv=True)
    for k, r in enumerate(rows):
        assert np.array_equal(dev_yuv.cpu().numpy()[k * yframe:(k + 1) * yframe], gold[r]), "I420 in HBM, row %d" % r
        assert np.array_equal(pin_yuv.numpy()[k * yframe:(k + 1) * yframe], gold[r]), "I420 page-locked, row %d" % r

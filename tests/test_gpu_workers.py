"""-m gpu: `num_devices` > 1 on the drop-in path.  The reference hands num_devices to its backend
(video_decoder_factory.cpp:90-91), whose software flavour uses it as libavcodec's thread count
(software_video_decoder.cpp:148).  The AMD backend does the same: ONE decoder behind ONE DecoderAutomata entropy
decodes consecutive pictures on that many threads, hands them to the GPU in decoding order and rotates the
intervals through several executor lanes.  Output must not depend on it: every case against the oracle-equal
reconstruction and against the single-threaded decoder, through `hwang.Decoder` (pybind11, the reference's Python
surface) and through the C ABI."""
import ctypes as C
import hashlib
import threading

import numpy as np
import pytest

from tests.corpus_util import make_clip

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; there is no fallback"
    return torch


@pytest.fixture(scope="module")
def ref_api(built):
    import hwang
    return hwang


def md5s(frames):
    return [hashlib.md5(np.ascontiguousarray(f).tobytes()).hexdigest() for f in frames]


CASES = [
    ("ibp", dict(frames=72, gop=12, bframes=2, slices=2, refs=2)),
    ("pyramid_temporal", dict(frames=68, gop=17, bframes=3, refs=4, bpyramid=1, temporal_direct=1, implicit_wp=1)),
    ("p_only_cavlc_mmco", dict(frames=60, gop=15, bframes=0, refs=4, cabac=0, mmco=1, rplm=1)),
    ("mixed_everything", dict(frames=52, gop=13, bframes=2, refs=3, slice_mbs=37, mixed_slices=1, multi_pps=2, qp_swing=1,
                              wild_mv=1, extra_nals=1)),
    ("all_intra", dict(frames=12, gop=1, bframes=0, no_stss=1)),
]


@pytest.mark.parametrize("name,kw", CASES, ids=[c[0] for c in CASES])
def test_reference_api_with_entropy_workers(ref_api, oracle, gpu, tmp_path, name, kw):
    path = str(tmp_path / (name + ".mp4"))
    gold = make_clip(path, **kw)
    n = len(gold)
    rng = np.random.default_rng(5)
    sparse = sorted(rng.choice(n, max(3, n // 8), replace=False).tolist())
    for rows in (list(range(n)), sparse):
        want = [oracle.i420_to_rgb(gold[r], 320, 240) for r in rows]
        for workers in (1, 4, 16):
            for group_bytes in (64 << 20, 0):   # the intervals in one initialize, and the reference's one-by-one loop
                dec = ref_api.Decoder(path, device_type=ref_api.DeviceType.GPU, num_devices=workers,
                                      interval_group_bytes=group_bytes)
                got = dec.retrieve(rows)
                assert len(got) == len(rows)
                for k, r in enumerate(rows):
                    assert np.array_equal(got[k], want[k]), "%s: row %d, %d workers, groups of %d bytes" % (name, r, workers,
                                                                                                         group_bytes)
                del dec


def test_workers_at_1080p_many_intervals(ref_api, oracle, gpu, tmp_path):
    """Bench-shaped: 1080p, GOP 60 with two non-reference B pictures between P pictures, a sparse sample over many
    GOPs.  16 workers == 1 worker == the multi-lane retriever, frame for frame; more surfaces in use than one interval
    needs (the intervals overlap on the GPU)."""
    import hwang_amd
    path = str(tmp_path / "hd.mp4")
    make_clip(path, yuv=False, width=1920, height=1080, frames=120, repeat=6, gop=60, bframes=2, refs=2, threads=2)
    idx = hwang_amd.index_video(path)
    rows = sorted(np.random.default_rng(11).choice(idx.frames(), 40, replace=False).tolist())
    lanes = md5s(hwang_amd.Decoder(path, video_index=idx, num_lanes=8).retrieve(rows))
    one = md5s(ref_api.Decoder(path, device_type=ref_api.DeviceType.GPU, num_devices=1).retrieve(rows))
    many = md5s(ref_api.Decoder(path, device_type=ref_api.DeviceType.GPU, num_devices=16).retrieve(rows))
    assert one == lanes
    assert many == lanes
    # again on the same object (state carried across calls), other rows
    dec = ref_api.Decoder(path, device_type=ref_api.DeviceType.GPU, num_devices=16)
    for seed in (1, 2, 3):
        rows = sorted(np.random.default_rng(seed).choice(idx.frames(), 25, replace=False).tolist())
        assert md5s(dec.retrieve(rows)) == md5s(hwang_amd.Decoder(path, video_index=idx, num_lanes=4).retrieve(rows))


def test_references_written_on_another_lane_are_waited_for(ref_api, oracle, gpu, tmp_path, monkeypatch):
    """A decoder with workers moves to the next executor lane at every sync sample; the executor orders work only within
    a lane.  With open GOPs a picture behind a sync sample may refer to the GOP before it - to a surface another lane
    wrote.  The corpus encoder only writes closed GOPs, so the test hook rotates the lanes at EVERY sample: every P and
    B picture then reads references that a different lane reconstructs, and must still come out bit-exact."""
    monkeypatch.setenv("HWANG_TEST_ROTATE_LANES", "1")
    for name, kw in (("ibp", dict(frames=72, gop=24, bframes=2, refs=3)),
                     ("pyramid", dict(frames=68, gop=17, bframes=3, refs=4, bpyramid=1, temporal_direct=1)),
                     ("p_long_refs", dict(frames=60, gop=30, bframes=0, refs=4, mmco=1, rplm=1))):
        path = str(tmp_path / (name + "_rot.mp4"))
        gold = make_clip(path, **kw)
        rows = list(range(len(gold)))
        want = [oracle.i420_to_rgb(gold[r], 320, 240) for r in rows]
        for rounds in range(3):
            dec = ref_api.Decoder(path, device_type=ref_api.DeviceType.GPU, num_devices=8)
            got = dec.retrieve(rows)
            for k, r in enumerate(rows):
                assert np.array_equal(got[k], want[k]), "%s: row %d (round %d)" % (name, r, rounds)
            del dec


FUZZ_TRIALS = int(__import__("os").environ.get("HWANG_FUZZ_TRIALS", "120")) // 2
FUZZ_CLIPS = [
    ("plain", dict(frames=24, gop=6, bframes=2, refs=3, slices=2, seed=5)),
    ("pyramid_mixed", dict(frames=26, gop=13, bframes=3, refs=4, bpyramid=1, mixed_slices=1, slice_mbs=23, multi_pps=2,
                           temporal_direct=1, implicit_wp=1, seed=6)),
]


@pytest.mark.parametrize("name,kw", FUZZ_CLIPS, ids=[c[0] for c in FUZZ_CLIPS])
def test_corrupt_streams_through_workers_never_hang(ref_api, oracle, gpu, tmp_path, name, kw):
    """Bit flips in slice data on the drop-in path with entropy workers: a picture that fails takes its interval with
    it (frames already fetched are cancelled, copier threads let go), absurd vectors / levels reach the kernels.  Every
    call returns - with frames or with an error - on ONE decoder object that lives through all of them, and the
    pristine clip decodes bit-exactly on that same object afterwards."""
    import hwang_amd
    path = str(tmp_path / "fzw.mp4")
    gold = make_clip(path, width=176, height=144, **kw)
    idx = hwang_amd.index_video(path)
    nf = idx.frames()
    data0 = np.fromfile(path, dtype=np.uint8)
    offs, sizes = idx.sample_offsets(), idx.sample_sizes()
    rng = np.random.default_rng(13)
    errors = decoded = 0
    for trial in range(FUZZ_TRIALS):
        data = data0.copy()
        for _ in range(1 + trial % 5):
            s = int(rng.integers(0, len(offs)))
            pos = offs[s] + int(rng.integers(6, sizes[s]))
            data[pos] ^= 1 << int(rng.integers(0, 8))
        bad = str(tmp_path / ("fzw_%d.mp4" % trial))
        data.tofile(bad)
        ridx = ref_api.VideoIndex.deserialize(idx.serialize())
        dec = ref_api.Decoder(bad, video_index=ridx, device_type=ref_api.DeviceType.GPU, num_devices=4,
                              interval_group_bytes=(64 << 20) if trial % 2 else 0)
        trace = __import__("os").environ.get("HWANG_FUZZ_TRACE")
        if trace:
            print("trial %d: retrieve" % trial, flush=True)
        try:
            out = dec.retrieve(list(range(nf)))
            assert len(out) == nf
            decoded += 1
        except RuntimeError as e:
            errors += 1
            if trace:
                print("trial %d: raised %s; again" % (trial, str(e)[:80]), flush=True)
            # the same object goes on: whatever the failed call left behind is cleared by the next initialize
            try:
                dec.retrieve([0, nf - 1])
            except RuntimeError:
                pass
        if trace:
            print("trial %d: del" % trial, flush=True)
        del dec
        __import__("os").unlink(bad)
    assert errors + decoded == FUZZ_TRIALS
    dec = ref_api.Decoder(path, device_type=ref_api.DeviceType.GPU, num_devices=4)
    got = dec.retrieve(list(range(nf)))
    for i in range(nf):
        assert np.array_equal(got[i], oracle.i420_to_rgb(gold[i], 176, 144)), "frame %d after %d errors" % (i, errors)


def test_gpu_fault_with_workers_fails_the_call_and_returns(gpu, tmp_path):
    """The executor's fault word (a row kernel whose bounded poll expired, HWANG_INJECT_FAULT) on the drop-in path: frames
    have been fetched before their pictures ran, copier threads are waiting for conversions that will never be good -
    retrieve() must raise what the executor says and return, not hang and not hand out frames.  (The error is sticky
    for the process's executor, hence the child process.)"""
    import os
    import subprocess
    import sys
    from tests.util import ROOT
    path = str(tmp_path / "fault_workers.mp4")
    make_clip(path, width=320, height=240, frames=24, gop=6, bframes=2, yuv=False)
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import hwang\n"
        "dec = hwang.Decoder(%r, device_type=hwang.DeviceType.GPU, num_devices=4)\n"
        "try:\n"
        "    frames = dec.retrieve(list(range(24)))\n"
        "except RuntimeError as e:\n"
        "    print('RAISED', e)\n"
        "else:\n"
        "    print('NO ERROR', len(frames))\n" % (ROOT, path))
    env = dict(os.environ, HWANG_INJECT_FAULT="1")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert "RAISED" in r.stdout and "hand-off timed out" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "NO ERROR 24" in r.stdout, r.stdout + r.stderr


def test_workers_report_a_corrupt_picture(ref_api, gpu, tmp_path):
    """A picture whose slice data fails with the workers: the error surfaces from retrieve (latched at a later feed or
    at the flush), nothing hangs, the decoder object can be used again."""
    path = str(tmp_path / "bad.mp4")
    make_clip(path, yuv=False, frames=36, gop=12, bframes=0, refs=2, cabac=0)
    import hwang_amd
    idx = hwang_amd.index_video(path)
    raw = bytearray(open(path, "rb").read())
    o, n = idx.sample_offsets()[17], idx.sample_sizes()[17]
    for k in range(o + 12, o + n):   # garbage behind the slice header of one P picture
        raw[k] = (raw[k] * 7 + 13) & 0xff
    bad = str(tmp_path / "bad2.mp4")
    open(bad, "wb").write(bytes(raw))
    done = {}

    def run():
        dec = ref_api.Decoder(bad, device_type=ref_api.DeviceType.GPU, num_devices=8)
        try:
            dec.retrieve(list(range(12, 24)))
            done["error"] = None
        except RuntimeError as e:
            done["error"] = str(e)
        try:
            done["good"] = len(dec.retrieve([0, 5, 30]))   # intervals that do not touch the damaged picture
        except RuntimeError as e:
            done["good"] = "second retrieve failed: " + str(e)

    t = threading.Thread(target=run)
    t.start()
    t.join(120)
    assert not t.is_alive(), "retrieve hung on a corrupt picture"
    assert done["error"], "the damaged interval decoded without an error"
    assert done["good"] == 3


def test_c_abi_decoder_with_workers(hw, oracle, gpu, tmp_path):
    """hwang_decoder_create(..., num_devices = 8, ...): the seam's own entry points, fed from one thread, fetched from
    another with the automata's back-pressure."""
    from hwang_amd._capi import lib
    path = str(tmp_path / "seam.mp4")
    gold = make_clip(path, frames=48, gop=12, bframes=2, refs=2)
    idx = hw.index_video(path)
    raw = open(path, "rb").read()
    offs, sizes = idx.sample_offsets(), idx.sample_sizes()
    kfs = set(idx.keyframe_indices())
    dec = lib.hwang_decoder_create(1, 0, 8, 3)   # GPU 0, num_devices 8, HWANG_DECODER_AMD
    assert dec
    ex = bytes(idx.metadata_bytes())
    assert lib.hwang_decoder_configure(dec, 320, 240, b"avc1", ex, len(ex)) == 0
    fs = 320 * 240 * 3
    out = np.zeros((48, fs), dtype=np.uint8)
    fed = threading.Event()
    errors = []

    def feeder():
        for first in range(0, 48, 12):
            for s in range(first, first + 12):
                while lib.hwang_decoder_decoded_frames_buffered(dec) > 8:
                    pass
                if lib.hwang_decoder_feed(dec, raw[offs[s]:offs[s] + sizes[s]], sizes[s], 1 if s in kfs else 0):
                    errors.append(lib.hwang_decoder_last_error(dec))
                    return
            lib.hwang_decoder_feed(dec, None, 0, 0)
            if lib.hwang_decoder_flush(dec):
                errors.append(lib.hwang_decoder_last_error(dec))
                return
        fed.set()

    t = threading.Thread(target=feeder)
    t.start()
    got = 0
    import time
    deadline = time.time() + 120
    while got < 48 and time.time() < deadline and not errors:
        if lib.hwang_decoder_decoded_frames_buffered(dec) > 0:
            assert lib.hwang_decoder_get_frame(dec, out[got].ctypes.data_as(C.c_void_p), fs) == 0
            got += 1
    t.join(30)
    assert not errors, errors
    assert got == 48 and fed.is_set()
    assert lib.hwang_decoder_wait_until_frames_copied(dec) == 0
    for r in range(48):
        assert np.array_equal(out[r].reshape(240, 320, 3), oracle.i420_to_rgb(gold[r], 320, 240)), "frame %d" % r
    lib.hwang_decoder_destroy(dec)

"""-m gpu: the drop-in seam itself, behaviourally.

The eight C-ABI entry points INTEGRATION.md's adapter forwards to
(hwang_decoder_create / configure / feed / flush / get_frame / discard_frame /
decoded_frames_buffered / wait_until_frames_copied) driven the way the
REFERENCE's DecoderAutomata drives a backend (decoder_automata.cpp:120-404):
a feeder thread (feed per sample with back-pressure at > 8 buffered frames,
then (NULL, 0) + flush) and a caller thread (poll decoded_frames_buffered,
get_frame the wanted frames, discard the rest), on the MI355X backend."""
import ctypes as C
import threading
import time

import numpy as np
import pytest

from tests.corpus_util import make_clip

pytestmark = pytest.mark.gpu

GPU, AMD = 1, 3


class Seam:
    def __init__(self, lib, idx):
        self.lib = lib
        self.h = lib.hwang_decoder_create(GPU, 0, 1, AMD)
        assert self.h, "hwang_decoder_create failed"
        self.w, self.hh = idx.frame_width(), idx.frame_height()
        self.configure(idx)

    def configure(self, idx):
        extra = bytes(idx.metadata_bytes())
        rc = self.lib.hwang_decoder_configure(self.h, self.w, self.hh, idx.format().encode(), extra, len(extra))
        assert rc == 0, self.err()

    def err(self):
        return self.lib.hwang_decoder_last_error(self.h).decode()

    def close(self):
        self.lib.hwang_decoder_destroy(self.h)
        self.h = None

    def run_interval(self, data, offs, sizes, keyframes, first, last, wanted):
        """Reference protocol over samples [first, last); returns {display ordinal: RGB array}."""
        lib, h = self.lib, self.h
        errors = []
        fed_all = threading.Event()

        def feeder():
            try:
                for s in range(first, last):
                    while lib.hwang_decoder_decoded_frames_buffered(h) > 8:   # MAX_BUFFERED_FRAMES back-pressure
                        time.sleep(0.0005)
                    buf = data[offs[s]:offs[s] + sizes[s]]
                    if lib.hwang_decoder_feed(h, buf.ctypes.data, len(buf), 1 if s in keyframes else 0):
                        errors.append("feed: " + self.err())
                        return
                if lib.hwang_decoder_feed(h, None, 0, 0) or lib.hwang_decoder_flush(h):   # end of interval
                    errors.append("flush: " + self.err())
            finally:
                fed_all.set()

        t = threading.Thread(target=feeder)
        t.start()
        got = {}
        n = last - first
        frame_bytes = self.w * self.hh * 3
        ordinal = 0
        deadline = time.time() + 120
        while ordinal < n and not errors and time.time() < deadline:
            if lib.hwang_decoder_decoded_frames_buffered(h) == 0:
                time.sleep(0.0005)
                continue
            if ordinal in wanted:
                out = np.empty(frame_bytes, dtype=np.uint8)
                if lib.hwang_decoder_get_frame(h, out.ctypes.data, out.nbytes):
                    errors.append("get_frame: " + self.err())
                    break
                got[ordinal] = out
            elif lib.hwang_decoder_discard_frame(h):
                errors.append("discard: " + self.err())
                break
            ordinal += 1
        t.join()
        assert not errors, errors
        assert ordinal == n, "one output per fed sample: got %d of %d" % (ordinal, n)
        assert lib.hwang_decoder_wait_until_frames_copied(h) == 0, self.err()
        assert lib.hwang_decoder_decoded_frames_buffered(h) == 0
        return {k: v.reshape(self.hh, self.w, 3) for k, v in got.items()}


def test_reference_protocol_from_two_threads(hw, oracle, tmp_path):
    from hwang_amd._capi import lib
    path = str(tmp_path / "seam.mp4")
    gold = make_clip(path, frames=48, gop=16, bframes=2, slices=2, refs=3)
    idx = hw.index_video(path)
    data = np.fromfile(path, dtype=np.uint8)
    offs, sizes = idx.sample_offsets(), idx.sample_sizes()
    kfs = set(idx.keyframe_indices())
    seam = Seam(lib, idx)
    # interval 1: GOPs 0-1 (two keyframes inside one interval), a sparse set of wanted ordinals
    wanted = {0, 1, 7, 15, 16, 17, 31}
    frames = seam.run_interval(data, offs, sizes, kfs, 0, 32, wanted)
    assert set(frames) == wanted
    for k, f in frames.items():
        assert np.array_equal(f, oracle.i420_to_rgb(gold[k], 320, 240)), "display ordinal %d" % k
    # interval 2 on the same object without configure: flush() left it ready for a keyframe-started interval
    frames = seam.run_interval(data, offs, sizes, kfs, 32, 48, {0, 15})
    assert np.array_equal(frames[0], oracle.i420_to_rgb(gold[32], 320, 240))
    assert np.array_equal(frames[15], oracle.i420_to_rgb(gold[47], 320, 240))
    # repeated configure on a live object, then every frame of GOP 1
    seam.configure(idx)
    frames = seam.run_interval(data, offs, sizes, kfs, 16, 32, set(range(16)))
    for k in range(16):
        assert np.array_equal(frames[k], oracle.i420_to_rgb(gold[16 + k], 320, 240))
    seam.close()


def test_get_frame_edge_semantics(hw, oracle, tmp_path):
    """software_video_decoder.cpp:286-290 (nothing buffered: OK, does nothing) and :321-323 (short buffer fails,
    the frame stays queued)."""
    from hwang_amd._capi import lib
    path = str(tmp_path / "edge.mp4")
    gold = make_clip(path, frames=4, gop=4, bframes=0)
    idx = hw.index_video(path)
    data = np.fromfile(path, dtype=np.uint8)
    offs, sizes = idx.sample_offsets(), idx.sample_sizes()
    seam = Seam(lib, idx)
    h = seam.h
    out = np.full(320 * 240 * 3, 7, dtype=np.uint8)
    assert lib.hwang_decoder_decoded_frames_buffered(h) == 0
    assert lib.hwang_decoder_get_frame(h, out.ctypes.data, out.nbytes) == 0 and (out == 7).all()   # empty queue: no-op
    assert lib.hwang_decoder_discard_frame(h) == 0
    for s in range(4):
        buf = data[offs[s]:offs[s] + sizes[s]]
        assert lib.hwang_decoder_feed(h, buf.ctypes.data, len(buf), 1 if s == 0 else 0) == 0, seam.err()
    assert lib.hwang_decoder_feed(h, None, 0, 0) == 0 and lib.hwang_decoder_flush(h) == 0
    assert lib.hwang_decoder_decoded_frames_buffered(h) == 4
    assert lib.hwang_decoder_get_frame(h, out.ctypes.data, out.nbytes - 1) != 0   # short buffer
    assert "small" in seam.err()
    assert lib.hwang_decoder_decoded_frames_buffered(h) == 4                      # nothing was consumed
    for k in range(4):
        assert lib.hwang_decoder_get_frame(h, out.ctypes.data, out.nbytes) == 0, seam.err()
        assert lib.hwang_decoder_wait_until_frames_copied(h) == 0
        assert np.array_equal(out.reshape(240, 320, 3), oracle.i420_to_rgb(gold[k], 320, 240))
    assert lib.hwang_decoder_decoded_frames_buffered(h) == 0
    # wrong container size is refused at configure (the stream is 320x240)
    extra = bytes(idx.metadata_bytes())
    assert lib.hwang_decoder_configure(h, 640, 480, b"avc1", extra, len(extra)) != 0 and "320x240" in seam.err()
    assert lib.hwang_decoder_configure(h, 320, 240, b"hev1", extra, len(extra)) != 0
    seam.close()
    # a CPU handle has no backend: create fails, nothing falls back
    assert not lib.hwang_decoder_create(0, 0, 1, AMD)
    assert not lib.hwang_decoder_create(GPU, 0, 1, 0)   # SOFTWARE is not in this build

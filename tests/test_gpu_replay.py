"""-m gpu: the replay hooks (include/hwang_amd.h: hwang_replay_keep / hwang_replay_run; bench.py's `replay` block) - a
measurement aid that decodes kept pictures in place again.  ADVICE r4: it must not run over surfaces somebody still has
a frame to collect from, it must size a chain launch's unit table instead of truncating it, and it must report a
hand-off that timed out."""
import ctypes as C

import numpy as np
import pytest

from tests.corpus_util import make_clip
from tests.test_gpu_seam import AMD, GPU

pytestmark = pytest.mark.gpu


def replay(lib, width, launches, chain):
    o = (C.c_double * 4)()
    err = C.create_string_buffer(512)
    rc = lib.hwang_replay_run(0, width, launches, chain, o, err, 512)
    return rc, [o[i] for i in range(4)], err.value.decode()


def test_replay_runs_wide_and_chained_and_leaves_the_decode_intact(hw, tmp_path):
    from hwang_amd._capi import lib
    path = str(tmp_path / "r.mp4")
    make_clip(path, width=640, height=368, yuv=False, frames=36, gop=12, bframes=0, refs=1)
    idx = hw.index_video(path)
    rows = list(range(idx.frames()))
    assert lib.hwang_replay_keep(0, 64) == 0
    try:
        dec = hw.Decoder(path, video_index=idx, num_lanes=3)
        first = [np.array(f) for f in dec.retrieve(rows)]
        for chain in (0, 1):
            rc, o, err = replay(lib, 8, 3, chain)
            assert rc == 0, err
            assert o[0] >= 3 and 1 <= o[1] <= 8 and o[2] == 3 and o[3] > 0
        # a width no unit table holds: fewer pictures per launch (or all there are), never a truncated list
        rc, o, err = replay(lib, 512, 2, 1)
        assert rc == 0 and o[1] <= o[0], err
        # the decoders are as good as new afterwards
        again = [np.array(f) for f in dec.retrieve(rows)]
        assert all((a == b).all() for a, b in zip(first, again))
        del dec
    finally:
        lib.hwang_replay_keep(0, 0)


def test_replay_refuses_while_frames_are_uncollected(hw, tmp_path):
    from hwang_amd._capi import lib
    path = str(tmp_path / "u.mp4")
    make_clip(path, width=320, height=192, yuv=False, frames=6, gop=6, bframes=0, refs=1)
    idx = hw.index_video(path)
    data = np.fromfile(path, dtype=np.uint8)
    offs, sizes = idx.sample_offsets(), idx.sample_sizes()
    assert lib.hwang_replay_keep(0, 16) == 0
    h = lib.hwang_decoder_create(GPU, 0, 0, AMD)
    assert h
    try:
        extra = bytes(idx.metadata_bytes())
        assert lib.hwang_decoder_configure(h, idx.frame_width(), idx.frame_height(), idx.format().encode(), extra, len(extra)) == 0
        for s in range(6):
            buf = data[offs[s]:offs[s] + sizes[s]]
            assert lib.hwang_decoder_feed(h, buf.ctypes.data, len(buf), 1 if s == 0 else 0) == 0
        assert lib.hwang_decoder_feed(h, None, 0, 0) == 0 and lib.hwang_decoder_flush(h) == 0
        assert lib.hwang_decoder_decoded_frames_buffered(h) == 6
        import time
        deadline = time.time() + 30
        while True:   # (the last pictures may still be on the GPU: "busy" is the other refusal)
            rc, o, err = replay(lib, 4, 1, 0)
            if "busy" not in err or time.time() > deadline:
                break
            time.sleep(0.01)
        assert rc != 0 and "not fetched or discarded" in err, (rc, err)
        out = np.empty(idx.frame_width() * idx.frame_height() * 3, dtype=np.uint8)
        for _ in range(3):
            assert lib.hwang_decoder_get_frame(h, out.ctypes.data, out.nbytes) == 0
        for _ in range(3):
            assert lib.hwang_decoder_discard_frame(h) == 0
        assert lib.hwang_decoder_wait_until_frames_copied(h) == 0
        rc, o, err = replay(lib, 4, 1, 0)
        assert rc == 0, err
    finally:
        lib.hwang_decoder_destroy(h)
        lib.hwang_replay_keep(0, 0)

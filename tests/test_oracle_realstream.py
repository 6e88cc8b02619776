"""The oracle decoder on a real x264 stream: every slice must end exactly on
its rbsp_stop_one_bit (a CABAC desync check that needs no second decoder and
pins the arithmetic-decoder tables and every context the stream touches), and
the output must match the committed digests."""
import hashlib
import json
import os

import pytest

from tests.util import REALSHORT, ROOT, have_file


@pytest.mark.skipif(not have_file(REALSHORT), reason="realshort.mp4 not in this image")
def test_realshort_desync_free_and_stable(hw, oracle):
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "realshort_md5.json")))
    idx = hw.index_video(REALSHORT)
    frames, c = oracle.decode_file(REALSHORT, idx)
    assert c["slices"] == 36 and c["slices_clean_end"] == 36
    assert c["macroblocks"] == 36 * 20 * 15
    assert c["t8x8_mbs"] > 0 and c["intra_mbs"] > 0 and c["skip_mbs"] > 0
    assert len(frames) == 36
    assert [f[3] for f in frames] == gold["pocs"]
    assert [hashlib.md5(f[0].tobytes()).hexdigest() for f in frames] == gold["md5"]


@pytest.mark.skipif(not have_file(REALSHORT), reason="realshort.mp4 not in this image")
def test_realshort_no_drift(hw, oracle):
    """Frame 29 ends a 29-picture P chain; frame 30 is an IDR of the same
    scene.  A reconstruction error anywhere (IDCT, intra, MC, deblock) drifts
    along the chain and would show as a jump in frame-to-frame PSNR at the
    IDR; a correct decode keeps it in family with its neighbours."""
    import numpy as np
    idx = hw.index_video(REALSHORT)
    frames, _ = oracle.decode_file(REALSHORT, idx)
    ys = [f[0][:320 * 240].astype(np.float64) for f in frames]

    def psnr(a, b):
        return 10 * np.log10(255.0 ** 2 / max(((a - b) ** 2).mean(), 1e-9))
    steps = [psnr(ys[i], ys[i + 1]) for i in range(35)]
    around = steps[25:29] + steps[30:34]
    assert steps[29] > min(around) - 1.5

"""-m gpu: the REFERENCE's own DecoderAutomata (compiled from /root/reference/hwang/decoder_automata.cpp in the build
container: tests/ref_driver.py, tests/fixtures/ref_automata_driver.cpp) driving the MI355X backend through
INTEGRATION.md's adapter - north_star's "plugs in behind hwang's existing DecoderAutomata", run rather than argued.
The binary is built where the reference tree exists and travels with the snapshot; without it the test is skipped
(and says so)."""
import os

import numpy as np
import pytest

from tests import ref_driver
from tests.corpus_util import make_clip

pytestmark = pytest.mark.gpu

HWANG_DECODER_AMD = 3   # include/hwang_amd.h:31


@pytest.mark.skipif(not os.path.exists(ref_driver.DRIVER) and not os.path.exists(ref_driver.REF),
                    reason="tests/_build/ref_automata_driver was not prebuilt (needs /root/reference)")
@pytest.mark.parametrize("chunks", [None, [2, 1, 4, 2]])
def test_reference_automata_over_the_mi355x_backend(hw, oracle, tmp_path, chunks):
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; there is no fallback"
    assert ref_driver.build_driver()
    path = str(tmp_path / "clip.mp4")
    gold = make_clip(path, frames=48, gop=12, bframes=2, slices=2)
    idx = hw.index_video(path)
    rows = [0, 1, 5, 11, 12, 13, 30, 35, 47]
    plan = hw.slice_into_video_intervals(idx, rows)
    ref_driver.write_plan(str(tmp_path / "plan.txt"), idx, plan, chunks or [len(rows)], extradata=bytes(idx.metadata_bytes()))
    r = ref_driver.run_driver(str(tmp_path / "plan.txt"), path, str(tmp_path / "out.bin"), HWANG_DECODER_AMD, 1)
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    assert "frames=%d" % len(rows) in r.stdout
    out = np.fromfile(str(tmp_path / "out.bin"), dtype=np.uint8).reshape(len(rows), 240, 320, 3)
    for k, row in enumerate(rows):
        assert np.array_equal(out[k], oracle.i420_to_rgb(gold[row], 320, 240)), "row %d" % row
    # GatherFramesComparison (decoder_automata_test.cpp:262-342): the same rows out of a sequential decode
    allrows = list(range(idx.frames()))
    ref_driver.write_plan(str(tmp_path / "plan_all.txt"), idx, hw.slice_into_video_intervals(idx, allrows), [len(allrows)],
                          extradata=bytes(idx.metadata_bytes()))
    r = ref_driver.run_driver(str(tmp_path / "plan_all.txt"), path, str(tmp_path / "all.bin"), HWANG_DECODER_AMD, 1)
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    seq = np.fromfile(str(tmp_path / "all.bin"), dtype=np.uint8).reshape(len(allrows), 240, 320, 3)
    for k, row in enumerate(rows):
        assert np.array_equal(seq[row], out[k]), "row %d" % row

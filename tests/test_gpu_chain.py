"""-m gpu: chain launches (k_chain: a lane's dependent pictures - keyframe, P / B pictures predicted from each other -
as ONE launch whose stages follow each other macroblock by macroblock).

The path is the default for a decoder behind ONE DecoderAutomata (the reference's caller); the retriever's lanes use the
wide kernels.  HWANG_CHAIN=1 puts every lane on it and HWANG_CHAIN_MAX_LANES lifts the "only while the executor is
nearly idle" rule, so the WHOLE corpus - every syntax path, every picture shape of the band tests - goes through the
chain instantiations of the inter / intra / deblocking code and is compared with the committed digests of the CPU
oracle's output (tests/golden/corpus_md5.json).  Each configuration runs in a process of its own (the switches are read
when a lane registers)."""
import json
import os
import subprocess
import sys

import pytest

from tests.corpus_util import BAND_SIZES, MATRIX, golden_md5, make_clip
from tests.util import ROOT

pytestmark = pytest.mark.gpu


def run_worker(tmp_path, clips, env=None, timeout=600):
    spec = str(tmp_path / "spec.json")
    with open(spec, "w") as f:
        json.dump({"clips": clips}, f)
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_chain_worker.py"), spec], env=e, capture_output=True, text=True,
                       timeout=timeout)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[7:]), r.stderr


CHAIN_EVERYWHERE = {"HWANG_CHAIN": "1", "HWANG_CHAIN_MAX_LANES": "1000", "HWANG_EXEC_DEBUG": "1"}


def test_whole_corpus_through_chain_launches(tmp_path):
    clips, names = [], []
    for name, kw in MATRIX:
        kw = dict(kw)
        w, h = kw.pop("width", 322 if name == "qp40" else 320), kw.pop("height", 242 if name == "qp40" else 240)
        path = str(tmp_path / (name + ".mp4"))
        make_clip(path, width=w, height=h, yuv=False, **kw)
        clips.append({"path": path, "width": w, "height": h, "rows": None, "lanes": 2})
        names.append(name)
    got, log = run_worker(tmp_path, clips, CHAIN_EVERYWHERE, timeout=900)
    assert "chain launches" in log, "the corpus did not go through k_chain"
    for name, frames in zip(names, got):
        assert frames == golden_md5(name)["frames"], name


def test_band_shapes_through_chain_launches(tmp_path):
    clips, names = [], []
    for name, w, h, kw in BAND_SIZES:
        path = str(tmp_path / (name + ".mp4"))
        make_clip(path, width=w, height=h, yuv=False, **kw)
        clips.append({"path": path, "width": w, "height": h, "rows": None, "lanes": 2})
        names.append(name)
    got, log = run_worker(tmp_path, clips, CHAIN_EVERYWHERE, timeout=900)
    assert "chain launches" in log
    for name, frames in zip(names, got):
        assert frames == golden_md5(name)["frames"], name


def test_long_chains_at_1080p_equal_the_wide_kernels(tmp_path):
    """The reference's per-interval caller on 1080p GOPs of 30 (I + P / B chains of up to ~20 pictures in one launch,
    the pictures of a GOP trailing each other by a few rows): identical, frame for frame, to the same rows through the
    retriever's wide kernels, with and without reference B pictures, several slices, weighted prediction."""
    cases = [dict(frames=90, gop=30, bframes=2, refs=3, qp=30), dict(frames=60, gop=30, bframes=3, bpyramid=1, refs=4, slices=3, qp=32),
             dict(frames=60, gop=20, bframes=0, refs=2, explicit_wp=1, wp_denom=5, qp=28, cabac=0)]
    clips_ref, clips_wide = [], []
    for i, kw in enumerate(cases):
        path = str(tmp_path / ("long%d.mp4" % i))
        make_clip(path, width=1920, height=1080, yuv=False, **kw)
        n = kw["frames"]
        rows = sorted({n - 1, n - 2, kw["gop"] - 1, kw["gop"] + 3, 2 * kw["gop"] - 2} & set(range(n)))
        clips_ref.append({"path": path, "width": 1920, "height": 1080, "rows": rows, "api": "reference"})
        clips_wide.append({"path": path, "width": 1920, "height": 1080, "rows": rows, "lanes": 4, "rgb": True})
    got, log = run_worker(tmp_path, clips_ref, {"HWANG_EXEC_DEBUG": "1"}, timeout=600)
    longest = max([int(l.rsplit(" ", 1)[1]) for l in log.splitlines() if "longest chain" in l] or [0])
    assert longest >= 5, "no chains formed (longest %d): the reference's caller should get them by default" % longest
    want, _ = run_worker(tmp_path, clips_wide, {"HWANG_CHAIN": "0"}, timeout=600)
    assert got == want


def test_serial_decoder_chains(tmp_path):
    """HWANG_ENTROPY_WORKERS=0: the thread that calls feed() decodes alone (rounds 1-3's shape of the drop-in path);
    its pictures still travel as chains, as far as its dozen staging sets and two dozen surfaces reach."""
    path = str(tmp_path / "serial.mp4")
    make_clip(path, width=640, height=368, yuv=False, frames=48, gop=24, bframes=2, refs=3)
    rows = [5, 23, 24, 40, 47]
    ref = [{"path": path, "width": 640, "height": 368, "rows": rows, "api": "reference"}]
    got, log = run_worker(tmp_path, ref, {"HWANG_ENTROPY_WORKERS": "0", "HWANG_EXEC_DEBUG": "1"})
    assert "chain launches" in log
    want, _ = run_worker(tmp_path, [{"path": path, "width": 640, "height": 368, "rows": rows, "lanes": 2, "rgb": True}], {"HWANG_CHAIN": "0"})
    assert got == want


def test_lost_chain_hand_off_is_run_again_through_the_wide_kernels(tmp_path):
    """A chain launch in which one run of inter macroblocks never reports (test hook HWANG_INJECT_FAULT=2): the stages
    that wait for it give up once NOTHING in the launch has moved for their bound - the first expiry makes the others
    give up at their next look - and the executor runs the batch's pictures again through the wide kernels, which
    need no hand-off between pictures (a chain's time-out must not be the end of the executor).  The
    request returns within seconds with the frames the wide kernels give, and says on stderr what happened."""
    import time
    path = str(tmp_path / "fault.mp4")
    make_clip(path, width=640, height=368, yuv=False, frames=12, gop=12, bframes=0, refs=2)
    clips = [{"path": path, "width": 640, "height": 368, "rows": None, "lanes": 1}]
    t0 = time.time()
    got, log = run_worker(tmp_path, clips, {"HWANG_CHAIN": "1", "HWANG_INJECT_FAULT": "2"})
    assert time.time() - t0 < 120
    assert "a chain launch timed out (k_chain: inter macroblocks that a row waits for never landed)" in log, log[-2000:]
    assert "again through the wide kernels" in log
    want, _ = run_worker(tmp_path, clips, {"HWANG_CHAIN": "0"})
    assert got == want
    # and the executor is not poisoned: the next request of the same process decodes (two clips, one worker process)
    got2, log2 = run_worker(tmp_path, clips + clips, {"HWANG_CHAIN": "1", "HWANG_INJECT_FAULT": "2"})
    assert got2 == want + want and log2.count("a chain launch timed out") >= 2

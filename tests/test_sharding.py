"""The N > 1 paths on CPU with the STAMP backend:
* bench.py's launch (one process per GPU): two gloo ranks decode their shares
  of one batched call, no collective on the data path;
* one process driving several devices: a request's GOP items land on
  different devices and come back in request order;
* retrieve_batch over clips of different sizes."""
import os
import socket
import subprocess
import sys

from tests.util import ROOT


def test_two_ranks_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_sharding_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "SHARDING_OK rows=120" in outs[0], outs[0]


def _stamp_decoder(hw, n, gop, w, h, num_lanes):
    import io
    import struct

    import numpy as np
    samples = [struct.pack("<I", i) + b"s" * (5 + i % 7) for i in range(n)]
    offs = np.cumsum([0] + [len(s) for s in samples[:-1]]).tolist()
    index = hw.VideoIndex(timescale=30000, duration=n * 1001, width=w, height=h, format="avc1",
                          sample_offsets=offs, sample_sizes=[len(s) for s in samples],
                          keyframe_indices=list(range(0, n, gop)), metadata=b"")
    return hw.Decoder(io.BytesIO(b"".join(samples)), video_index=index, device_type=hw.DeviceType.CPU,
                      decoder_type=hw.STAMP, num_lanes=num_lanes)


def test_automatic_lane_count(hw):
    """num_lanes=0 lets the retriever choose (3 lanes per usable CPU, 2 for
    pictures beyond 2x1080p, never more than there are work items): the rows
    must come back in request order whatever it chose."""
    import struct
    for (w, h, n, rows) in [(4, 2, 900, [0, 1, 29, 30, 31, 450, 899]),
                            (3840, 2160, 120, [3, 59, 61, 119])]:     # 24.9 MB a frame: keep the request small
        dec = _stamp_decoder(hw, n, 30, w, h, 0)
        frames = dec.retrieve(rows)
        assert len(frames) == len(rows)
        for r, f in zip(rows, frames):
            assert f.shape == (h, w, 3)
            assert struct.unpack("<I", f.tobytes()[:4])[0] == r


def test_lanes_shared_by_threads(hw, monkeypatch):
    """A retriever thread drives several lanes in turn, a sample of each at a time (HWANG_LANES_PER_THREAD; default: a thread per usable CPU when there are eight work items per lane or more, else a thread per lane):
    whatever the split - a thread per lane as in rounds 1-4, two, three, more lanes than there are - every row comes
    back, in request order, with the sample it names."""
    import struct
    rows = sorted(set(list(range(0, 600, 7)) + [1, 2, 3, 299, 300, 301, 599]))
    for share in ("1", "2", "3", "8"):
        monkeypatch.setenv("HWANG_LANES_PER_THREAD", share)
        for lanes in (1, 2, 5, 12):
            dec = _stamp_decoder(hw, 600, 20, 4, 2, lanes)
            frames = dec.retrieve(rows)
            assert len(frames) == len(rows)
            for r, f in zip(rows, frames):
                assert struct.unpack("<I", f.tobytes()[:4])[0] == r, (share, lanes, r)
            st = dec.stats()
            assert st["frames_returned"] == len(rows)
            del dec


def _stamp_clip(hw, n, gop, w, h, tag=0):
    import struct

    import numpy as np
    samples = [struct.pack("<I", (tag << 24) | i) + b"s" * (5 + i % 7) for i in range(n)]
    offs = np.cumsum([0] + [len(s) for s in samples[:-1]]).tolist()
    index = hw.VideoIndex(timescale=30000, duration=n * 1001, width=w, height=h, format="avc1",
                          sample_offsets=offs, sample_sizes=[len(s) for s in samples],
                          keyframe_indices=list(range(0, n, gop)), metadata=bytes([tag]))
    return index, b"".join(samples)


def test_one_request_fans_out_over_devices(hw):
    """one request's intervals must land on different
    devices and come back in request order."""
    import io
    import struct

    import numpy as np
    index, blob = _stamp_clip(hw, 3000, 30, 4, 2)
    rows = sorted(np.random.default_rng(5).choice(3000, 90, replace=False).tolist())
    for devices, lanes in (([0, 1], 4), ([0, 1, 2, 3], 0), ([0, 1, 2], 3)):
        dec = hw.Decoder(io.BytesIO(blob), video_index=index, device_type=hw.DeviceType.CPU,
                         decoder_type=hw.STAMP, num_lanes=lanes, device_ids=devices)
        frames = dec.retrieve(rows)
        assert [struct.unpack("<I", f.tobytes()[:4])[0] for f in frames] == rows
        ds = dec.device_stats()
        assert len(ds) == len(devices)
        assert all(d["lanes"] >= 1 and d["items"] >= 1 and d["frames_returned"] >= 1 for d in ds), ds
        assert sum(d["frames_returned"] for d in ds) == len(rows)
        assert sum(d["items"] for d in ds) == dec.stats()["intervals"]


def test_retrieve_batch_pools_clips(hw):
    """BASELINE C3's shape: many clips, a few rows of each, one call.  Clips of
    different sizes share lanes (a lane re-configures when the clip changes)."""
    import io
    import struct

    import numpy as np
    from hwang_amd.decoder import BatchRetriever
    rng = np.random.default_rng(11)
    clips = [_stamp_clip(hw, 600 + 30 * k, 30, 4 + 2 * (k % 3), 2 + (k % 2), tag=k + 1) for k in range(12)]
    reqs, want = [], []
    for k, (index, blob) in enumerate(clips):
        rows = sorted(rng.choice(index.frames(), 6, replace=False).tolist()) if k != 5 else []   # one empty request
        reqs.append((index, io.BytesIO(blob), rows))
        want.append([((k + 1) << 24) | r for r in rows])
    br = BatchRetriever(device_ids=[0, 1], device_type=hw.DeviceType.CPU, decoder_type=hw.STAMP,
                        num_lanes=5)
    got = br.retrieve(reqs)
    for k in range(len(clips)):
        assert [struct.unpack("<I", f.tobytes()[:4])[0] for f in got[k]] == want[k], k
        for f in got[k]:
            assert f.shape == (clips[k][0].frame_height(), clips[k][0].frame_width(), 3)
    ds = br.device_stats()
    assert all(d["items"] > 0 for d in ds) and sum(d["frames_returned"] for d in ds) == sum(len(w) for w in want)
    # shares of a sharded job are disjoint and complete
    seen = [np.zeros(len(w), dtype=np.int32) for w in want]
    for rank in range(3):
        brs = BatchRetriever(device_ids=[0], device_type=hw.DeviceType.CPU, decoder_type=hw.STAMP,
                             num_lanes=2, shard=(rank, 3))
        part = brs.retrieve(reqs)
        for k in range(len(clips)):
            for i, f in enumerate(part[k]):
                v = struct.unpack("<I", f.tobytes()[:4])[0]
                if v:   # zero-initialised output: rows of other shares stay untouched
                    assert v == want[k][i]
                    seen[k][i] += 1
    assert all((s == 1).all() for s in seen)

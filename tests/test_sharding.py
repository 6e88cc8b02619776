"""The N > 1 path of bench.py (one process per GPU, interval sharding, no
collective on the data path) on CPU: two gloo ranks, STAMP backend."""
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
                      decoder_type=hw.VideoDecoderType.STAMP, num_lanes=num_lanes)


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

"""One rank of tests/test_sharding.py: the multi-GPU structure of bench.py on
CPU (gloo), with the STAMP fixture backend standing in for the GPU decoder.
Every rank retrieves its own 1 % sample of the same clip (no data exchange);
ranks only meet in the barrier and in the MAX / SUM reductions."""
import io
import os
import struct
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hwang_amd as hw  # noqa: E402

SEED = 20240601


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n, gop, w, h = 6000, 30, 4, 2
    samples = [struct.pack("<I", i) + b"s" * (5 + i % 7) for i in range(n)]
    offs = np.cumsum([0] + [len(s) for s in samples[:-1]]).tolist()
    index = hw.VideoIndex(timescale=30000, duration=n * 1001, width=w, height=h, format="avc1",
                          sample_offsets=offs, sample_sizes=[len(s) for s in samples],
                          keyframe_indices=list(range(0, n, gop)), metadata=b"")
    blob = io.BytesIO(b"".join(samples))
    rows = sorted(np.random.default_rng(SEED + rank).choice(n, round(0.01 * n), replace=False).tolist())
    dec = hw.Decoder(blob, video_index=index, device_type=hw.DeviceType.CPU,
                     decoder_type=hw.VideoDecoderType.STAMP, num_lanes=3)
    dist.barrier()
    t0 = time.perf_counter()
    frames = dec.retrieve(rows)
    dist.barrier()
    elapsed = time.perf_counter() - t0
    # STAMP: a frame carries the first four bytes of the sample it came from
    for r, f in zip(rows, frames):
        assert f.shape == (h, w, 3)
        assert struct.unpack("<I", f.tobytes()[:4])[0] == r, (rank, r)
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    cnt = torch.tensor([len(rows), dec.stats()["samples_fed"]], dtype=torch.float64)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    other = [None] * world
    dist.all_gather_object(other, rows)
    if rank == 0:
        assert int(cnt[0]) == world * round(0.01 * n)
        assert other[0] != other[1], "ranks must sample independently"
        assert float(t) >= elapsed - 1e-9
        print("SHARDING_OK rows=%d fed=%d" % (int(cnt[0]), int(cnt[1])))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""One rank of tests/test_sharding.py: the multi-GPU structure of bench.py on
CPU (gloo), with the STAMP fixture backend standing in for the GPU decoder.

The job is ONE batched call of `world` requests over the same clip (request k
= the 1 % sample with seed SEED + k, so per-GPU work stays fixed as the world
grows).  Every rank plans the whole batch and decodes only its share of the
GOP work items (hwang_retriever_set_shard: a deterministic longest-first
split); no frame data is exchanged.  The test then gathers the outputs - which
bench.py never does - to check that the shares are disjoint, that together
they cover every row of every request exactly once, and that each frame sits
at its row's position."""
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
from tests.conftest import STAMP, load_stamp_backend  # noqa: E402

hw.stamp_fixture = load_stamp_backend()
hw.STAMP = STAMP
from hwang_amd.decoder import BatchRetriever  # noqa: E402

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
    blob = b"".join(samples)
    requests, bufs = [], []
    for k in range(world):
        rows = sorted(np.random.default_rng(SEED + k).choice(n, round(0.01 * n), replace=False).tolist())
        buf = np.full((len(rows), h, w, 3), 255, dtype=np.uint8)   # 255: "not written by this rank"
        bufs.append((rows, buf))
        requests.append((index, io.BytesIO(blob), rows, buf.ctypes.data, buf.nbytes))
    br = BatchRetriever(device_ids=[0], device_type=hw.DeviceType.CPU, decoder_type=hw.STAMP,
                        num_lanes=3, shard=(rank, world))
    dist.barrier()
    t0 = time.perf_counter()
    br.retrieve_into(requests)
    dist.barrier()
    elapsed = time.perf_counter() - t0
    st = br.stats()
    # STAMP: a frame carries the first four bytes of the sample it came from; rows of other shares stay 255
    mine = 0
    written = []
    for rows, buf in bufs:
        flat = buf.reshape(len(rows), -1)
        wr = ~(flat == 255).all(axis=1)
        for i in np.nonzero(wr)[0]:
            assert struct.unpack("<I", flat[i, :4].tobytes())[0] == rows[i], (rank, rows[i])
        mine += int(wr.sum())
        written.append(wr.astype(np.int32))
    assert mine == st["frames_returned"] > 0, (rank, mine, st)
    t = torch.tensor([elapsed], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    cnt = torch.tensor([mine, st["samples_fed"], st["intervals"]], dtype=torch.float64)
    dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
    cover = torch.from_numpy(np.concatenate(written))
    dist.all_reduce(cover, op=dist.ReduceOp.SUM)
    shares = [None] * world
    dist.all_gather_object(shares, (mine, st["intervals"]))
    if rank == 0:
        assert int(cnt[0]) == world * round(0.01 * n)
        assert bool((cover == 1).all()), "every row of every request exactly once over all ranks"
        assert all(s[0] > 0 and s[1] > 0 for s in shares), shares
        assert max(s[0] for s in shares) < 0.75 * int(cnt[0]), "the split must be balanced: %r" % (shares,)
        assert float(t) >= elapsed - 1e-9
        print("SHARDING_OK rows=%d fed=%d items=%d shares=%r" % (int(cnt[0]), int(cnt[1]), int(cnt[2]), shares))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

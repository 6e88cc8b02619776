"""Development aid: the H.264 path through the automata (feeding and fetching on two threads) against the retriever,
many times over - the check that found a race in the HEVC decoder's fetched-frame ring."""
import sys
import numpy as np
sys.path.insert(0, ".")
from tests.corpus_util import make_clip
import hwang_amd as hw
path = "/tmp/race_h264.mp4"
make_clip(path, width=320, height=240, frames=72, gop=12, bframes=2, refs=2, yuv=False)
idx = hw.index_video(path)
dec = hw.Decoder(path, video_index=idx, num_lanes=2)
for rows in ([3, 11, 12, 30], [24, 25, 26, 31, 60, 71], list(range(0, 72, 5)), list(range(10))):
    ref = dec.retrieve(rows)
    bad = 0
    for it in range(250):
        got = dec.retrieve_serial(rows)
        bad += sum(not np.array_equal(got[k], ref[k]) for k in range(len(rows)))
    print("H.264", rows[:8], "frames that differ in 250 requests:", bad)

#!/usr/bin/env python3
"""Development aid (GPU box): determinism soak of the banded row kernels.  The
same sparse request is decoded with 1 lane (one picture per launch, nothing
concurrent) and then repeatedly with many lanes (wide batches, all engines
busy); every returned frame must hash identically every time - a lost
hand-off between workgroups would show up as a differing frame."""
import hashlib
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hwang_amd as hw  # noqa: E402
import torch  # noqa: E402

W, H = int(os.environ.get("W", 1920)), int(os.environ.get("H", 1080))
FRAMES, REPEAT, ROUNDS = int(os.environ.get("FRAMES", 600)), int(os.environ.get("REPEAT", 20)), int(os.environ.get("ROUNDS", 12))
ENC_ARGS = os.environ.get("ENC_ARGS", "").split()   # extra corpus-encoder options, e.g. "--slices 3 --qp 40"
path = "/tmp/soak_%dx%d_%d_%d_%s.mp4" % (W, H, FRAMES, REPEAT, "".join(ENC_ARGS).replace("-", ""))
if not os.path.exists(path):
    subprocess.check_call([os.path.join(ROOT, "corpus/_build/h264_encoder"), "--width", str(W), "--height", str(H),
                           "--frames", str(FRAMES), "--gop", "60", "--bframes", "2", "--threads", "64",
                           "--repeat", str(REPEAT), "--out", path] + ENC_ARGS, stdout=subprocess.DEVNULL)
idx = hw.index_video(path)
n = idx.frames()
rng = np.random.default_rng(7)
rows = sorted(rng.choice(n, max(1, round(0.02 * n)), replace=False).tolist())
out = torch.empty((len(rows), H, W, 3), dtype=torch.uint8, device="cuda")


def digest():
    # per-frame sums on the GPU, hashed on the host: cheap and sensitive enough (any differing sample changes a sum)
    s = out.to(torch.int64).sum(dim=(1, 2, 3)).cpu().numpy()
    w = (out[:, ::7, ::5, :].to(torch.int64) * 31).sum(dim=(1, 2, 3)).cpu().numpy()
    return hashlib.sha1(s.tobytes() + w.tobytes()).hexdigest(), s


dec = hw.Decoder(path, video_index=idx, num_lanes=1)
t = time.time()
dec.retrieve_into(rows, out.data_ptr(), out.numel())
ref, ref_s = digest()
print("reference (1 lane): %d frames in %.1fs  %s" % (len(rows), time.time() - t, ref[:12]))
del dec
bad = 0
for lanes in (48, 64, 24):
    dec = hw.Decoder(path, video_index=idx, num_lanes=lanes)
    for r in range(ROUNDS):
        out.zero_()
        dec.retrieve_into(rows, out.data_ptr(), out.numel())
        d, s = digest()
        if d != ref:
            bad += 1
            print("MISMATCH lanes %d round %d: frames %s" % (lanes, r, np.nonzero(s != ref_s)[0][:10].tolist()))
    print("lanes %d: %d rounds done" % (lanes, ROUNDS))
    del dec
# the drop-in path: ONE DecoderAutomata, ONE backend with entropy workers (deferred conversions, copier threads, 16
# executor lanes) - frames come back in host memory and must be the same bytes
if int(os.environ.get("WORKERS", 1)):
    import hwang
    ridx = hwang.VideoIndex.deserialize(idx.serialize())
    ref_host = out.cpu().numpy()   # the last round above was identical to the reference, or counted as bad
    # (1, 0): the reference's unmodified caller - one interval per initialize, the backend's own workers, the pictures
    # of a GOP as chain launches (k_chain); the others: the grouped caller through the wide kernels
    for workers, group in ((1, 0), (16, int(os.environ.get("GROUP_MB", 16)) << 20), (4, int(os.environ.get("GROUP_MB", 16)) << 20)):
        rdec = hwang.Decoder(path, video_index=ridx, device_type=hwang.DeviceType.GPU, num_devices=workers,
                             interval_group_bytes=group)
        for r in range(ROUNDS):
            frames = rdec.retrieve(rows)
            wrong = [i for i, f in enumerate(frames) if not np.array_equal(f, ref_host[i])]
            if wrong:
                bad += 1
                print("MISMATCH workers %d group %d round %d: frames %s" % (workers, group, r, wrong[:10]))
            del frames
        print("workers %d, interval groups of %d bytes: %d rounds done" % (workers, group, ROUNDS))
        del rdec
print("soak: %s" % ("OK, every round identical" if bad == 0 else "%d rounds differ" % bad))
sys.exit(1 if bad else 0)

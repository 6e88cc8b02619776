"""One process of tests/test_gpu_chain.py (HWANG_CHAIN / HWANG_CHAIN_MAX_LANES are read when a lane registers, so every
configuration decodes in a process of its own): decodes the clips named on the command line and prints, per clip, the
MD5 of every returned frame's Y / U / V plane.
usage: _chain_worker.py spec.json   (spec: {"clips": [{"path", "width", "height", "rows": null | [...], "lanes", "api"}]})"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from tests.corpus_util import plane_md5s  # noqa: E402


def main():
    import hwang_amd as hw
    spec = json.load(open(sys.argv[1]))
    out = []
    for clip in spec["clips"]:
        idx = hw.index_video(clip["path"])
        rows = clip["rows"] if clip["rows"] is not None else list(range(idx.frames()))
        w, h = clip["width"], clip["height"]
        if clip.get("api") == "reference":
            # the reference's per-interval caller over the pybind11 module: ONE automata, chains of dependent pictures
            import hashlib
            import hwang
            ridx = hwang.VideoIndex.deserialize(idx.serialize())
            dec = hwang.Decoder(clip["path"], video_index=ridx, device_type=hwang.DeviceType.GPU, num_devices=clip.get("num_devices", 1),
                                interval_group_bytes=0)
            frames = dec.retrieve(rows)
            out.append([hashlib.md5(np.ascontiguousarray(f).tobytes()).hexdigest() for f in frames])
            del dec
        else:
            dec = hw.Decoder(clip["path"], video_index=idx, num_lanes=clip.get("lanes", 2))
            if clip.get("rgb"):
                import hashlib
                out.append([hashlib.md5(np.ascontiguousarray(f).tobytes()).hexdigest() for f in dec.retrieve(rows)])
            else:
                out.append([plane_md5s(f, w, h) for f in dec.retrieve_yuv(rows)])
            del dec
    print("RESULT " + json.dumps(out))


if __name__ == "__main__":
    main()

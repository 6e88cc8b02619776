"""Body of __graft_entry__.smoke(): hot path on cuda:0 vs the CPU oracle."""
import numpy as np
import torch


def run():
    import hwang_amd
    from hwang_amd._capi import lib
    from oracle import oracle_py

    # K5 on a random I420 picture (odd size: exercises the general path too)
    rng = np.random.default_rng(1)
    for (w, h) in ((64, 48), (50, 34)):
        cw, ch = (w + 1) // 2, (h + 1) // 2
        i420 = rng.integers(0, 256, w * h + 2 * cw * ch, dtype=np.uint8)
        d = torch.from_numpy(i420).cuda()
        out = torch.empty(h * w * 3, dtype=torch.uint8, device="cuda")
        y = d.data_ptr()
        rc = lib.hwang_k_yuv420_to_rgb24(y, y + w * h, y + w * h + cw * ch, w, cw, w, h,
                                         out.data_ptr(), None)
        assert rc == 0
        torch.cuda.synchronize()
        ref = oracle_py.i420_to_rgb(i420, w, h)
        assert np.array_equal(out.cpu().numpy().reshape(h, w, 3), ref), "RGB kernel != oracle"
    print("smoke: yuv420_to_rgb24 bit-exact vs oracle")

    # the whole hot path: a small I/P/B CABAC clip (8x8 transform, 2 refs)
    # through hwang_retriever_retrieve on cuda:0, sparse rows, YUV and RGB,
    # against the oracle decoding the same file on the CPU
    import tempfile

    from tests.corpus_util import make_clip
    with tempfile.TemporaryDirectory() as tmp:
        path = tmp + "/smoke.mp4"
        make_clip(path, width=176, height=144, yuv=False, frames=13, gop=13, bframes=2, cabac=1, t8x8=1, refs=2)
        index = hwang_amd.index_video(path)
        frames, _ = oracle_py.decode_file(path, index)
        rows = [0, 3, 4, 7, 12]
        dec = hwang_amd.Decoder(path, video_index=index, num_lanes=2)
        yuv = dec.retrieve_yuv(rows)
        rgb = dec.retrieve(rows)
        for k, r in enumerate(rows):
            buf, w, h, _ = frames[r]
            assert np.array_equal(yuv[k], buf), "frame %d: I420 != oracle" % r
            assert np.array_equal(rgb[k], oracle_py.i420_to_rgb(buf, w, h)), "frame %d: RGB != oracle" % r
        st = dec.stats()
        assert st["pictures_reconstructed"] > 0
    print("smoke: sparse decode of %d rows bit-exact vs oracle (%d pictures reconstructed on the GPU)"
          % (len(rows), st["pictures_reconstructed"]))

    # HEVC: a small I / P / B clip with SAO through the same entry points, against the HEVC oracle
    from tests.corpus_util import make_hevc_clip
    with tempfile.TemporaryDirectory() as tmp:
        path = tmp + "/smoke_hevc.mp4"
        make_hevc_clip(path, width=176, height=144, yuv=False, frames=9, gop=9, bframes=2, refs=2, sao=1, ctb=32)
        index = hwang_amd.index_video(path)
        frames, ctr = oracle_py.decode_file_hevc(path, index)
        assert ctr["md5_mismatch"] == 0 and ctr["md5_checked"] == 9
        rows = [0, 2, 3, 8]
        dec = hwang_amd.Decoder(path, video_index=index, num_lanes=2)
        yuv = dec.retrieve_yuv(rows)
        rgb = dec.retrieve(rows)
        for k, r in enumerate(rows):
            buf, w, h, _ = frames[r]
            assert np.array_equal(yuv[k], buf), "HEVC frame %d: I420 != oracle" % r
            assert np.array_equal(rgb[k], oracle_py.i420_to_rgb(buf, w, h)), "HEVC frame %d: RGB != oracle" % r
    print("smoke: HEVC sparse decode of %d rows bit-exact vs the HEVC oracle" % len(rows))

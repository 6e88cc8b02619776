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

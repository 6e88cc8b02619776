"""-m gpu: HIP kernels vs the CPU oracle through the C ABI (bit-exact)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a GPU; there is no fallback"
    return torch


@pytest.mark.parametrize("w,h", [(1920, 1080), (64, 48), (50, 34), (322, 242), (8, 2), (3840, 2160)])
def test_yuv420_to_rgb24(hw, oracle, torch_cuda, w, h):
    torch = torch_cuda
    from hwang_amd._capi import lib
    rng = np.random.default_rng(w * 31 + h)
    cw, ch = (w + 1) // 2, (h + 1) // 2
    i420 = rng.integers(0, 256, w * h + 2 * cw * ch, dtype=np.uint8)
    # include the extremes that exercise saturation
    i420[:16] = [0, 255, 16, 235, 0, 255, 1, 254, 0, 0, 255, 255, 128, 128, 17, 234]
    d = torch.from_numpy(i420).cuda()
    out = torch.zeros(h * w * 3, dtype=torch.uint8, device="cuda")
    y = d.data_ptr()
    assert lib.hwang_k_yuv420_to_rgb24(y, y + w * h, y + w * h + cw * ch, w, cw, w, h,
                                       out.data_ptr(), None) == 0
    torch.cuda.synchronize()
    ref = oracle.i420_to_rgb(i420, w, h)
    assert np.array_equal(out.cpu().numpy().reshape(h, w, 3), ref)

"""HEVC behind the boundary, as far as it can be checked on this pool: the indexer
names an hev1 / hvc1 track and hands out its hvcC payload exactly as the reference's indexer does (compiled in place:
oracle/_ref/ref_mp4_index, mp4_index_creator.cpp:454-469); the backend's format switch recognises the HEVC names of the
reference's software backend (software_video_decoder.cpp:128-132), reads the record and refuses the track by name.  There
is no HEVC decoder: no HEVC stream and no HEVC decoder exist on the build image or the GPU boxes to check one against."""
import ctypes as C
import os

import pytest

from tests.corpus_util import make_clip
from tests.util import ROOT


def describe(lib, payload):
    out = C.create_string_buffer(512)
    rc = lib.hwang_hevc_describe_config(payload, len(payload), out, 512)
    return rc, out.value.decode()


@pytest.mark.parametrize("entry,fourcc", [(1, "hev1"), (2, "hvc1")])
def test_hevc_track_is_indexed_and_described(hw, tmp_path, entry, fourcc):
    from hwang_amd._capi import lib
    path = str(tmp_path / (fourcc + ".mp4"))
    make_clip(path, width=320, height=180, yuv=False, frames=12, gop=6, bframes=1, hevc_entry=entry)
    idx = hw.index_video(path)
    assert idx.format() == fourcc and idx.frames() == 12 and idx.keyframe_indices() == [0, 6]
    payload = bytes(idx.metadata_bytes())
    assert payload[0] == 1 and len(payload) > 23
    rc, text = describe(lib, payload)
    assert rc == 0
    # 320x180 is coded as 320x184 (a multiple of the 8-sample minimum coding block) with a conformance window
    assert text == ("HEVC Main profile, Main tier, level 4.0; 320x180 (coded 320x184), 4:2:0 8-bit, CTB 64; "
                    "1 VPS / 1 SPS / 1 PPS, 4-byte NAL lengths"), text


def test_hev1_index_equals_the_reference_indexers(hw, tmp_path):
    """The reference's indexer knows 'hev1' (not 'hvc1': mp4_index_creator.cpp:454): same tables, same extradata."""
    import json
    import subprocess
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "ref_mp4_index")   # the reference indexer compiled in place
    if not os.path.exists(ref_bin):
        pytest.skip("reference indexer not built here")
    path = str(tmp_path / "hev1.mp4")
    make_clip(path, width=640, height=368, yuv=False, frames=20, gop=10, bframes=2, hevc_entry=1)
    ref = json.loads(subprocess.check_output([ref_bin, path]))
    idx = hw.index_video(path)
    assert not ref["error"] and ref["format"] == "hev1" == idx.format()
    assert ref["metadata"] == bytes(idx.metadata_bytes()).hex() and len(ref["metadata"]) > 46
    assert ref["sample_offsets"] == idx.sample_offsets() and ref["sample_sizes"] == idx.sample_sizes()
    assert ref["keyframes"] == idx.keyframe_indices()
    assert (ref["width"], ref["height"]) == (idx.frame_width(), idx.frame_height())


def test_damaged_records_are_named(hw):
    from hwang_amd._capi import lib
    assert describe(lib, b"\x01" * 10) == (1, "the hvcC payload is shorter than the 23 bytes of its fixed part")
    rc, text = describe(lib, b"\x02" + b"\x00" * 30)
    assert rc == 1 and "configurationVersion 2" in text
    good = bytes([1, 1, 0x60, 0, 0, 0, 0x90, 0, 0, 0, 0, 0, 93, 0xf0, 0, 0xfc, 0xfd, 0xfa, 0xfa, 0, 0, 0x0f, 1, 0xa1, 0, 1, 0, 40])
    rc, text = describe(lib, good)   # an SPS array that claims 40 bytes and has none
    assert rc == 1 and "runs past the end" in text
    rc, text = describe(lib, good[:23][:22] + b"\x00")   # no arrays at all: described from the record's header alone
    assert rc == 0 and text.startswith("HEVC Main profile, Main tier, level 3.1; 4:2:0 10-bit (no SPS in the record)"), text


class _Bits:
    def __init__(self):
        self.bits = []

    def u(self, n, v):
        self.bits += [(v >> (n - 1 - i)) & 1 for i in range(n)]

    def ue(self, v):
        n = (v + 1).bit_length()
        self.u(n - 1, 0)
        self.u(n, v + 1)

    def nal(self, header):
        bits = self.bits + [1]
        bits += [0] * (-len(bits) % 8)
        raw = bytes(int("".join(map(str, bits[i:i + 8])), 2) for i in range(0, len(bits), 8))
        out, zeros = bytearray(header), 0
        for b in raw:   # emulation prevention (7.4.2)
            if zeros >= 2 and b <= 3:
                out.append(3)
                zeros = 0
            out.append(b)
            zeros = zeros + 1 if b == 0 else 0
        return bytes(out)


def _hvcc_with_sps(width, height, crop):
    b = _Bits()
    b.u(4, 0); b.u(3, 0); b.u(1, 1)                 # VPS id, max_sub_layers_minus1, temporal id nesting
    b.u(2, 0); b.u(1, 0); b.u(5, 1); b.u(32, 0x60000000); b.u(4, 9); b.u(43, 0); b.u(1, 0); b.u(8, 93)   # profile_tier_level
    b.ue(0); b.ue(1); b.ue(width); b.ue(height)     # SPS id, 4:2:0, size
    b.u(1, 1)
    for v in crop:
        b.ue(v)
    b.ue(0); b.ue(0); b.ue(4); b.u(1, 1); b.ue(1); b.ue(0); b.ue(0)   # bit depths, POC bits, one ordering entry
    b.ue(0); b.ue(3); b.ue(0); b.ue(3)              # coding block 8 .. 64, transform block 4 .. 32
    sps = b.nal(bytes([0x42, 0x01]))
    head = bytes([1, 1, 0x60, 0, 0, 0, 0x90, 0, 0, 0, 0, 0, 93, 0xf0, 0, 0xfc, 0xfd, 0xf8, 0xf8, 0, 0, 0x0f])
    return head + bytes([1, 0xa1, 0, 1]) + len(sps).to_bytes(2, "big") + sps


def test_hostile_conformance_windows_are_not_believed(hw):
    """crop offsets are Exp-Golomb values an attacker picks - up to 2^32 - 2 each, times the chroma
    sub-sampling factor: no signed overflow (this file is in the sanitizer build), no negative size in the text."""
    from hwang_amd._capi import lib
    rc, text = describe(lib, _hvcc_with_sps(640, 368, (0, 0, 0, 4)))
    assert rc == 0 and "; 640x360 (coded 640x368), 4:2:0 8-bit, CTB 64;" in text, text
    for crop in [(0x7fffffff, 0, 0, 0), (0xfffffffe, 0xfffffffe, 0xfffffffe, 0xfffffffe), (160, 160, 0, 0), (0, 0, 92, 92),
                 (0x40000000, 0x40000000, 0, 0)]:
        rc, text = describe(lib, _hvcc_with_sps(640, 368, crop))
        assert rc == 0 and "(its SPS could not be read)" in text and "-" not in text.replace("-bit", "").replace("-byte", ""), (crop, text)


@pytest.mark.gpu
def test_backend_takes_hevc_by_every_name_the_reference_reads(hw, tmp_path):
    """The format switch of software_video_decoder.cpp:124-135: "h265" / "hev1" / "hevc" (and the other sample entry
    type, "hvc1") select the HEVC half of the backend; an unknown name is refused with the list of what is supported;
    an hev1 track whose samples are not HEVC (the container fixtures of the H.264 corpus generator) fails as a Result,
    not as a crash."""
    import torch
    assert torch.cuda.is_available()
    from hwang_amd._capi import lib
    from tests.corpus_util import make_hevc_clip
    path = str(tmp_path / "real.mp4")
    make_hevc_clip(path, width=320, height=240, yuv=False, frames=3, gop=3)
    idx = hw.index_video(path)
    dec = lib.hwang_decoder_create(1, 0, 1, 3)
    assert dec
    extradata = bytes(idx.metadata_bytes())
    try:
        for name in ("hev1", "hevc", "h265", "hvc1"):
            assert lib.hwang_decoder_configure(dec, 320, 240, name.encode(), extradata, len(extradata)) == 0, lib.hwang_decoder_last_error(dec).decode()
        assert lib.hwang_decoder_configure(dec, 322, 240, b"hev1", extradata, len(extradata)) != 0
        assert "but the container says 322x240" in lib.hwang_decoder_last_error(dec).decode()
        assert lib.hwang_decoder_configure(dec, 320, 240, b"vp09", extradata, len(extradata)) != 0
        msg = lib.hwang_decoder_last_error(dec).decode()
        assert "Unsupported video codec: vp09" in msg and "hevc" in msg and "h264" in msg
    finally:
        lib.hwang_decoder_destroy(dec)
    fake = str(tmp_path / "fake.mp4")
    make_clip(fake, width=320, height=240, yuv=False, frames=6, gop=6, bframes=0, hevc_entry=1)
    import hwang
    with pytest.raises(RuntimeError):
        hwang.Decoder(fake, device_type=hwang.DeviceType.GPU).retrieve([0])

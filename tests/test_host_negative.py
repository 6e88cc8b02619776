"""Hostile inputs must produce errors, never crashes.

* crafted SPS/PPS fields through hwang_h264_probe (the same code any MP4
  handed to index_video/Decoder reaches): out-of-range log2_max_frame_num,
  log2_max_poc_lsb, max_num_ref_frames, cropping larger than the picture,
  absurd Exp-Golomb prefixes;
* random bit flips over a valid clip's samples and avcC: the parser may
  report an error or decode garbage, but returns;
* malformed indexes: unequal offset/size lengths, unsorted or out-of-range
  keyframes, `stss` entry 0, `stsz` counts beyond the file."""
import ctypes as C
import struct

import numpy as np
import pytest

from tests.corpus_util import make_clip


class BitW:
    def __init__(self):
        self.bits = []

    def u(self, n, v):
        for i in range(n - 1, -1, -1):
            self.bits.append((v >> i) & 1)
        return self

    def ue(self, v):
        v += 1
        n = v.bit_length()
        self.u(n - 1, 0)
        self.u(n, v)
        return self

    def se(self, v):
        return self.ue(2 * v - 1 if v > 0 else -2 * v)

    def trailing(self):
        self.bits.append(1)
        while len(self.bits) % 8:
            self.bits.append(0)
        return self

    def bytes(self):
        out = bytearray()
        for i in range(0, len(self.bits), 8):
            b = 0
            for j in range(8):
                b = (b << 1) | (self.bits[i + j] if i + j < len(self.bits) else 0)
            out.append(b)
        # emulation prevention
        esc = bytearray()
        z = 0
        for b in out:
            if z >= 2 and b <= 3:
                esc.append(3)
                z = 0
            esc.append(b)
            z = z + 1 if b == 0 else 0
        return bytes(esc)


def sps(log2_fn=0, poc_type=0, log2_poc=2, refs=1, mbw=20, mbh=15, crop=None, profile=66):
    w = BitW()
    w.u(8, profile).u(8, 0).u(8, 30).ue(0)
    if profile == 100:
        w.ue(1).ue(0).ue(0).u(1, 0).u(1, 0)
    w.ue(log2_fn).ue(poc_type)
    if poc_type == 0:
        w.ue(log2_poc)
    w.ue(refs).u(1, 0).ue(mbw - 1).ue(mbh - 1).u(1, 1).u(1, 1)
    if crop:
        w.u(1, 1)
        for c in crop:
            w.ue(c)
    else:
        w.u(1, 0)
    w.u(1, 0).trailing()
    return b"\x67" + w.bytes()


def pps():
    w = BitW()
    w.ue(0).ue(0).u(1, 0).u(1, 0).ue(0).ue(0).ue(0).u(1, 0).u(2, 0).se(0).se(0).se(0).u(1, 0).u(1, 0).u(1, 0).trailing()
    return b"\x68" + w.bytes()


def avcc(sps_nal, pps_nal):
    return (bytes([1, 66, 0, 30, 0xFF, 0xE1]) + struct.pack(">H", len(sps_nal)) + sps_nal + bytes([1]) +
            struct.pack(">H", len(pps_nal)) + pps_nal)


def idr_slice():
    w = BitW()
    w.ue(0).ue(7).ue(0)
    w.u(32, 0xFFFFFFFF).u(32, 0xFFFFFFFF).u(32, 0xA5A5A5A5).u(32, 0x12345678)  # frame_num etc. + junk
    for _ in range(16):
        w.u(32, 0x9E3779B9)
    nal = b"\x65" + w.bytes()
    return struct.pack(">I", len(nal)) + nal


def run_probe(lib, extradata, sample):
    data = np.frombuffer(sample, dtype=np.uint8).copy()
    offs = np.array([0], dtype=np.uint64)
    sizes = np.array([len(sample)], dtype=np.uint64)
    out = (C.c_int64 * 8)()
    err = C.create_string_buffer(256)
    rc = lib.hwang_h264_probe(extradata, len(extradata), data.ctypes.data_as(C.POINTER(C.c_uint8)),
                              offs.ctypes.data_as(C.POINTER(C.c_uint64)), sizes.ctypes.data_as(C.POINTER(C.c_uint64)),
                              1, None, 0, out, err, 256)
    return rc, err.value.decode()


@pytest.mark.parametrize("kw,needle", [
    (dict(log2_fn=0x00FFFFFC), "log2_max_frame_num"),
    (dict(log2_fn=0x8000000C), "log2_max_frame_num"),
    (dict(log2_fn=13), "log2_max_frame_num"),
    (dict(log2_poc=0x00FFFFFC), "pic_order_cnt_lsb"),
    (dict(log2_poc=13), "pic_order_cnt_lsb"),
    (dict(refs=17), "max_num_ref_frames"),
    (dict(refs=0x7FFFFFF0), "max_num_ref_frames"),
    (dict(crop=(80, 80, 0, 0)), "cropping"),
    (dict(crop=(0, 0, 60, 60)), "cropping"),
    (dict(crop=(0x7FFFFFFF, 0x7FFFFFFF, 0, 0)), "cropping"),
    (dict(crop=(0xFFFFFFFE, 2, 0, 0)), "cropping"),
    (dict(mbw=513), "size"),
    (dict(mbw=0x7FFFFFFF), "size"),
    (dict(poc_type=3), "pic_order_cnt_type"),
])
def test_crafted_sps_is_refused(hw, kw, needle):
    from hwang_amd._capi import lib
    rc, err = run_probe(lib, avcc(sps(**kw), pps()), idr_slice())
    assert rc != 0 and needle in err, err


def test_sane_sps_with_junk_slice_returns(hw):
    """A valid header followed by a garbage slice: error or garbage, but it returns."""
    from hwang_amd._capi import lib
    for kw in (dict(), dict(log2_fn=12, log2_poc=12, refs=16), dict(crop=(0, 4, 0, 4))):
        run_probe(lib, avcc(sps(**kw), pps()), idr_slice())


def test_exp_golomb_prefix_longer_than_32_bits(hw):
    from hwang_amd._capi import lib
    bad = b"\x67" + bytes([66, 0, 30]) + b"\x00" * 12 + b"\x80"  # sps_id with 96 leading zeros
    rc, err = run_probe(lib, avcc(bad, pps()), idr_slice())
    assert rc != 0


def test_bit_flips_never_crash(hw, tmp_path):
    from hwang_amd._capi import lib
    path = str(tmp_path / "f.mp4")
    make_clip(path, yuv=False, width=96, height=64, frames=7, gop=7, bframes=2, cabac=1, t8x8=1, refs=2)
    idx = hw.index_video(path)
    data0 = np.fromfile(path, dtype=np.uint8)
    offs = np.ascontiguousarray(idx.sample_offsets_np().astype(np.uint64))
    sizes = np.ascontiguousarray(idx.sample_sizes_np().astype(np.uint64))
    extradata0 = bytes(idx.metadata_bytes())
    rng = np.random.default_rng(7)
    errors = 0
    for trial in range(150):
        data = data0.copy()
        extradata = bytearray(extradata0)
        if trial % 3 == 0:
            for _ in range(1 + trial % 4):
                extradata[int(rng.integers(5, len(extradata)))] ^= 1 << int(rng.integers(0, 8))
        else:
            for _ in range(1 + trial % 6):
                s = int(rng.integers(0, len(offs)))
                pos = int(offs[s]) + int(rng.integers(4, int(sizes[s])))
                data[pos] ^= 1 << int(rng.integers(0, 8))
        out = (C.c_int64 * 8)()
        err = C.create_string_buffer(256)
        rc = lib.hwang_h264_probe(bytes(extradata), len(extradata), data.ctypes.data_as(C.POINTER(C.c_uint8)),
                                  offs.ctypes.data_as(C.POINTER(C.c_uint64)),
                                  sizes.ctypes.data_as(C.POINTER(C.c_uint64)), idx.frames(), None, 0, out, err, 256)
        errors += rc != 0
    assert errors > 0  # some flips must be detected; none may crash


def test_duplicated_slice_nal_is_refused(hw, tmp_path):
    """a sample whose last slice NAL is repeated re-decodes macroblocks of the picture in
    progress; the per-picture streams (motion extras, coefficients) must not grow past their buffers.
    Slices have to start at the next undecoded macroblock - anything else is an error."""
    from hwang_amd._capi import lib
    for kw in (dict(cabac=0, slices=3, qp=51), dict(cabac=1, slices=2, qp=30)):
        path = str(tmp_path / "d.mp4")
        make_clip(path, yuv=False, width=64, height=48, frames=6, gop=6, bframes=0, refs=2, **kw)
        idx = hw.index_video(path)
        raw = open(path, "rb").read()
        offs, sizes = idx.sample_offsets(), idx.sample_sizes()
        blob, noffs, nsizes = bytearray(), [], []
        for o, n in zip(offs, sizes):
            sample = raw[o:o + n]
            pos, last = 0, None
            while pos + 4 <= len(sample):
                ln = struct.unpack(">I", sample[pos:pos + 4])[0]
                last = (pos, 4 + ln)
                pos += 4 + ln
            assert sample[last[0] + 4] & 31 in (1, 5)
            sample = sample + sample[last[0]:last[0] + last[1]] * 250
            noffs.append(len(blob))
            nsizes.append(len(sample))
            blob += sample
        data = np.frombuffer(bytes(blob), dtype=np.uint8)
        o = np.ascontiguousarray(np.array(noffs, dtype=np.uint64))
        z = np.ascontiguousarray(np.array(nsizes, dtype=np.uint64))
        ex = bytes(idx.metadata_bytes())
        out = (C.c_int64 * 8)()
        err = C.create_string_buffer(256)
        rc = lib.hwang_h264_probe(ex, len(ex), data.ctypes.data_as(C.POINTER(C.c_uint8)),
                                  o.ctypes.data_as(C.POINTER(C.c_uint64)), z.ctypes.data_as(C.POINTER(C.c_uint64)),
                                  len(noffs), None, 0, out, err, 256)
        assert rc != 0 and b"next undecoded macroblock" in err.value, err.value


# ------------------------------------------------------------------ indexes
def test_inconsistent_index_is_refused(hw):
    good = dict(timescale=90000, duration=9000, width=64, height=48, format="avc1",
                sample_offsets=[100, 200, 300, 400], sample_sizes=[100, 100, 100, 100], keyframe_indices=[0, 2],
                metadata=b"\x01\x42\x00\x1e")
    hw.VideoIndex(**good)
    assert hw.VideoIndex(**good).problem() == ""
    with pytest.raises((RuntimeError, ValueError)):
        hw.VideoIndex(**dict(good, sample_sizes=[100, 100, 100]))
    for patch in (dict(keyframe_indices=[2, 0]), dict(keyframe_indices=[0, 0]), dict(keyframe_indices=[0, 4]),
                  dict(keyframe_indices=[])):
        bad = hw.VideoIndex(**dict(good, **patch))
        assert bad.problem(), patch
        with pytest.raises(RuntimeError):
            hw.slice_into_video_intervals(bad, [0])
    # the reference's own wire golden has keyframes beyond its 3 samples: it can be held and
    # re-serialized, but nothing decodes from it
    wire = bytes.fromhex("08c00210f0011a05208f29c22f2206ef28b3068d092a02001e3204016400283890bf054098cb064a0461766331")
    idx = hw.VideoIndex.deserialize(wire)
    assert idx.serialize() == wire and idx.problem()
    with pytest.raises(RuntimeError):
        hw.slice_into_video_intervals(idx, [0])
    import io
    with pytest.raises(RuntimeError) as e:
        hw.Decoder(io.BytesIO(b"\0" * 64), video_index=idx, device_type=hw.DeviceType.CPU,
                   decoder_type=hw.STAMP, num_lanes=1).retrieve([0])
    assert "keyframe" in str(e.value)


def _box(kind, payload):
    return struct.pack(">I", 8 + len(payload)) + kind + payload


def _full(kind, payload, version=0, flags=0):
    return _box(kind, struct.pack(">I", (version << 24) | flags) + payload)


def tiny_mp4(stss=(1,), stsz_count=2, stsz_fixed=0, spc=1, sizes=(10, 12), avcc_payload=None):
    """Smallest moov the indexer accepts, with knobs for the malformed cases."""
    avcc_payload = avcc_payload or avcc(sps(), pps())
    entry = (b"\0" * 6 + struct.pack(">H", 1) + b"\0" * 16 + struct.pack(">HH", 320, 240) + struct.pack(">II", 0x480000, 0x480000) +
             b"\0" * 4 + struct.pack(">H", 1) + b"\0" * 32 + struct.pack(">Hh", 24, -1) + _box(b"avcC", avcc_payload))
    stsd = _full(b"stsd", struct.pack(">I", 1) + _box(b"avc1", entry))
    stts = _full(b"stts", struct.pack(">III", 1, len(sizes), 3000))
    stss_b = _full(b"stss", struct.pack(">I", len(stss)) + b"".join(struct.pack(">I", s) for s in stss))
    stsc = _full(b"stsc", struct.pack(">IIII", 1, 1, spc, 1))
    if stsz_fixed:
        stsz = _full(b"stsz", struct.pack(">II", stsz_fixed, stsz_count))
    else:
        stsz = _full(b"stsz", struct.pack(">II", 0, stsz_count) + b"".join(struct.pack(">I", s) for s in sizes))
    ftyp = _box(b"ftyp", b"isom\0\0\2\0isomavc1")
    mdat = _box(b"mdat", b"\0" * sum(sizes))
    off = len(ftyp) + 8
    stco = _full(b"stco", struct.pack(">I", len(sizes)) + b"".join(
        struct.pack(">I", off + sum(sizes[:i])) for i in range(len(sizes))))
    stbl = _box(b"stbl", stsd + stts + stss_b + stsc + stsz + stco)
    hdlr = _full(b"hdlr", b"\0" * 4 + b"vide" + b"\0" * 12 + b"v\0")
    mdhd = _full(b"mdhd", struct.pack(">IIIIHH", 0, 0, 90000, 6000, 0, 0))
    minf = _box(b"minf", _full(b"vmhd", b"\0" * 8, flags=1) + stbl)
    mdia = _box(b"mdia", mdhd + hdlr + minf)
    trak = _box(b"trak", mdia)
    mvhd = _full(b"mvhd", struct.pack(">IIII", 0, 0, 90000, 6000) + b"\0" * 80)
    return ftyp + mdat + _box(b"moov", mvhd + trak)


def index_bytes(hw, blob):
    c = hw.MP4IndexCreator(len(blob))
    off, size = 0, min(1024, len(blob))
    for _ in range(64):
        if c.is_done():
            break
        _more, off, size = c.feed(blob[off:off + size], size)
    return c


def test_tiny_mp4_indexes(hw):
    c = index_bytes(hw, tiny_mp4())
    assert c.is_done() and not c.is_error(), c.error_message()
    idx = c.get_video_index()
    assert idx.frames() == 2 and list(idx.keyframe_indices()) == [0]


@pytest.mark.parametrize("kw", [
    dict(stss=(0,)),                         # 1-based table holding 0
    dict(stss=(3,)),                         # beyond sample_count
    dict(stss=(2, 1)),                       # not increasing
    dict(stsz_count=0xFFFFFFFF, stsz_fixed=4096),   # 2^32-1 samples of fixed size in a 1 KB file
    dict(stsz_count=0x10000000),             # table longer than the file
    dict(spc=0),                             # zero samples per chunk
])
def test_malformed_tables_are_errors(hw, kw):
    c = index_bytes(hw, tiny_mp4(**kw))
    assert c.is_error() or not c.is_done(), "accepted a malformed table: %r" % (kw,)
    if c.is_error():
        assert c.error_message()


def test_limits_are_errors_not_overruns(hw, tmp_path):
    """A picture cut into more slices than the per-picture slice table holds (256) is refused by name; 255 slices
    decode (host side: every slice ends on its trailing bits)."""
    from tests.corpus_util import make_clip
    from tests.test_host_parser import probe
    many = str(tmp_path / "many.mp4")
    make_clip(many, yuv=False, frames=3, gop=3, bframes=0, slice_mbs=1)       # 300 slices per picture
    with pytest.raises(AssertionError, match="too many slices in one picture"):
        probe(hw, many)
    ok = str(tmp_path / "ok.mp4")
    make_clip(ok, yuv=False, width=240, height=272, frames=3, gop=3, bframes=1, slice_mbs=1)   # 15 x 17 = 255 slices
    idx, r = probe(hw, ok)
    assert r["slices"] == r["clean_end"] == 3 * 255 and r["pictures"] == 3
